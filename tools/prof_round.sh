#!/bin/bash
# Round evidence (run on the GPU box: gpurun -- tools/prof_round.sh r05): bench lines, rocprofv3 kernel stats of the default command, PMC passes
# (memory side and SQ side, separate passes, no trace domains beside --kernel-trace; the profiled commands run the headline only: --side-legs 0),
# the digest kernel alone and its VALU count, timelines of the short job and of the consumer-mode shard, blsw_verify_batch rates.
# tools/collect_profiles.sh <tag> copies the summaries into profiles/.
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}prof
mkdir -p $O
cd $R && python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err || exit 1
python3 tools/digest_rate.py > $O/digest_rate.json 2> $O/digest_rate.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-cpu-baseline --consumer-shard 0 --side-legs 0 > $O/stats_bench.json 2> $O/stats.err || exit 1
for p in "WRITE_SIZE" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES"; do
  tag=$(echo $p | cut -d' ' -f1)
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmc_$tag -o pmc -- python3 $R/bench.py --steps 48 --warmup 16 --no-cpu-baseline --consumer-shard 0 --side-legs 0 > $O/pmc_$tag.json 2> $O/pmc_$tag.err || exit 1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_digest -o pmc -- python3 $R/tools/digest_rate.py 1024 3 > $O/pmc_digest.json 2> $O/pmc_digest.err || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/short -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --consumer-shard 0 --side-legs 0 > $O/short.json 2> $O/short.err || exit 1
# the 8 192-instance consumer-mode shard (bench.py's value_consumer_mode leg) under the kernel trace: the first group's critical path
rocprofv3 --kernel-trace --output-format csv -d $O/consumer -o t -- python3 $R/tools/consumer_probe.py "[8192,null,false]" "[8192,null,false]" > $O/consumer_probe.txt 2> $O/consumer.err || exit 1
cd $R
python3 tools/timeline.py $(ls $O/consumer/*kernel_trace.csv $O/consumer/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/consumer_timeline_full.txt
rm -rf $O/consumer
# the same shard with no profiler attached: stage times of every launch group from the engine's own HIP events (BLSW_TRACE_GROUP) + first_step_ms
BLSW_TRACE_GROUP=1 python3 tools/consumer_probe.py "[8192,null,false]" "[8192,null,false]" "[8192,null,false]" "[8192,null,false]" "[8192,null,false]" "[32768,null,false]" > $O/consumer_group_trace.txt 2>&1
python3 tools/verify_rate.py > $O/verify_rate.txt 2> $O/verify_rate.err
python3 tools/pmc_summary.py $O/pmc_WRITE_SIZE $O/pmc_FETCH_SIZE --json > $O/pmc_hbm.txt
python3 tools/pmc_summary.py $O/pmc_SQ_WAVE_CYCLES > $O/pmc_sq.txt
python3 tools/pmc_summary.py $O/pmc_digest > $O/pmc_digest.txt
python3 tools/timeline.py $(ls $O/short/*kernel_trace.csv $O/short/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/short_timeline.txt
cp $(ls $O/stats/*kernel_stats.csv $O/stats/*/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
rm -rf $O/stats $O/short $O/pmc_WRITE_SIZE $O/pmc_FETCH_SIZE $O/pmc_SQ_WAVE_CYCLES $O/pmc_digest
echo done
python3 tools/bench_configs.py > $O/side_configs.jsonl 2> $O/side_configs.err
make -C tools engine_bench > /dev/null 2>&1 && tools/engine_bench > $O/engine_bench_c.json 2> $O/engine_bench_c.err
echo all done
