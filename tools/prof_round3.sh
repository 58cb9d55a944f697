#!/bin/bash
# round-3 evidence (run on the GPU box): bench lines, rocprofv3 kernel stats of the default command, PMC passes (memory side and
# SQ side, separate passes, no trace domains beside --kernel-trace), timelines of one group alone and of the 20-step job
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03prof
mkdir -p $O
cd $R && python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_5.json 2> $O/bench_20_5.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-cpu-baseline --consumer-shard 0 > $O/stats_bench.json 2> $O/stats.err || exit 1
for p in "WRITE_SIZE" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $p | cut -d' ' -f1)
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmc_$tag -o pmc -- python3 $R/bench.py --steps 48 --warmup 16 --no-cpu-baseline --consumer-shard 0 > $O/pmc_$tag.json 2> $O/pmc_$tag.err || exit 1
done
rocprofv3 --kernel-trace --output-format csv -d $O/alone -o a -- python3 $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline --consumer-shard 0 > $O/alone.json 2> $O/alone.err || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/short -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --consumer-shard 0 > $O/short.json 2> $O/short.err || exit 1
cd $R
python3 tools/pmc_summary.py $O/pmc_WRITE_SIZE $O/pmc_FETCH_SIZE --json > $O/pmc_hbm.txt
python3 tools/pmc_summary.py $O/pmc_SQ_WAVE_CYCLES > $O/pmc_sq.txt
python3 tools/pmc_summary.py $O/pmc_TCP_TCC_WRITE_REQ_sum > $O/pmc_l2.txt
python3 tools/timeline.py $(ls $O/alone/*kernel_trace.csv $O/alone/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/alone_timeline.txt
python3 tools/timeline.py $(ls $O/short/*kernel_trace.csv $O/short/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/short_timeline.txt
echo done
python3 tools/bench_configs.py > $O/side_configs.jsonl 2> $O/side_configs.err
make -C tools engine_bench > /dev/null 2>&1 && tools/engine_bench > $O/engine_bench_c.json 2> $O/engine_bench_c.err
echo all done
