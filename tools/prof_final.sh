#!/bin/bash
# round-end evidence: default bench line, rocprofv3 kernel stats of the same command, PMC passes (memory side, SQ side)
R=$GRAFT_REPO_ROOT
cd $R && python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o final -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1 || exit 1
for p in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "WRITE_SIZE" "FETCH_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $p | cut -d' ' -f1)
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_$tag -o pmc -- python3 $R/bench.py --steps 32 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcf_$tag.log 2>&1 || exit 1
done
bash $R/tools/prof_alone.sh > $R/gpurun_out/alone.txt 2>&1
