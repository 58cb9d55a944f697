#!/bin/bash
# PMC passes for the memory-side picture of every kernel (separate passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for p in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $p | cut -d' ' -f1)
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python3 $R/bench.py --steps 32 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1 || exit 1
done
