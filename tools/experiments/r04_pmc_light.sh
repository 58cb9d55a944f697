#!/bin/bash
# VALU wave-instructions per launch of the shipped expansion kernel and of the light one (rocprofv3 --pmc, own pass each)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_pmc_light
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 10 11; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/v$v -o pmc -- python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --consumer-shard 0 --expand-variant $v > $O/v$v.json 2> $O/v$v.err || exit 1
  python3 $R/tools/pmc_summary.py $O/v$v | grep -E "k_sha_expand" | sed "s/^/expand_variant $v: /"
  rm -rf $O/v$v
done
