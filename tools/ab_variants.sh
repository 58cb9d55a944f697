#!/bin/bash
# A/B of library builds inside ONE gpurun call (box-to-box spread is larger than most deltas): for every build given as
# name=path (path "default" = the in-tree library): chain kernels only, the driver-style short job twice, a 256-step run.
# usage: tools/ab_variants.sh out_prefix name=path [name=path ...]
out=$1; shift
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = default ]; then unset BLSW_LIB; else export BLSW_LIB=$PWD/$lib; fi
  c=$(timeout -k 10 200 python tools/chains_only.py 256 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['instances_per_s']))")
  line="$name chains_only $c"
  for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 256 --warmup 48"; do
    v=$(timeout -k 10 200 python bench.py --no-cpu-baseline --allgather-steps 0 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), d['config']['results_ok'])")
    line="$line | $args: $v"
  done
  echo "$line" | tee -a gpurun_out/$out.txt
done
