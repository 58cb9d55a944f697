#!/bin/bash
# timing experiments with the debug build (build.py --out build/libblsw_dbg.so -DBLSW_DEBUG_KNOBS): which part of a step costs what
run() { # name lib skip variant
  export BLSW_DEBUG_SKIP=$3 BLSW_EXPAND_VARIANT=$4
  if [ "$2" = default ]; then unset BLSW_LIB; else export BLSW_LIB=$PWD/build/$2; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --allgather-steps 0 $BENCH_ARGS > gpurun_out/dbg_$1.json 2>gpurun_out/dbg_$1.err && echo "$1 $(python -c "import json;d=json.load(open('gpurun_out/dbg_$1.json'));print(round(d['value']),round(d['ms_per_step'],2),round(d['roofline']['avg_launch_ms'],2),d['config']['results_ok'])")"
}
for v in ${VARIANTS:-0}; do
  run expandonly_v$v libblsw_dbg.so 3 $v && { [ -n "$ONLY" ] || run full_v$v libblsw_dbg.so 0 $v; } || exit 1
done
