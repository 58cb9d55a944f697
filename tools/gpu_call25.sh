#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_ab_global2.txt
for round in 1 2 3; do
  for lib in build/libblsw_flat.so bls-verify-gadget_amd/libblsw.so; do
    BLSW_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --consumer-shard 0 > gpurun_out/ab_gl.json 2> gpurun_out/ab_gl.err || exit 1
    BLSW_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 512 --warmup 48 --no-cpu-baseline --consumer-shard 0 > gpurun_out/ab_gl2.json 2>> gpurun_out/ab_gl.err || exit 1
    python - "$lib" >> gpurun_out/r03_ab_global2.txt <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_gl.json").read().strip().splitlines()[-1])
e = json.loads(open("gpurun_out/ab_gl2.json").read().strip().splitlines()[-1])
print("%-36s 20-step %6d  512-step %6d (expand launch %.2f ms)" % (sys.argv[1], round(d["value"]), round(e["value"]), e["roofline"]["avg_launch_ms"]))
PY
  done
done
cat gpurun_out/r03_ab_global2.txt
