#!/bin/bash
# A/B in one call: witness / workspace accesses as GLOBAL instead of FLAT memory instructions (build/libblsw_flat.so = before)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_bit_exact or grouped_batches or kernel_variants or multi_small or aggregate_verify_reference or params or hash_to_g2_batch" > gpurun_out/r03_gputest24.log 2>&1
rc=$?
tail -5 gpurun_out/r03_gputest24.log
[ $rc -eq 0 ] || exit $rc
: > gpurun_out/r03_ab_global.txt
for round in 1 2; do
  for lib in build/libblsw_flat.so bls-verify-gadget_amd/libblsw.so; do
    BLSW_LIB=$PWD/$lib timeout -k 10 120 python tools/chains_only.py > gpurun_out/ab_gl_c.json 2> gpurun_out/ab_gl.err || exit 1
    BLSW_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_gl.json 2>> gpurun_out/ab_gl.err || exit 1
    BLSW_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 256 --warmup 48 --no-cpu-baseline --consumer-shard 0 > gpurun_out/ab_gl2.json 2>> gpurun_out/ab_gl.err || exit 1
    python - "$lib" >> gpurun_out/r03_ab_global.txt <<'PY'
import json, sys
c = json.loads(open("gpurun_out/ab_gl_c.json").read().strip().splitlines()[-1])
d = json.loads(open("gpurun_out/ab_gl.json").read().strip().splitlines()[-1])
e = json.loads(open("gpurun_out/ab_gl2.json").read().strip().splitlines()[-1])
print("%-36s chains-only %6d  20-step %6d  consumer 8192 %6d  steady 32768 %6d  256-step %6d  %s %s" % (sys.argv[1], round(c["instances_per_s"]), round(d["value"]), round(d["value_consumer_mode"]), round(d["value_consumer_mode_steady"]), round(e["value"]), d["witness_ok"], e["witness_ok"]))
PY
  done
done
cat gpurun_out/r03_ab_global.txt
