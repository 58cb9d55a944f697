#!/bin/bash
# EXPERIMENT A/B in one call: chunked cofactor loops at two waves per SIMD with LDS-parked accumulator (build/libblsw_coflds.so, -DBLSW_COFACTOR_LDS)
set -o pipefail
mkdir -p gpurun_out
BLSW_LIB=$PWD/build/libblsw_coflds.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kernel_variants or batch_bit_exact or grouped_batches or consumer_mode or edge_inputs or multi_small" > gpurun_out/r03_gputest33.log 2>&1
rc=$?
tail -5 gpurun_out/r03_gputest33.log
[ $rc -eq 0 ] || exit $rc
: > gpurun_out/r03_ab_coflds.txt
for round in 1 2; do
  for lib in bls-verify-gadget_amd/libblsw.so build/libblsw_coflds.so; do
    BLSW_LIB=$PWD/$lib BLSW_COFACTOR_MODE=2 timeout -k 10 120 python tools/chains_only.py > gpurun_out/ab_c2.json 2> gpurun_out/ab.err || exit 1
    BLSW_LIB=$PWD/$lib timeout -k 10 120 python tools/chains_only.py --coalesce 4 --steps 48 --warmup 24 > gpurun_out/ab_c4.json 2>> gpurun_out/ab.err || exit 1
    BLSW_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>> gpurun_out/ab.err || exit 1
    python - "$lib" >> gpurun_out/r03_ab_coflds.txt <<'PY'
import json, sys
c2 = json.loads(open("gpurun_out/ab_c2.json").read().strip().splitlines()[-1])
c4 = json.loads(open("gpurun_out/ab_c4.json").read().strip().splitlines()[-1])
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("%-36s chains-only(groups of 10, chunked forced) %6d  chains-only(groups of 4) %6d  20-step %6d  consumer 8192 %6d  steady 32768 %6d  %s" % (sys.argv[1], round(c2["instances_per_s"]), round(c4["instances_per_s"]), round(d["value"]), round(d["value_consumer_mode"]), round(d["value_consumer_mode_steady"]), d["witness_ok"]))
PY
  done
done
cat gpurun_out/r03_ab_coflds.txt
