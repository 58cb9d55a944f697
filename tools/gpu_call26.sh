#!/bin/bash
# A/B in one call: six-lane programs with by-value operands / cursor copies in their out-of-line parts (build/libblsw_before.so = before)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_bit_exact or grouped_batches or multi_small or multi_grouped or params or mode_options or aggregate_verify_reference" > gpurun_out/r03_gputest26.log 2>&1
rc=$?
tail -5 gpurun_out/r03_gputest26.log
[ $rc -eq 0 ] || exit $rc
: > gpurun_out/r03_ab_team_byvalue.txt
for round in 1 2 3; do
  for lib in build/libblsw_before.so bls-verify-gadget_amd/libblsw.so; do
    BLSW_LIB=$PWD/$lib timeout -k 10 120 python tools/chains_only.py > gpurun_out/ab_c.json 2> gpurun_out/ab.err || exit 1
    BLSW_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>> gpurun_out/ab.err || exit 1
    python - "$lib" >> gpurun_out/r03_ab_team_byvalue.txt <<'PY'
import json, sys
c = json.loads(open("gpurun_out/ab_c.json").read().strip().splitlines()[-1])
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("%-36s chains-only %6d  20-step %6d  consumer 8192 %6d  steady 32768 %6d  %s" % (sys.argv[1], round(c["instances_per_s"]), round(d["value"]), round(d["value_consumer_mode"]), round(d["value_consumer_mode_steady"]), d["witness_ok"]))
PY
  done
done
cat gpurun_out/r03_ab_team_byvalue.txt
