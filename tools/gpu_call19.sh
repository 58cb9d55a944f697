#!/bin/bash
# A/B in one call: prepare(sig) + G1 behind prepare(H) on the main stream for small groups (BLSW_BALANCE_MAX_LANES 0 / 8192 / 16384)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_ab_balance.txt
for round in 1 2; do
  for lib in build/libblsw_nobal.so bls-verify-gadget_amd/libblsw.so build/libblsw_balprep.so build/libblsw_bal16k.so; do
    BLSW_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/ab_bal.json 2> gpurun_out/ab_bal.err || exit 1
    python - "$lib" >> gpurun_out/r03_ab_balance.txt <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_bal.json").read().strip().splitlines()[-1])
print("%-36s 20-step %6d  consumer 8192 %6d  steady 32768 %6d  %s" % (sys.argv[1], round(d["value"]), round(d["value_consumer_mode"]), round(d["value_consumer_mode_steady"]), d["witness_ok"]))
PY
  done
done
cat gpurun_out/r03_ab_balance.txt
