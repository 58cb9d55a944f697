#!/bin/bash
# ParametersVar allocated as witnesses: the new GPU test, then the tests closest to what the change touched, then the two bench lines (k_g1 changed)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "params or batch_bit_exact or grouped_batches or mode_options or c_caller or compact_form or satisfies" > gpurun_out/r03_gputest17.log 2>&1
rc=$?
tail -15 gpurun_out/r03_gputest17.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench17_20_5.json 2> gpurun_out/r03_bench17_20_5.err &&
timeout -k 10 300 python bench.py --steps 256 --warmup 48 > gpurun_out/r03_bench17_256.json 2> gpurun_out/r03_bench17_256.err &&
python - <<'PY'
import json
for f in ("gpurun_out/r03_bench17_20_5.json", "gpurun_out/r03_bench17_256.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d["witness_ok"])
PY
