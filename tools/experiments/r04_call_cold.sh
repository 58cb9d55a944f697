#!/bin/bash
# after the cold-small-group rule: consumer legs (default options) twice, the multi engine legs, bench 20-step twice; then the GPU test suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/experiments/r04_consumer_probe.py "[8192, null, false]" "[8192, null, false]" "[32768, null, false]" "[8192, null, false]" "[32768, null, false]" > gpurun_out/r04_consumer_cold.txt 2> gpurun_out/r04_consumer_cold.err || { tail -5 gpurun_out/r04_consumer_cold.err; exit 1; }
cat gpurun_out/r04_consumer_cold.txt
for rep in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab_line.json 2> gpurun_out/ab_line.err || { tail -5 gpurun_out/ab_line.err; exit 1; }
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/ab_line.json").read().strip().splitlines()[-1])
print("20-step value %7d  ms/step %6.3f  k_sha_expand %6.3f ms  consumer %6s steady %6s  witness_ok %s" % (round(d["value"]), d["ms_per_step"], d["roofline"]["avg_launch_ms"],
      round(d.get("value_consumer_mode") or 0), round(d.get("value_consumer_mode_steady") or 0), d["witness_ok"]))
PY
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest3.log 2>&1
rc=$?
tail -4 gpurun_out/r04_gputest3.log
exit $rc
