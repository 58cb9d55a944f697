#!/bin/bash
# round 4, GPU call 4: consumer legs after the stream-priority fix (ramp on / off), GPU_MAX_HW_QUEUES A/B, then the whole GPU test suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/experiments/r04_consumer_probe.py > gpurun_out/r04_consumer_probe2.txt 2> gpurun_out/r04_consumer_probe2.err || { tail -5 gpurun_out/r04_consumer_probe2.err; exit 1; }
cat gpurun_out/r04_consumer_probe2.txt
for q in "" 8; do
  for rep in 1 2; do
    if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab_line.json 2> gpurun_out/ab_line.err || { tail -5 gpurun_out/ab_line.err; exit 1; }
    python - "GPU_MAX_HW_QUEUES=${q:-default}" >> gpurun_out/r04_hwq.txt <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read().strip().splitlines()[-1])
print("%-28s 20-step value %7d  ms/step %6.3f  k_sha_expand %6.3f ms  consumer %6s steady %6s  witness_ok %s" % (sys.argv[1], round(d["value"]), d["ms_per_step"], d["roofline"]["avg_launch_ms"],
      round(d.get("value_consumer_mode") or 0), round(d.get("value_consumer_mode_steady") or 0), d["witness_ok"]))
PY
    tail -1 gpurun_out/r04_hwq.txt
  done
done
unset GPU_MAX_HW_QUEUES
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest1.log 2>&1
rc=$?
tail -5 gpurun_out/r04_gputest1.log
exit $rc
