#!/bin/bash
# Round 4, A/B inside ONE gpurun call (box-to-box spread is larger than most deltas): store geometries of the SHA expansion under load.
# usage: tools/experiments/r04_ab_expand.sh out_name variant [variant ...]     (variant = BLSW_EXPAND_VARIANT value)
# per variant: the driver-style short job twice and a 256-step run; prints instances/s, ms per step, k_sha_expand avg launch ms (HIP events),
# the two consumer-mode legs and witness_ok.
out=gpurun_out/$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
for v in "$@"; do
  for args in "--steps 20 --warmup 5" "--steps 256 --warmup 48"; do
    BLSW_EXPAND_VARIANT=$v timeout -k 10 300 python bench.py --no-cpu-baseline $args > gpurun_out/ab_line.json 2> gpurun_out/ab_line.err || { tail -5 gpurun_out/ab_line.err; exit 1; }
    python - "$v" "$args" >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read().strip().splitlines()[-1])
print("variant %-4s %-22s value %7d  ms/step %6.3f  k_sha_expand %6.3f ms (%4.2f of 8 TB/s)  consumer %6s steady %6s  witness_ok %s" % (
    sys.argv[1], sys.argv[2], round(d["value"]), d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"],
    round(d.get("value_consumer_mode") or 0), round(d.get("value_consumer_mode_steady") or 0), d["witness_ok"]))
PY
    tail -1 $out
  done
done
done
