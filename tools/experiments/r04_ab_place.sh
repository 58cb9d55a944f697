#!/bin/bash
# A/B inside one call: the placement kernels with all their gathers in flight (default build) against one load at a time (build/libblsw_place_serial.so,
# -DBLSW_LAB_PLACE_SERIAL), interleaved twice: 256-step run, 20-step job, both with the consumer legs
mkdir -p gpurun_out
: > gpurun_out/r04_ab_place.txt
for rep in 1 2; do
for lib in default build/libblsw_place_serial.so; do
  if [ "$lib" = default ]; then unset BLSW_LIB; else export BLSW_LIB=$PWD/$lib; fi
  for args in "--steps 256 --warmup 48" "--steps 20 --warmup 5"; do
    timeout -k 10 300 python bench.py --no-cpu-baseline $args > gpurun_out/ab_line.json 2> gpurun_out/ab_line.err || { tail -5 gpurun_out/ab_line.err; exit 1; }
    python - "$lib" "$args" >> gpurun_out/r04_ab_place.txt <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read().strip().splitlines()[-1])
print("%-34s %-24s value %7d  ms/step %6.3f  k_sha_expand %6.3f ms  consumer %6s steady %6s  witness_ok %s" % (sys.argv[1], sys.argv[2], round(d["value"]), d["ms_per_step"], d["roofline"]["avg_launch_ms"],
      round(d.get("value_consumer_mode") or 0), round(d.get("value_consumer_mode_steady") or 0), d["witness_ok"]))
PY
    tail -1 gpurun_out/r04_ab_place.txt
  done
done
done
