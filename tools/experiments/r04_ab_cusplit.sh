#!/bin/bash
# Round 4: the chip partitioned between streaming and chain kernels (options.cu_split) x expansion geometry, A/B inside one gpurun call.
# usage: tools/experiments/r04_ab_cusplit.sh out_name "E variant" ["E variant" ...]
out=gpurun_out/$1; shift
mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg
  for args in "--steps 256 --warmup 48" "--steps 20 --warmup 5"; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --consumer-shard 0 --expand-variant $2 $args > gpurun_out/ab_line.json 2> gpurun_out/ab_line.err || { tail -5 gpurun_out/ab_line.err; exit 1; }
    python - "$1" "$2" "$args" >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read().strip().splitlines()[-1])
print("cu_split %-2s expand_variant %-2s %-22s value %7d  ms/step %6.3f  k_sha_expand %6.3f ms (%4.2f of 8 TB/s)  witness_ok %s results_ok %s" % (
    sys.argv[1], sys.argv[2], sys.argv[3], round(d["value"]), d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["witness_ok"], d["config"]["results_ok"]))
PY
    tail -1 $out
  done
done
