#!/bin/bash
# after batching the placement kernels' loads: GPU suite, then bench 20-step x2 + 256-step x2 + consumer probe, kernel stats of a 48-step run
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest5.log 2>&1
rc=$?
tail -3 gpurun_out/r04_gputest5.log
[ $rc -eq 0 ] || exit $rc
for args in "--steps 20 --warmup 5" "--steps 256 --warmup 48" "--steps 20 --warmup 5" "--steps 256 --warmup 48"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $args > gpurun_out/ab_line.json 2> gpurun_out/ab_line.err || { tail -5 gpurun_out/ab_line.err; exit 1; }
  python - "$args" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read().strip().splitlines()[-1])
print("%-24s value %7d  ms/step %6.3f  k_sha_expand %6.3f ms  consumer %6s steady %6s  witness_ok %s" % (sys.argv[1], round(d["value"]), d["ms_per_step"], d["roofline"]["avg_launch_ms"],
      round(d.get("value_consumer_mode") or 0), round(d.get("value_consumer_mode_steady") or 0), d["witness_ok"]))
PY
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/place_stats -o s -- python3 $R/bench.py --no-cpu-baseline --consumer-shard 0 --steps 96 --warmup 32 > $R/gpurun_out/place_stats.json 2> $R/gpurun_out/place_stats.err || exit 1
f=$(ls $R/gpurun_out/place_stats/*kernel_stats.csv $R/gpurun_out/place_stats/*/*kernel_stats.csv 2>/dev/null | head -1)
grep -E "k_place_field|k_sha_expand|Name" $f | cut -c1-160
rm -rf $R/gpurun_out/place_stats
