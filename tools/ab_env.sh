#!/bin/bash
# A/B inside ONE gpurun call (boxes of the pool differ by more than most deltas): the same bench.py command under several environment settings.
# usage: tools/ab_env.sh out_name reps "ENV=.. ENV=.." ["ENV=.." ...] -- bench.py arguments
#   e.g. tools/ab_env.sh ab_latency.txt 2 "BLSW_LATENCY_MODE=0" "BLSW_LATENCY_MODE=1" -- --steps 20 --warmup 5
#        tools/ab_env.sh ab_expand.txt 2 "BLSW_EXPAND_VARIANT=0" "BLSW_EXPAND_VARIANT=8" -- --steps 256 --warmup 48
# (the BLSW_* variables are read by the Python mirror's engine_options(), never by the library). One line per run: instances/s, ms per step,
# k_sha_expand launch average (HIP events) and its fraction of the same-box fill rate, the two consumer-mode legs, witness_ok.
out=gpurun_out/$1; reps=$2; shift 2
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
mkdir -p gpurun_out
for rep in $(seq 1 $reps); do
for e in "${envs[@]}"; do
  env $e timeout -k 10 400 python bench.py --no-cpu-baseline --side-legs 0 "$@" > gpurun_out/ab_line.json 2> gpurun_out/ab_line.err || { tail -5 gpurun_out/ab_line.err; exit 1; }
  python - "$e" "$*" >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("%-36s %-24s value %7d  ms/step %6.3f  k_sha_expand %6.3f ms (%4.2f of 8 TB/s, %4.2f of this box's fill)  consumer %6s steady %6s  witness_ok %s" % (
    sys.argv[1], sys.argv[2], round(d["value"]), d["ms_per_step"], r["avg_launch_ms"], r["frac"], r.get("frac_of_fill") or 0,
    round(d.get("value_consumer_mode") or 0), round(d.get("value_consumer_mode_steady") or 0), d["witness_ok"]))
PY
  tail -1 $out
done
done
