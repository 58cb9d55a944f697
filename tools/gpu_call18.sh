#!/bin/bash
# A/B in one call: six-lane G2 allocation (options.g2_mode = team) against the default, short job and consumer legs
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_ab_g2team.txt
for round in 1 2; do
  for g2 in lane team; do
    BLSW_G2=$g2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/ab_g2.json 2> gpurun_out/ab_g2.err || exit 1
    python - "$g2" >> gpurun_out/r03_ab_g2team.txt <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_g2.json").read().strip().splitlines()[-1])
print("g2", sys.argv[1], "20-step", round(d["value"]), "consumer 8192", round(d["value_consumer_mode"]), "steady 32768", round(d["value_consumer_mode_steady"]), d["witness_ok"])
PY
  done
done
cat gpurun_out/r03_ab_g2team.txt
