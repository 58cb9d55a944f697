#!/bin/bash
# repeated default bench lines (value, ms/step, expand ms) — run-to-run noise on one box is about +-2 %
for i in 1 2 3; do
  timeout -k 10 250 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), d['config']['results_ok'])" >> gpurun_out/ab.txt
done
