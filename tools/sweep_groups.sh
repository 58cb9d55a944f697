#!/bin/bash
# group size x groups in flight (same workspace budget): one line per configuration
for cfg in "16 3" "12 4" "8 6" "10 5" "20 2" "16 3"; do
  set -- $cfg
  timeout -k 10 250 python bench.py --no-cpu-baseline --coalesce $1 --buffers $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('coalesce $1 buffers $2 ->', d['config']['groups_in_flight'], round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), d['config']['results_ok'])" >> gpurun_out/sweep_groups.txt || echo "$cfg failed" >> gpurun_out/sweep_groups.txt
done
