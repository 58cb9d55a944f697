#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/env_$i.log 2>&1
  echo "== run $i exit $?" >> gpurun_out/env_summary.txt
  grep "^{" gpurun_out/env_$i.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['batches_fused_per_launch_group'], d['config']['groups_in_flight'])" >> gpurun_out/env_summary.txt 2>&1
done
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/env_5.log 2>&1
echo "== small K exit $?" >> gpurun_out/env_summary.txt
grep "^{" gpurun_out/env_5.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/env_summary.txt 2>&1
true
