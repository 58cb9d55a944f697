#!/bin/bash
cd $GRAFT_REPO_ROOT
i=0
for v in "--coalesce 12 --buffers 4" "--coalesce 10 --buffers 5" "--coalesce 8 --buffers 6" "--coalesce 14 --buffers 3" "--coalesce 16 --buffers 3 --outputs 3"; do
  i=$((i+1))
  timeout -k 10 200 python bench.py --no-cpu-baseline $v > gpurun_out/env_$i.log 2>&1
  echo "== $v exit $?" >> gpurun_out/env_summary.txt
  grep "^{" gpurun_out/env_$i.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['batches_fused_per_launch_group'], d['config']['groups_in_flight'])" >> gpurun_out/env_summary.txt 2>&1
done
true
