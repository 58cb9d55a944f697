#!/bin/bash
cd $GRAFT_REPO_ROOT
i=0
for v in "BLSW_PRIO_MODE=0" "BLSW_PRIO_MODE=1" "BLSW_PRIO_MODE=2" "BLSW_PRIO_MODE=1 BLSW_EXPAND_NT=1"; do
  i=$((i+1))
  env $v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 256 --warmup 32 > gpurun_out/env_$i.log 2>&1
  echo "== $v exit $?" >> gpurun_out/env_summary.txt
  grep "^{" gpurun_out/env_$i.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" >> gpurun_out/env_summary.txt 2>&1
done
true
