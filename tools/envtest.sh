#!/bin/bash
cd $GRAFT_REPO_ROOT
i=0
for v in "BLSW_EXPAND_NT=1 BLSW_PLACE_LDS=40000" "BLSW_EXPAND_NT=1 BLSW_PLACE_LDS=0" "BLSW_EXPAND_NT=0 BLSW_PLACE_LDS=80000" "BLSW_EXPAND_NT=0 BLSW_PLACE_LDS=40000" "BLSW_EXPAND_NT=0 BLSW_PLACE_LDS=0"; do
  i=$((i+1))
  env $v timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/env_$i.log 2>&1
  echo "== $v exit $?" >> gpurun_out/env_summary.txt
  grep "^{" gpurun_out/env_$i.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['results_ok'])" >> gpurun_out/env_summary.txt 2>&1
done
true
