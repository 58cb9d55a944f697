#!/bin/bash
cd $GRAFT_REPO_ROOT
P="bls-verify-gadget_amd"
cp $P/libblsw.so $P/libblsw_orig.so
for v in i16_u16 i32_u32 i64_u16 i8_u8 orig; do
  cp $P/libblsw_$v.so $P/libblsw.so
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --coalesce 1 --buffers 1 > gpurun_out/env_$v.log 2>&1
  echo "== $v alone exit $?" >> gpurun_out/env_summary.txt
  grep "^{" gpurun_out/env_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_ms'], d['roofline']['achieved'])" >> gpurun_out/env_summary.txt 2>&1
  timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/env_$v.log 2>&1
  grep "^{" gpurun_out/env_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('full', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> gpurun_out/env_summary.txt 2>&1
done
cp $P/libblsw_orig.so $P/libblsw.so
true
