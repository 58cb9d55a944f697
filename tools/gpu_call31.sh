#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "decode or aggregate_points or c_caller or cpp_host or fast_aggregate or sign_batch or verify_fixtures" > gpurun_out/r03_gputest31.log 2>&1
rc=$?
tail -8 gpurun_out/r03_gputest31.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/bench_configs.py > gpurun_out/r03_side_configs_d.jsonl 2> gpurun_out/r03_side_configs_d.err || { tail -5 gpurun_out/r03_side_configs_d.err; exit 1; }
grep -E "decode|aggregate \(|Signature::aggregate|PublicKey::aggregate" gpurun_out/r03_side_configs_d.jsonl | cut -c1-260
