#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "aggregate_points or decode_and_verify or sign_batch" > gpurun_out/r03_gputest20.log 2>&1
rc=$?
tail -25 gpurun_out/r03_gputest20.log
exit $rc
