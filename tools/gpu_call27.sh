#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cpp_host_mirror or c_caller" > gpurun_out/r03_gputest27.log 2>&1
rc=$?
tail -25 gpurun_out/r03_gputest27.log
exit $rc
