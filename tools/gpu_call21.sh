#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python tools/bench_configs.py > gpurun_out/r03_side_configs_c.jsonl 2> gpurun_out/r03_side_configs_c.err
rc=$?
cut -c1-260 gpurun_out/r03_side_configs_c.jsonl
tail -3 gpurun_out/r03_side_configs_c.err
exit $rc
