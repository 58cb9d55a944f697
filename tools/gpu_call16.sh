#!/bin/bash
# final validation of the round-3 tree: whole GPU suite in one process, smoke(), the two bench lines
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest_final.log 2>&1
rc=$?
tail -3 gpurun_out/r03_gputest_final.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 &&
timeout -k 10 300 python bench.py > gpurun_out/r03_bench_final_default.json 2> gpurun_out/r03_bench_final_default.err &&
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_final_20_5.json 2> gpurun_out/r03_bench_final_20_5.err &&
cat gpurun_out/r03_bench_final_default.json gpurun_out/r03_bench_final_20_5.json
