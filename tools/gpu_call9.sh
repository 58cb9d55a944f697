set -o pipefail
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi" > gpurun_out/r03_gputest9.log 2>&1; tail -12 gpurun_out/r03_gputest9.log
python tools/bench_configs.py --hash-n 262144 --agg-n 64 --agg-steps 3 > gpurun_out/r03_side_configs_c.jsonl 2> gpurun_out/r03_side_configs_c.err; tail -3 gpurun_out/r03_side_configs_c.jsonl | cut -c1-400; tail -3 gpurun_out/r03_side_configs_c.err
