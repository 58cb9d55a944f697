set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest5.log 2>&1; tail -3 gpurun_out/r03_gputest5.log
python tools/multi_profile.py 40 128 > gpurun_out/r03_multi_40.json 2>&1; tail -1 gpurun_out/r03_multi_40.json
python tools/multi_profile.py 48 128 > gpurun_out/r03_multi_48.json 2>&1; tail -1 gpurun_out/r03_multi_48.json
python tools/multi_profile.py 1 128 > gpurun_out/r03_multi_1.json 2>&1; tail -1 gpurun_out/r03_multi_1.json
for v in 0 2 0 2; do
  for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 256 --warmup 48"; do
    BLSW_CHAIN_VARIANT=$v timeout -k 10 200 python bench.py --no-cpu-baseline --allgather-steps 0 --consumer-shard 0 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('chain_variant $v', '$args', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), d['config']['results_ok'], d['witness_ok'])" | tee -a gpurun_out/r03_ab4.txt
  done
done
python tools/bench_configs.py > gpurun_out/r03_side_configs_b.jsonl 2> gpurun_out/r03_side_configs_b.err; cut -c1-250 gpurun_out/r03_side_configs_b.jsonl
