#!/bin/bash
# round 4, GPU call 5: the driver-style 20-step job under {cofactor chunked forced, expansion variant 8, three smaller groups}; configs[3] legs
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r04_short_job_ab.txt
for rep in 1 2; do
for cfg in "0 0 10" "0 8 10" "2 0 10" "2 8 10" "0 0 7" "2 8 7" "2 0 7" "0 8 7"; do
  set -- $cfg
  BLSW_COFACTOR_MODE=$1 BLSW_EXPAND_VARIANT=$2 timeout -k 10 300 python bench.py --no-cpu-baseline --consumer-shard 0 --steps 20 --warmup 5 --coalesce $3 > gpurun_out/ab_line.json 2> gpurun_out/ab_line.err || { tail -5 gpurun_out/ab_line.err; exit 1; }
  python - "$1" "$2" "$3" >> gpurun_out/r04_short_job_ab.txt <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read().strip().splitlines()[-1])
print("cofactor_mode %s expand_variant %s coalesce<=%-2s (groups of %d)  value %7d  ms/step %6.3f  k_sha_expand %6.3f ms  witness_ok %s" % (
    sys.argv[1], sys.argv[2], sys.argv[3], d["config"]["batches_fused_per_launch_group"], round(d["value"]), d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["witness_ok"]))
PY
  tail -1 gpurun_out/r04_short_job_ab.txt
done
done
timeout -k 10 600 python tools/bench_configs.py > gpurun_out/r04_side_configs.jsonl 2> gpurun_out/r04_side_configs.err || { tail -5 gpurun_out/r04_side_configs.err; exit 1; }
cut -c1-400 gpurun_out/r04_side_configs.jsonl
