set -o pipefail
one() { # name args...
  name=$1; shift
  timeout -k 10 200 python bench.py --no-cpu-baseline --allgather-steps 0 --consumer-shard 0 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', '$*', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), d['config']['batches_fused_per_launch_group'], d['config']['groups_in_flight'], d['witness_ok'])" | tee -a gpurun_out/r03_ab8.txt
}
for rep in 1 2 3; do
  one c10 --steps 20 --warmup 5 --coalesce 10
  one c8 --steps 20 --warmup 5 --coalesce 8
  one c7 --steps 20 --warmup 5 --coalesce 7
  BLSW_COFACTOR_MODE=2 one c10chunk --steps 20 --warmup 5 --coalesce 10
done
for rep in 1 2; do
  one c10 --steps 256 --warmup 48 --coalesce 10
  one c8 --steps 256 --warmup 48 --coalesce 8
  one c12 --steps 256 --warmup 48 --coalesce 12
done
