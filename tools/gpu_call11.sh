set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest11.log 2>&1; tail -3 gpurun_out/r03_gputest11.log
one() { # name lib cofmode args...
  name=$1; lib=$2; cm=$3; shift 3
  if [ "$lib" = default ]; then unset BLSW_LIB; else export BLSW_LIB=$PWD/$lib; fi
  BLSW_COFACTOR_MODE=$cm timeout -k 10 200 python bench.py --no-cpu-baseline --allgather-steps 0 --consumer-shard 0 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', '$*', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), d['config']['results_ok'], d['witness_ok'])" | tee -a gpurun_out/r03_ab6.txt
}
for rep in 1 2 3 4; do
  one serial default 1 --steps 20 --warmup 5
  one chunked default 0 --steps 20 --warmup 5
  one chunked_inl build/libblsw_cofinl.so 0 --steps 20 --warmup 5
done
for rep in 1 2; do
  one serial default 1 --steps 256 --warmup 48
  one chunked default 0 --steps 256 --warmup 48
  one chunked_inl build/libblsw_cofinl.so 0 --steps 256 --warmup 48
done
unset BLSW_LIB
for cm in 1 0 1 0; do
  for sh in 8192 32768; do
  BLSW_COFACTOR_MODE=$cm python tools/shard_rehearsal.py --no-check --shard $sh | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('consumer shard $sh cofactor_mode $cm', round(d['instances_per_s']), d['results_ok'])" | tee -a gpurun_out/r03_ab6.txt
  done
done
python tools/multi_profile.py 48 128 2>/dev/null | tail -1 | tee -a gpurun_out/r03_ab6.txt
python tools/multi_profile.py 1 128 2>/dev/null | tail -1 | tee -a gpurun_out/r03_ab6.txt
BLSW_COFACTOR_MODE=1 python tools/chains_only.py 256 2>/dev/null | tail -1 | tee -a gpurun_out/r03_ab6.txt
BLSW_COFACTOR_MODE=0 python tools/chains_only.py 256 2>/dev/null | tail -1 | tee -a gpurun_out/r03_ab6.txt
