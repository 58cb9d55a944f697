set -o pipefail
one() { # name lib args...
  name=$1; lib=$2; shift 2
  if [ "$lib" = default ]; then unset BLSW_LIB; else export BLSW_LIB=$PWD/$lib; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --allgather-steps 0 --consumer-shard 0 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', '$*', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), d['config']['results_ok'], d['witness_ok'])" | tee -a gpurun_out/r03_ab7.txt
}
for rep in 1 2 3 4; do
  one aux1 default --steps 20 --warmup 5
  one aux2 build/libblsw_aux2.so --steps 20 --warmup 5
done
for rep in 1 2; do
  one aux1 default --steps 256 --warmup 48
  one aux2 build/libblsw_aux2.so --steps 256 --warmup 48
done
for lib in default build/libblsw_aux2.so default build/libblsw_aux2.so; do
  if [ "$lib" = default ]; then unset BLSW_LIB; else export BLSW_LIB=$PWD/$lib; fi
  for sh in 8192 32768; do
  python tools/shard_rehearsal.py --no-check --shard $sh | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('consumer shard $sh $lib', round(d['instances_per_s']), d['results_ok'])" | tee -a gpurun_out/r03_ab7.txt
  done
done
