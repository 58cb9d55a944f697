set -o pipefail
one() { # name lib args...
  name=$1; lib=$2; shift 2
  if [ "$lib" = default ]; then unset BLSW_LIB; else export BLSW_LIB=$PWD/$lib; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --allgather-steps 0 --consumer-shard 0 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', '$*', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), d['config']['results_ok'], d['witness_ok'])" | tee -a gpurun_out/r03_ab5.txt
}
for rep in 1 2 3 4 5; do
  one base default --steps 20 --warmup 5
  one sel2 build/libblsw_sel2.so --steps 20 --warmup 5
  one selm build/libblsw_selm.so --steps 20 --warmup 5
done
for rep in 1 2; do
  one base default --steps 256 --warmup 48
  one sel2 build/libblsw_sel2.so --steps 256 --warmup 48
  one selm build/libblsw_selm.so --steps 256 --warmup 48
done
unset BLSW_LIB
for ov in 1 0 1 0; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --consumer-shard 0 --allgather-overlap $ov 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('allgather overlap $ov', round(d['value']), round(d['value_with_allgather']))" | tee -a gpurun_out/r03_ab5.txt
done
for cv in 0 2 0 2; do
  BLSW_CHAIN_VARIANT=$cv python tools/shard_rehearsal.py --no-check --shard 8192 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('consumer shard 8192 chain_variant $cv', round(d['instances_per_s']), d['results_ok'])" | tee -a gpurun_out/r03_ab5.txt
  BLSW_CHAIN_VARIANT=$cv python tools/shard_rehearsal.py --no-check --shard 32768 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('consumer shard 32768 chain_variant $cv', round(d['instances_per_s']), d['results_ok'])" | tee -a gpurun_out/r03_ab5.txt
done
