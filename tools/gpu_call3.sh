set -o pipefail
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi or padding or rehearsal" > gpurun_out/r03_gputest3.log 2>&1; tail -5 gpurun_out/r03_gputest3.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tests/rccl_worker.py > gpurun_out/r03_rccl_worker1.json 2> gpurun_out/r03_rccl_worker1.err; cat gpurun_out/r03_rccl_worker1.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_multi_prof2 -o m -- python3 $GRAFT_REPO_ROOT/tools/multi_profile.py 40 128 > $GRAFT_REPO_ROOT/gpurun_out/r03_multi_prof2.log 2>&1 )
python tools/timeline.py $(ls gpurun_out/r03_multi_prof2/*kernel_trace.csv gpurun_out/r03_multi_prof2/*/*kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/r03_multi_timeline2.txt; grep -v "expand\|copyBuffer\|at::" gpurun_out/r03_multi_timeline2.txt | tail -16; tail -1 gpurun_out/r03_multi_prof2.log
python tools/multi_profile.py 56 128 > gpurun_out/r03_multi_56.json 2>&1; tail -1 gpurun_out/r03_multi_56.json
python bench.py > gpurun_out/r03_bench_default_a.json 2> gpurun_out/r03_bench_default_a.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_bench_default_a.json"))
print({k:d.get(k) for k in ("value","ms_per_step","witness_ok","value_consumer_mode")}, d.get("consumer_mode"), d["roofline"]["avg_launch_ms"])
PY
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline > gpurun_out/r03_bench_ag1.json 2> gpurun_out/r03_bench_ag1.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03_bench_ag1.json") if l.startswith("{")][-1])
print({k:d.get(k) for k in ("value","value_with_allgather")}, d.get("allgather"))
PY
