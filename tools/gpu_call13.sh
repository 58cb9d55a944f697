set -o pipefail
BLSW_TEST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 tests/rccl_worker.py > gpurun_out/r03_worker_gloo2.json 2> gpurun_out/r03_worker_gloo2.err; echo rc=$?; cat gpurun_out/r03_worker_gloo2.json; tail -5 gpurun_out/r03_worker_gloo2.err
