#!/bin/bash
# round 4, GPU call 1: the new tests, the digest kernel alone, then the expansion geometries under load (one call = one box)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_multi_gpu_rccl.py -m gpu -x -q -k "digest_kernel or consumer_mode_small_ring or bench_shape or release_table or batch_bit_exact or bench_two_ranks or cpp_host_mirror_reference or c_caller" > gpurun_out/r04_call1_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r04_call1_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/digest_rate.py > gpurun_out/r04_digest_rate.json 2> gpurun_out/r04_digest_rate.err || { tail -5 gpurun_out/r04_digest_rate.err; exit 1; }
cat gpurun_out/r04_digest_rate.json
: > gpurun_out/r04_ab_expand1.txt
tools/experiments/r04_ab_expand.sh r04_ab_expand1.txt 0 8 9 1 6 7
