set -o pipefail
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hash or sign or c_caller or bytes or smoke" > gpurun_out/r03_gputest2.log 2>&1; tail -5 gpurun_out/r03_gputest2.log
python tools/bench_configs.py > gpurun_out/r03_side_configs_a.jsonl 2> gpurun_out/r03_side_configs_a.err; cut -c1-260 gpurun_out/r03_side_configs_a.jsonl
tools/ab_variants.sh r03_ab2 base=default sel=build/libblsw_sel.so
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_multi_prof -o m -- python3 $GRAFT_REPO_ROOT/tools/multi_profile.py 40 128 > $GRAFT_REPO_ROOT/gpurun_out/r03_multi_prof.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $(ls gpurun_out/r03_multi_prof/*kernel_trace.csv gpurun_out/r03_multi_prof/*/*kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/r03_multi_timeline.txt; tail -30 gpurun_out/r03_multi_timeline.txt; tail -2 gpurun_out/r03_multi_prof.log
