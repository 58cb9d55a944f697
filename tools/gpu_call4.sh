set -o pipefail
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi" > gpurun_out/r03_gputest4.log 2>&1; tail -3 gpurun_out/r03_gputest4.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_multi_prof3 -o m -- python3 $GRAFT_REPO_ROOT/tools/multi_profile.py 40 128 > $GRAFT_REPO_ROOT/gpurun_out/r03_multi_prof3.log 2>&1 )
python tools/timeline.py $(ls gpurun_out/r03_multi_prof3/*kernel_trace.csv | head -1) > gpurun_out/r03_multi_timeline3.txt; grep -v "expand\|copyBuffer\|at::" gpurun_out/r03_multi_timeline3.txt | tail -14; tail -1 gpurun_out/r03_multi_prof3.log
( cd /tmp && export TMPDIR=/tmp && BLSW_LIB=$GRAFT_REPO_ROOT/build/libblsw_inl.so rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_multi_prof4 -o m -- python3 $GRAFT_REPO_ROOT/tools/multi_profile.py 40 128 > $GRAFT_REPO_ROOT/gpurun_out/r03_multi_prof4.log 2>&1 )
python tools/timeline.py $(ls gpurun_out/r03_multi_prof4/*kernel_trace.csv | head -1) > gpurun_out/r03_multi_timeline4.txt; grep -v "expand\|copyBuffer\|at::" gpurun_out/r03_multi_timeline4.txt | tail -14; tail -1 gpurun_out/r03_multi_prof4.log
python tools/multi_profile.py 56 128 > gpurun_out/r03_multi_56.json 2>&1; tail -1 gpurun_out/r03_multi_56.json
BLSW_LIB=$PWD/build/libblsw_inl.so python tools/multi_profile.py 56 128 > gpurun_out/r03_multi_56_inl.json 2>&1; tail -1 gpurun_out/r03_multi_56_inl.json
tools/ab_variants.sh r03_ab3 base=default sel2=build/libblsw_sel2.so sel=build/libblsw_sel.so base2=default
