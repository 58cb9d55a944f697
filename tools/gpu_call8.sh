set -o pipefail
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_consumer_prof -o c -- python3 $GRAFT_REPO_ROOT/tools/shard_rehearsal.py --no-check --shard 16384 > $GRAFT_REPO_ROOT/gpurun_out/r03_consumer_prof.log 2>&1 )
python tools/timeline.py $(ls gpurun_out/r03_consumer_prof/*kernel_trace.csv | head -1) > gpurun_out/r03_consumer_timeline.txt
grep -E "k_sha_expand|k_digest|k_place" gpurun_out/r03_consumer_timeline.txt | head -70; tail -1 gpurun_out/r03_consumer_prof.log
