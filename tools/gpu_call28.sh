#!/bin/bash
# kernel timeline of the consumer-mode legs (8 192-instance shard and 32 768-instance steady leg) of the bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03cons
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cd $R
python3 tools/timeline.py $(ls $O/trace/*kernel_trace.csv $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/timeline.txt
wc -l $O/timeline.txt
