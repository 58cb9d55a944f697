#!/bin/bash
# round 4, GPU call 6: configs[3] consumer / compact legs alone, then under rocprofv3 --kernel-trace --stats (which kernel takes the time?)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_multi_consumer
mkdir -p $O
cd $R
timeout -k 10 300 python tools/bench_configs.py --only multi-consumer --multi-engine-steps 12 > $O/plain.jsonl 2> $O/plain.err || { tail -5 $O/plain.err; exit 1; }
cut -c1-330 $O/plain.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/tools/bench_configs.py --only multi-consumer --multi-engine-steps 8 > $O/stats.jsonl 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
f=$(ls $O/stats/*kernel_stats.csv $O/stats/*/*kernel_stats.csv 2>/dev/null | head -1)
cp $f $O/kernel_stats.csv
t=$(ls $O/stats/*kernel_trace.csv $O/stats/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/timeline.py $t > $O/timeline.txt
rm -rf $O/stats
head -25 $O/kernel_stats.csv | cut -c1-200
