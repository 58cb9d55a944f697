#!/bin/bash
# kernel timelines of the 8 192-instance consumer-mode shard with the group ramp on and off (rocprofv3 --kernel-trace only)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_ctrace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ramp in true false; do
  rocprofv3 --kernel-trace --output-format csv -d $O/$ramp -o t -- python3 $R/tools/experiments/r04_consumer_probe.py "[8192, null, $ramp]" > $O/$ramp.json 2> $O/$ramp.err || { tail -5 $O/$ramp.err; exit 1; }
  cat $O/$ramp.json
  f=$(ls $O/$ramp/*kernel_trace.csv $O/$ramp/*/*kernel_trace.csv 2>/dev/null | head -1)
  python3 $R/tools/timeline.py $f > $O/timeline_$ramp.txt
  rm -rf $O/$ramp
done
