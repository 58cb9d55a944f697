#!/bin/bash
# kernel durations of one launch group running alone (warm-up group of 16 steps, then one timed group)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_alone
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_alone -o p -- python3 $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline > $R/gpurun_out/prof_alone.log 2>&1
python3 - <<PY
import sqlite3,glob
c=sqlite3.connect(glob.glob("$R/gpurun_out/prof_alone/*.db")[0])
t0=None
for n,s,e in c.execute("select name,start,end from kernels order by start"):
    if "k_" in n and "expand" not in n and "place" not in n and "bench" not in n:
        t0 = t0 or s
        print("%-16s start %9.2f ms  duration %8.2f ms"%(n.split("(anonymous namespace)::")[1].split("(")[0][:16],(s-t0)/1e6,(e-s)/1e6))
PY
