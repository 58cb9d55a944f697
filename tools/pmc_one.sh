#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc1 -o pmc -- python3 $R/bench.py --steps 16 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc1.log 2>&1
python3 - <<PY
import csv,collections
a=collections.defaultdict(list)
for r in csv.DictReader(open("$R/gpurun_out/pmc1/pmc_counter_collection.csv")):
    n=r["Kernel_Name"]
    if "k_" in n: a[n.split("(anonymous namespace)::")[1].split("(")[0][:16]].append(float(r["Counter_Value"]))
for k,v in a.items(): print(k, len(v), "WRITE_SIZE per launch %.3f GB"%(sum(v)/len(v)*1024/1e9))
PY
