#!/bin/bash
# L2 write-side counters of plain fills: one 16-byte store per thread against eight (build/fill_lab) — window or latency?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03fill
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for p in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmc_$i -o pmc -- $R/build/fill_lab > $O/pmc_$i.txt 2> $O/pmc_$i.err || { tail -5 $O/pmc_$i.err; exit 1; }
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r03fill"
acc = collections.defaultdict(list)
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True) + glob.glob(O + "/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].replace("void ", "")[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
ks = sorted({k for k, _ in acc})
print("%-62s %12s %12s %12s %12s %10s %10s" % ("kernel", "WRREQ", "STALL", "LEVEL", "CYCLE", "req/cyc", "latency"))
for k in ks:
    g = lambda c: (sum(acc[(k, c)]) / len(acc[(k, c)])) if acc.get((k, c)) else float("nan")
    w, s, l, c = g("TCC_EA0_WRREQ_sum"), g("TCC_EA0_WRREQ_STALL_sum"), g("TCC_EA0_WRREQ_LEVEL_sum"), g("TCC_CYCLE_sum")
    print("%-62s %12.4g %12.4g %12.4g %12.4g %10.3f %10.1f" % (k, w, s, l, c, w / c, l / w))
PY
