#!/usr/bin/env python3
"""From a tools/timeline.py listing of `tools/consumer_probe.py "[8192,null,false]" ...` under rocprofv3 --kernel-trace: the LAST 8 192-instance shard
(first kernel of its first launch group to its last digest), times relative to the group's first kernel. Usage: consumer_timeline_excerpt.py <listing>"""
import sys

rows = []
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) < 7 or not f[0].replace(".", "").isdigit():
        continue
    rows.append((float(f[0]), float(f[1]), float(f[2]), line.rstrip("\n")))
starts = [i for i, r in enumerate(rows) if " k_map_q " in r[3] and r[3].split()[-1] == "32768"]  # 4 x 1024 instances x 2 field elements x 4 lanes
if not starts:
    sys.exit("no latency group of 4 x 1024 instances in the listing")
i0 = starts[-1]
while i0 > 0 and rows[i0][0] - rows[i0 - 1][0] < 2.0 and " k_sign " not in rows[i0 - 1][3]:
    i0 -= 1
t0 = rows[i0][0]
print("# one 8 192-instance consumer-mode shard (ring of two tensors, groups of 4 steps) under rocprofv3 --kernel-trace: ms relative to its first kernel.")
print("# First group = latency kernels (k_map_q, k_cofv_*: the cofactor chain in segments, its points / additions on other streams, k_prepv_*), its")
print("# expansions after k_cofv_join; second group = throughput kernels. The profiler changes the host's timing: see the group-trace file beside this one.")
print("#  start_ms     end_ms    dur_ms  kernel                       queue stream grid_x")
for t, e, d, line in rows[i0:]:
    if t - t0 > 400.0:
        break
    f = line.split(None, 3)
    print("%10.2f %10.2f %9.2f  %s" % (t - t0, e - t0, d, f[3]))
