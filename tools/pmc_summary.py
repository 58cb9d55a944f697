#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (counter_collection CSVs): tools/pmc_summary.py <dir> [<dir> ...]
prints `counter kernel launches per-launch-average`; with --json also profiles-style traffic JSON for k_sha_expand
(WRITE_SIZE / FETCH_SIZE are in KiB; FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950, MI355X_MICROARCH.md §HBM)."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("blsw::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)[:34]


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    acc = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[(r["Counter_Name"], short(r["Kernel_Name"]))].append(float(r["Counter_Value"]))
    for (c, k), v in sorted(acc.items()):
        print("%-24s %-36s launches %5d  per-launch %.4e" % (c, k, len(v), sum(v) / len(v)))
    if "--json" in sys.argv:
        w = [v for (c, k), v in acc.items() if c == "WRITE_SIZE" and k.startswith("k_sha_expand")]
        f = [v for (c, k), v in acc.items() if c == "FETCH_SIZE" and k.startswith("k_sha_expand")]
        if w and f:
            w, f = w[0], f[0]
            print(json.dumps({"k_sha_expand": {"instances_per_launch": 1024, "write_kib": sum(w) / len(w), "fetch_kib": 2 * sum(f) / len(f),
                                               "fetch_kib_raw_counter": sum(f) / len(f)}}))


if __name__ == "__main__":
    main()
