#!/usr/bin/env python3
"""Kernel timeline of a `rocprofv3 --kernel-trace --output-format csv` run: start, end, duration, kernel, queue, stream.
usage: tools/timeline.py <kernel_trace.csv> [from_ms [to_ms]]   (times relative to the first kernel of the trace)"""
import csv
import re
import sys


def load(path):
    rows = list(csv.DictReader(open(path)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    ev = []
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("blsw::", "")
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        ev.append(((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, name, r["Queue_Id"], r["Stream_Id"], r["Grid_Size_X"]))
    ev.sort()
    return ev


def main():
    ev = load(sys.argv[1])
    lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
    print("%10s %10s %9s  %-28s %5s %6s %s" % ("start_ms", "end_ms", "dur_ms", "kernel", "queue", "stream", "grid_x"))
    for s, e, n, q, st, g in ev:
        if lo <= s <= hi:
            print("%10.2f %10.2f %9.2f  %-28s %5s %6s %s" % (s, e, e - s, n[:28], q, st, g))


if __name__ == "__main__":
    main()
