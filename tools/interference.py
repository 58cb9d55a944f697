#!/usr/bin/env python3
"""Which chain kernel slows the placement kernel down? Least-squares fit of every k_sha_expand launch's duration on the
average number of concurrently running kernels of each chain kind, from a rocprofv3 kernel trace CSV
(tools/prof_final.sh writes gpurun_out/prof_final/final_kernel_trace.csv)."""
import csv
import sys

import numpy as np

KINDS = ["k_cofactor", "k_sha_expand", "k_pairing_team", "k_g2_alloc", "k_map", "k_sha_values", "k_sha(", "k_prepare", "k_place_field", "k_g1(", "k_sign"]


def short(n):
    for k in KINDS:
        if k in n:
            return k.rstrip("(")
    return None


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if short(r["Kernel_Name"])]
    chains = [e for e in ev if e[0] not in ("k_sha_expand", "k_place_field", "k_sign")]
    kinds = sorted({e[0] for e in chains})
    X, y = [], []
    for k, s, e in ev:
        if k != "k_sha_expand":
            continue
        X.append([sum(max(0, min(e, ce) - max(s, cs)) for kk, cs, ce in chains if kk == kind) / (e - s) for kind in kinds] + [1.0])
        y.append((e - s) / 1e6)
    X, y = np.array(X), np.array(y)
    n = len(y)
    sl = slice(n // 10, n - n // 10)  # without pipeline fill and drain
    coef = np.linalg.lstsq(X[sl], y[sl], rcond=None)[0]
    print("k_sha_expand launches %d, mean %.2f ms" % (n, y[sl].mean()))
    for kind, c, m in zip(kinds + ["(alone)"], coef, X[sl].mean(0)):
        print("%-16s %+6.2f ms per running kernel x %.2f running on average = %+5.2f ms" % (kind, c, m, c * m))


if __name__ == "__main__":
    main()
