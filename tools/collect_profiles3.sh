#!/bin/bash
# after `gpurun -- bash tools/prof_round3.sh`: copies the summaries the judge reads from gpurun_out/r03prof into profiles/ (tracked)
O=gpurun_out/r03prof
P=profiles
cp $O/bench_default.json $P/r03_bench_default.json
cp $O/bench_20_5.json $P/r03_bench_20_5.json
cp $(ls $O/stats/*kernel_stats.csv $O/stats/*/*kernel_stats.csv 2>/dev/null | head -1) $P/r03_kernel_stats_bench.csv
grep -v '^{' $O/pmc_hbm.txt > $P/r03_pmc_hbm_traffic.txt; tail -1 $O/pmc_hbm.txt >> $P/r03_pmc_hbm_traffic.txt
cp $O/pmc_sq.txt $P/r03_pmc_sq.txt
cp $O/pmc_l2.txt $P/r03_pmc_l2.txt
cp $O/alone_timeline.txt $P/r03_group_timeline.txt
cp $O/short_timeline.txt $P/r03_short_job_timeline.txt
[ -s $O/side_configs.jsonl ] && cp $O/side_configs.jsonl $P/r03_side_configs.jsonl
[ -s $O/engine_bench_c.json ] && cp $O/engine_bench_c.json $P/r03_engine_bench_c.json
python3 - <<'PY'
import json
line = [l for l in open("gpurun_out/r03prof/pmc_hbm.txt") if l.startswith("{")][-1]
d = json.loads(line)
d["_comment"] = json.load(open("profiles/r02_traffic.json"))["_comment"].replace("prof_round2", "prof_round3")
import re
rows = []
for l in open("profiles/r03_pmc_hbm_traffic.txt"):
    m = re.match(r"(\w+)\s+(\S.*?)\s+launches\s+(\d+)\s+per-launch\s+(\S+)", l)
    if m:
        rows.append((m.group(1), m.group(2).strip(), int(m.group(3)), float(m.group(4))))
steps = [r for r in rows if r[0] == "WRITE_SIZE" and r[1].startswith("k_sha_expand")][0][2]
# kernels of the timed path only: not the micro-benchmarks, the signer that mints the inputs, nor the witness_ok check after the timed
# region (digest kernel + a direct-mode engine = the *_inl compilation)
prod = lambda name: name.startswith("k_") and not name.startswith(("k_bench", "k_sign", "k_digest")) and not name.endswith(("_inl", "_values")) or name.startswith("k_sha_values")
w = sum(r[2] * r[3] for r in rows if r[0] == "WRITE_SIZE" and prod(r[1])) * 1024 / steps
f = sum(r[2] * r[3] for r in rows if r[0] == "FETCH_SIZE" and prod(r[1])) * 1024 / steps
d["whole_step"] = {"instances_per_step": 1024, "steps_profiled": steps, "write_bytes": w, "fetch_bytes_raw_counter": f, "fetch_bytes": 2 * f,
                   "comment": "all product kernels of the profiled run (chains, SHA bits, expansion, placement), per 1024-instance step: sum over kernels of per-launch bytes x launches / steps; FETCH_SIZE doubled as for k_sha_expand"}
json.dump(d, open("profiles/r03_traffic.json", "w"), indent=1)
print(d["k_sha_expand"])
PY
