#!/bin/bash
# after `gpurun -- tools/prof_round.sh <tag>`: copies the summaries the judge reads from gpurun_out/<tag>prof into profiles/ (tracked)
TAG=${1:-r05}
O=gpurun_out/${TAG}prof
P=profiles
cp $O/bench_default.json $P/${TAG}_bench_default.json
cp $O/bench_20_5.json $P/${TAG}_bench_20_5.json
cp $O/kernel_stats.csv $P/${TAG}_kernel_stats_bench.csv
grep -v '^{' $O/pmc_hbm.txt > $P/${TAG}_pmc_hbm_traffic.txt; tail -1 $O/pmc_hbm.txt >> $P/${TAG}_pmc_hbm_traffic.txt
cp $O/pmc_sq.txt $P/${TAG}_pmc_sq.txt
{ echo "# k_digest alone (tools/digest_rate.py): rate from HIP events, then SQ_INSTS_VALU / SQ_WAVES per launch of 1024 instances (rocprofv3 --pmc, own pass)"; cat $O/digest_rate.json; cat $O/pmc_digest.txt; } > $P/${TAG}_digest.txt
cp $O/short_timeline.txt $P/${TAG}_short_job_timeline.txt
[ -s $O/consumer_timeline_full.txt ] && cp $O/consumer_probe.txt $P/${TAG}_consumer_probe.txt
[ -s $O/consumer_timeline_full.txt ] && python3 tools/consumer_timeline_excerpt.py $O/consumer_timeline_full.txt > $P/${TAG}_consumer_timeline.txt
[ -s $O/consumer_group_trace.txt ] && grep "blsw group\|instances_per_s" $O/consumer_group_trace.txt > $P/${TAG}_consumer_group_trace.txt
[ -s $O/verify_rate.txt ] && cp $O/verify_rate.txt $P/${TAG}_verify_rate.txt
[ -s $O/side_configs.jsonl ] && cp $O/side_configs.jsonl $P/${TAG}_side_configs.jsonl
[ -s $O/engine_bench_c.json ] && cp $O/engine_bench_c.json $P/${TAG}_engine_bench_c.json
TAG=$TAG python3 - <<'PY'
import json, os, re
tag = os.environ["TAG"]
line = [l for l in open("gpurun_out/%sprof/pmc_hbm.txt" % tag) if l.startswith("{")][-1]
d = json.loads(line)
d["_comment"] = ("HBM traffic per k_sha_expand launch and per whole step from rocprofv3 --pmc passes of `bench.py --steps 48 --warmup 16` (tools/prof_round.sh): WRITE_SIZE and "
                 "FETCH_SIZE in separate passes, KiB per launch; FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B: MI355X_MICROARCH.md, HBM)")
rows = []
for l in open("profiles/%s_pmc_hbm_traffic.txt" % tag):
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
json.dump(d, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
print(d["k_sha_expand"], d["whole_step"])
PY
