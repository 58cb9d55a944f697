#!/usr/bin/env python3
"""bench.py — BLS-verify witness instances/sec (full pairing circuit) on MI355X.

One "step" = one pass of the hot path (every witness of the circuit of /root/reference/src/constraints.rs:335-366, for
each instance) over one batch of 1 024 synthetic (pk, msg, sig) instances (BASELINE.json configs[1]). Inputs are resident
in HBM before the timed region; every step writes a complete [1024][n_witness] witness tensor (34 MB per instance).

Steps are SUBMITTED to the engine, which fuses up to `--coalesce` pending batches into one group of launches (a single
batch of 1024 instances is 16 wavefronts per chain kernel on a 1024-SIMD chip) and writes every step's witness tensor, in
submission order, into a ring of `--outputs` output tensors. EXACTLY K steps (K full witness tensors) are timed between
barrier + synchronize on both sides.

Multi-GPU: one process per GPU over RCCL. `python bench.py --gpus N` starts the N ranks itself (a child
`python -m torch.distributed.run`, spawned before this process touches the GPU) and prints the child's one JSON line; under
an external launcher (RANK / WORLD_SIZE in the environment) it is a rank and checks WORLD_SIZE == --gpus. Instances are
independent: each rank processes its own 1 024-instance shard per step (weak scaling, NO data-path collective in `value`);
the north-star's all-gather of witness shards is timed separately (`value_with_allgather`). Default `--allgather-form compact`:
every step leaves the engine in its compact wire form (bit-packed SHA witnesses + field witnesses, 2.6 MB per instance), THAT is
all-gathered over RCCL, and every rank expands all ranks' batches back into full witness tensors (34 MB per instance), each
consumed by the digest kernel before the next; `--allgather-form full` gathers the full tensors in micro-batches instead.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The library needs no runtime environment. This harness sets ONE variable for its own process, before HIP is initialised: the runtime backs the
# streams of each priority level with GPU_MAX_HW_QUEUES (default 4) hardware queues, and streams that share a queue serialise. One rank has the
# null stream + up to three group buffers' main streams at normal priority; the all-gather legs add a communication stream and the process
# group's own RCCL stream: with four queues two of them share one, and a marker behind a waiting command starts the next launch group ~70 ms
# late (profiles/r04_consumer_timeline.txt). Eight queues per level remove the aliasing; measured neutral to +1 % at N = 1
# (profiles/r04_consumer_probe.txt, block 3). A value the caller has set is kept.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)
# algorithmic field work per instance, from the oracle's op counter on the reference gadget case
# (tests/golden/oracle_opcount.json): Fp products and Fp inversions; 300 multiply-adds per product, 570 products per
# inversion (Fermat pricing)
MAD_PER_FPMUL = 300
FPMUL_PER_INV = 570
# what the kernels execute instead of a Fermat inversion: one safegcd inversion = 26.9 Fp-product times (blsw_microbench 2 / 1);
# 636 of the 940 cofactor-chain inversions are shared with a neighbour (Montgomery's trick: +9 products each, -1 inversion)
FPMUL_PER_INV_EXECUTED = 27
TRAFFIC_FILES = ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=48)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU per step (configs[1]: 1024)")
    ap.add_argument("--coalesce", type=int, default=10, help="max submitted batches fused into one launch group")
    ap.add_argument("--buffers", type=int, default=3, help="launch groups in flight (each owns streams + a workspace slice)")
    ap.add_argument("--outputs", type=int, default=2, help="ring of output witness tensors (34 MB x batch each)")
    ap.add_argument("--mem-frac", type=float, default=0.68, help="share of the free HBM the engine workspace and the output ring may take")
    ap.add_argument("--expand-variant", type=int, default=-1, help="options.expand_variant of the timed engine (-1 = the library's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=256, help="instances of the batch timed on the host (all cores, and one thread)")
    ap.add_argument("--consumer-shard", type=int, default=8192,
                    help="instances of the consumer-mode leg at N = 1 (configs[2]'s per-GPU shard streamed through a ring of two tensors with the digest kernel reading every one; 0 = skip)")
    ap.add_argument("--allgather-steps", type=int, default=32, help="steps of the generation + all-gather leg (0 = skip; runs when a process group exists)")
    ap.add_argument("--allgather-chunk", type=int, default=64, help="instances per rank in one all-gathered micro-batch (form full)")
    ap.add_argument("--allgather-group", type=int, default=16, help="steps per launch group in the all-gather leg (consumer-mode engine)")
    ap.add_argument("--allgather-overlap", type=int, default=-1,
                    help="compact form: 1 = the gather of step k + 1 on a communication stream beside expansion + digest of step k; 0 = one stream; -1 = 1 when there "
                         "is something to overlap (world > 1): at world 1 the gather is a local copy and the second stream only adds contention (48.5 k vs 44.2 k instances/s)")
    ap.add_argument("--consumer-reps", type=int, default=5, help="runs of the consumer-mode leg (the median is reported)")
    ap.add_argument("--consumer-steady-shard", type=int, default=32768, help="a second, longer consumer-mode run (steady state: HBM-bound rather than chain-latency-bound; 0 = skip)")
    ap.add_argument("--side-legs", type=int, default=1, help="N = 1: bounded legs for BASELINE configs[3] (one signature over 128 pairs), configs[4] (hash-to-G2, 1 M messages) and "
                    "the compact wire form of configs[1], each with its own roofline and CPU baseline (0 = skip)")
    ap.add_argument("--allgather-form", choices=("compact", "full"), default="compact",
                    help="what travels in the all-gather leg: the compact wire form, expanded by every receiver, or the full witness tensors")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N outside a launcher: start N ranks as a CHILD torch.distributed.run (this process has not touched the GPU and
    never will) and relay the one JSON line rank 0 prints."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1])
    else:
        sys.stderr.write(p.stdout)
    sys.exit(p.returncode if p.returncode or lines else 1)


def balanced_coalesce(steps, max_group):
    """Equal launch groups: 25 steps with groups of at most 10 run as 9 + 8 + 8 rather than 10 + 10 + 5."""
    groups = (steps + max_group - 1) // max_group
    return max(1, (steps + groups - 1) // groups)


def host_cores():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (a GPU box grants a share)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(args, d_pk, d_msg, d_sig, n):
    """The C++ restatement of the reference path (oracle/) on the host cores: all cores and one thread, on the first
    `--cpu-sample` instances of the batch. Built -O3 -march=native on THIS machine when a compiler is present."""
    import numpy as np

    from tests import oracle_lib  # the CPU restatement: used for this baseline leg only

    build = "portable (-O3 -march=x86-64-v3 -madx)"
    oracle = None
    try:
        import ctypes

        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        oracle = oracle_lib.Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "_native", "liboracle.so")))
        build = "-O3 -march=native, built on this host"
    except Exception:
        oracle = oracle_lib.load()
    cores = host_cores()
    m = min(max(args.cpu_sample, 8 * cores), n)
    h_pk, h_msg, h_sig = d_pk[:m].cpu().numpy().view(np.uint64), d_msg[:m].cpu().numpy(), d_sig[:m].cpu().numpy().view(np.uint64)
    t1 = time.perf_counter()
    oracle.witness_batch(h_pk[:4], h_msg[:4], h_sig[:4], threads=1, want_digests=False)  # probe: seconds per instance on one thread
    per_inst = (time.perf_counter() - t1) / 4
    m1 = min(m, args.cpu_sample, max(8, int(30.0 / max(per_inst, 1e-9))))  # one thread: the sample, capped at ~30 s of work
    t1 = time.perf_counter()
    oracle.witness_batch(h_pk[:m1], h_msg[:m1], h_sig[:m1], threads=1, want_digests=False)
    one_dt = time.perf_counter() - t1
    t1 = time.perf_counter()
    oracle.witness_batch(h_pk, h_msg, h_sig, threads=cores, want_digests=False)
    all_dt = time.perf_counter() - t1
    # SURVEY 8d config 1 as written: the single instance the reference hard-codes (constraints.rs:337-343 =
    # tests/test_cases/verify/verify_valid_case_2ea479adf8c40300.json), constraint synthesis + witness generation on one thread
    single_ms, single_ok = None, None
    try:
        case = json.load(open(os.path.join(ROOT, "tests", "golden", "ethereum_bls12_381_v0.1.2", "verify", "verify_valid_case_2ea479adf8c40300.json")))
        unhex = oracle_lib.unhex
        pk_xy = oracle.g1_decompress(unhex(case["input"]["pubkey"]))[1]
        sig_xy = oracle.g2_decompress(unhex(case["input"]["signature"]))[1]
        msg = unhex(case["input"]["message"])
        oracle.witness(pk_xy, msg, sig_xy, want_vector=False)  # warm
        reps = 5
        t1 = time.perf_counter()
        for _ in range(reps):
            r = oracle.witness(pk_xy, msg, sig_xy, want_vector=False)
        single_ms = (time.perf_counter() - t1) / reps * 1e3
        single_ok = bool(r[2]) == bool(case["output"])
    except Exception as exc:  # noqa: BLE001 (the baseline line survives a missing fixture)
        single_ok = "%s: %s" % (type(exc).__name__, exc)
    return {"value": m / all_dt, "unit": "instances/s", "cores": cores, "kind": "port", "value_1thread": m1 / one_dt,
            "single_case_ms": single_ms, "single_case": "verify_valid_case_2ea479adf8c40300.json (= constraints.rs:337-343), one thread, mean of 5", "single_case_result_ok": single_ok,
            "sample": "first %d instances of the bench batch through the C++ restatement of the reference path (oracle/, %s), %d threads: %.2f s; "
                      "first %d instances on one thread: %.2f s" % (m, build, cores, all_dt, m1, one_dt)}


def side_legs(args, pkg, workload, dev, fpmul_peak, opc):
    """Bounded legs for the BASELINE configs the headline does not time (N = 1 only), each under this process's clock with a roofline of its own and
    the oracle timed beside it (tools/bench_configs.py holds the longer sweeps):
      multi_128       configs[3]: one signature over 128 (pk, msg) pairs, 129-pair Miller product, through the grouped engine (16 instances per step,
                      groups of 4, three in flight, ring of two 67 GB tensors, free running) — HBM-bound: output bytes / time against 8 TB/s
      hash_to_g2_1M   configs[4]: SSWU + cofactor clearing for 1 M messages (value-only kernels) — VALU-bound: executed Fp products against the measured rate
      compact_form    configs[1] with the steps leaving in the compact wire form (2.6 MB per instance, what a sharded run ships) — VALU-bound
      verify_batch    BLS::verify as values for 65 536 triples from compressed bytes (the native batch verifier) — VALU-bound
    """
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from tests import oracle_lib  # the CPU restatement: baselines of these legs only

    oracle = oracle_lib.load()
    cores = host_cores()
    legs = {}

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # ---- configs[4]: hash-to-G2, 1 M messages in chunks of 262 144; message 0 is the 32-zero-byte message of bls.rs:645
    try:
        total_n, chunk = 1 << 20, 1 << 18
        msgs = workload.messages(0x5EED, 0, chunk, tag=b"h")
        msgs[0] = 0
        d = torch.from_numpy(msgs).to(dev)
        out = torch.empty((chunk, 24), dtype=torch.int64, device=dev)
        pkg.hash_to_g2_batch(d, out=out)
        dt = sum(timed(lambda: pkg.hash_to_g2_batch(d, out=out)) for _ in range(total_n // chunk))
        first = out[0].cpu().numpy().view(np.uint64)
        ok = bool(np.array_equal(first, oracle.hash_to_g2(bytes(32))[1]))
        # executed Fp products per message of the value-only kernels: two maps of ~2.3 k (joint-ladder exponentiation) + ~2.9 k (psi-based clearing)
        fpmul_per_msg = 2 * 2300 + 2900
        sample = msgs[1:1 + 4 * cores]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            list(ex.map(lambda m: oracle.hash_to_g2(m.tobytes()), sample))
        cdt = time.perf_counter() - t0
        legs["hash_to_g2_1M"] = {"workload": "configs[4]: hash-to-G2 (SSWU + cofactor clearing, hasher.rs:727-740 values), 1 048 576 distinct 32-byte messages, affine outputs", "value": total_n / dt,
                                 "unit": "messages/s", "seconds": dt, "first_output_is_bls_rs_645_point": ok,
                                 "roofline": {"bound": "valu-fp-mul", "executed_fpmul_per_message": fpmul_per_msg, "achieved": total_n / dt * fpmul_per_msg, "peak": fpmul_peak,
                                              "unit": "Fp products/s (peak = blsw_microbench 1, measured on this box)", "frac": total_n / dt * fpmul_per_msg / fpmul_peak},
                                 "cpu_baseline": {"value": len(sample) / cdt, "unit": "messages/s", "cores": cores, "kind": "port",
                                                  "sample": "%d messages through the oracle's hash_to_g2 on %d threads: %.2f s" % (len(sample), cores, cdt)}}
        del d, out
    except Exception as exc:  # noqa: BLE001 (reported in the JSON line)
        legs["hash_to_g2_1M"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    torch.cuda.empty_cache()

    # ---- configs[1] leaving in compact wire form: groups of 16, three in flight, ring of four compact buffers, free running
    try:
        nb, steps = args.batch, 96
        cpk, cmsg, csig, cexp = workload.make_batch(pkg, nb, seed=0x5EED, device=dev)
        eng = pkg.WitnessEngine(nb, 32, max_steps=16, device=dev, n_buffers=3)
        cbufs = eng.new_compact_buffer(4)
        cres = [torch.empty(nb, dtype=torch.int32, device=dev) for _ in range(4)]

        def run_compact(k_steps):
            for k in range(k_steps):
                eng.submit_compact(cpk, csig, cmsg, cbufs[k % 4], result=cres[k % 4])
            eng.flush()

        run_compact(48)
        dt = timed(lambda: run_compact(steps))
        executed = opc["fp_mul"] + opc["fp_inv"] * FPMUL_PER_INV_EXECUTED
        cb = eng.compact_bytes() / nb
        legs["compact_form"] = {"workload": "configs[1] instances leaving in compact wire form (bit-packed SHA witnesses + field witnesses, no expansion): %d steps of %d, groups of 16" % (steps, nb),
                                "value": nb * steps / dt, "unit": "instances/s", "seconds": dt, "wire_bytes_per_instance": cb,
                                "results_ok": bool(np.array_equal(cres[0].cpu().numpy().astype(bool), cexp)),
                                "roofline": {"bound": "valu-fp-mul", "executed_fpmul_per_instance": executed, "achieved": nb * steps / dt * executed, "peak": fpmul_peak,
                                             "unit": "Fp products/s (peak = blsw_microbench 1, measured on this box)", "frac": nb * steps / dt * executed / fpmul_peak,
                                             "hbm_GBps_of_wire_bytes": nb * steps / dt * cb / 1e9},
                                "cpu_baseline": "the headline's (same chains, the oracle does not build the wire form)"}
        eng.close()
        del eng, cbufs
    except Exception as exc:  # noqa: BLE001
        legs["compact_form"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    torch.cuda.empty_cache()

    # ---- BLS::verify as values (bls.rs:427-458; tests/tests.rs:239-268): the native batch verifier, 65 536 (pk, msg, sig) triples from compressed bytes
    try:
        nv = 1 << 16
        sk = np.frombuffer(b"".join(workload.secret_keys(0x5EED, 16)[i % 16].to_bytes(32, "little") for i in range(nv)), dtype=np.uint8).reshape(nv, 32).copy()
        vmsg_h = workload.messages(0x5EED, 0, nv)
        vmsg_h[15::16, 31] ^= 1  # every 16th message tampered after signing
        signed = pkg.sign_batch(torch.from_numpy(sk).to(dev), torch.from_numpy(workload.messages(0x5EED, 0, nv)).to(dev))
        vmsg = torch.from_numpy(vmsg_h).to(dev)
        res = pkg.verify_batch(signed["pk48"], vmsg, signed["sig96"])
        dt = min(timed(lambda: pkg.verify_batch(signed["pk48"], vmsg, signed["sig96"])) for _ in range(3))
        verdicts = res.cpu().numpy().astype(bool)
        # executed Fp products per verdict (estimate): decode + subgroup checks ~7 k, hash 7.5 k, projective lines 5.2 k, Miller loop 7.5 k, final exponentiation 9.5 k
        fpmul_per_verdict = 36700
        ns = 2 * cores
        pk_b, sig_b = signed["pk48"][:ns].cpu().numpy(), signed["sig96"][:ns].cpu().numpy()
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            cpu = list(ex.map(lambda i: oracle.verify_bytes(pk_b[i].tobytes(), vmsg_h[i].tobytes(), sig_b[i].tobytes()), range(ns)))
        cdt = time.perf_counter() - t0
        legs["verify_batch"] = {"workload": "BLS::verify as values (blsw_verify_batch): 65 536 (pk, msg, sig) triples from compressed bytes, every 16th tampered", "value": nv / dt, "unit": "verdicts/s",
                                "seconds": dt, "verdicts_ok": bool((verdicts == (np.arange(nv) % 16 != 15)).all()), "equals_oracle_on_sample": bool(cpu == verdicts[:ns].tolist()),
                                "roofline": {"bound": "valu-fp-mul", "executed_fpmul_per_verdict": fpmul_per_verdict, "achieved": nv / dt * fpmul_per_verdict, "peak": fpmul_peak,
                                             "unit": "Fp products/s (peak = blsw_microbench 1, measured on this box)", "frac": nv / dt * fpmul_per_verdict / fpmul_peak},
                                "cpu_baseline": {"value": ns / cdt, "unit": "verdicts/s", "cores": cores, "kind": "port",
                                                 "sample": "%d triples through the oracle's native verify on %d threads: %.2f s" % (ns, cores, cdt)}}
        del signed, vmsg, res
    except Exception as exc:  # noqa: BLE001
        legs["verify_batch"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    torch.cuda.empty_cache()

    # ---- configs[3]: one signature over 128 pairs; with distinct messages the signature of pair 0 stands in (result false, identical witness work)
    try:
        Kp, ne, st, co = 128, 16, 12, 4
        msk = workload.secret_keys(0x5EED, 16)
        mm = workload.messages(0x5EED, 1000, Kp, tag=b"mm")
        mr = pkg.sign_batch(torch.from_numpy(np.frombuffer(b"".join(msk[j % 16].to_bytes(32, "little") for j in range(Kp)), dtype=np.uint8).reshape(Kp, 32).copy()).to(dev),
                            torch.from_numpy(mm).to(dev), want_bytes=False)
        epks = mr["pk_xy"].unsqueeze(0).repeat(ne, 1, 1).contiguous()
        emsg = torch.from_numpy(mm).to(dev).unsqueeze(0).repeat(ne, 1, 1).contiguous()
        esig = mr["sig_xy"][0:1].repeat(ne, 1).contiguous()
        ress = [torch.empty(ne, dtype=torch.int32, device=dev) for _ in range(2)]
        eng = pkg.WitnessEngine(ne, 32, max_steps=co, device=dev, n_buffers=3, n_pairs=Kp)
        outs = [eng.new_witness_tensor() for _ in range(2)]

        def run_multi(k_steps):
            for k in range(k_steps):
                eng.submit_multi(epks, emsg, esig, witness=outs[k % 2], result=ress[k % 2])
            eng.flush()

        run_multi(co)
        eng.expand_stats()
        dt = timed(lambda: run_multi(st))
        out_bytes = eng.n_witness * 48
        # one instance alone through the direct entry (latency: what BASELINE configs[3] is as a single job)
        pkg.verify_multi(pkg.ParametersVar(), pkg.PublicKeyVar(epks[:1]), emsg[:1], pkg.SignatureVar(esig[:1]), want_witness=False)
        one_dt = timed(lambda: pkg.verify_multi(pkg.ParametersVar(), pkg.PublicKeyVar(epks[:1]), emsg[:1], pkg.SignatureVar(esig[:1]), want_witness=False))
        t0 = time.perf_counter()
        n_w, o_res, _, _ = oracle.witness_multi(mr["pk_xy"].cpu().numpy().view(np.uint64), mm, mr["sig_xy"][0].cpu().numpy().view(np.uint64), want_vector=False)
        cdt = time.perf_counter() - t0
        legs["multi_128"] = {"workload": "configs[3]: one signature over 128 (pk, msg) pairs, 129-pair Miller product, grouped engine: %d instances per step, %d steps, groups of %d, ring of 2 tensors, free running" % (ne, st, co),
                             "value": ne * st / dt, "unit": "instances/s", "pairs_per_s": Kp * ne * st / dt, "seconds": dt, "n_witness": eng.n_witness,
                             "one_instance_alone_ms": one_dt * 1e3, "result_equals_oracle": bool(bool(ress[0][0].item()) == o_res and n_w == eng.n_witness),
                             "roofline": {"bound": "hbm", "algorithmic_bytes_per_instance": out_bytes, "achieved": ne * st / dt * out_bytes / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                          "frac": ne * st / dt * out_bytes / 1e9 / HBM_PEAK_GBPS},
                             "cpu_baseline": {"value": 1.0 / cdt, "unit": "instances/s", "cores": 1, "kind": "port",
                                              "sample": "one 128-pair instance through the oracle's witness_multi on one thread: %.2f s" % cdt}}
        eng.close()
        del eng, outs
    except Exception as exc:  # noqa: BLE001
        legs["multi_128"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    torch.cuda.empty_cache()
    return legs


def allgather_leg(args, pkg, sharding, dist, dev, inputs, lay, world):
    """Generation + RCCL all-gather of the witness shards (north_star): every step's tensor is all-gathered in
    micro-batches of `--allgather-chunk` instances per rank and each gathered micro-batch is consumed by the digest kernel
    before the next one; a ring slot is released to the engine when its last micro-batch has been consumed."""
    import torch

    n, ring = args.batch, 2
    steps = min(args.allgather_steps, args.steps)
    d_pk, d_msg, d_sig = inputs
    # consumer mode: groups of chains run ahead into the staging, a step is expanded into its ring tensor when the gather has
    # released that tensor's previous user
    group = max(1, min(args.allgather_group, steps // 2))
    eng = pkg.WitnessEngine(n, 32, max_steps=group, device=dev, n_buffers=max(2, min(3, (steps + group - 1) // group)), consumer_mode=1)
    outs = [eng.new_witness_tensor() for _ in range(ring)]
    results = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(ring)]
    chunk = max(1, min(args.allgather_chunk, n))
    gathered = torch.empty((world * chunk, lay["n_witness"], 6), dtype=torch.int64, device=dev)
    dig = torch.empty((world * chunk, 2), dtype=torch.int64, device=dev)
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    # the consumer's stream in the HIGH-priority pool of hardware queues (with the engine's sha / expand / place streams): the runtime backs each
    # priority level with four hardware queues, and a fifth normal-priority stream (null stream + three group buffers' main streams + this one) shares
    # a queue with the null stream — every submit's input-ready marker then queues behind the consumer's waiting digests and the next launch group
    # starts ~70 ms late (profiles/r04_consumer_timeline.txt)
    consumer = torch.cuda.Stream(device=dev, priority=-1)
    state = {"next": 0}

    def drain():
        while state["next"] < eng.materialised():
            s = state["next"]
            eng.wait_step(s, consumer)
            with torch.cuda.stream(consumer):
                def consume(g, c0, rows):
                    d = pkg.witness_digest(g, out=dig[: g.shape[0]], stream=consumer)
                    acc.add_(d.sum(dim=0))

                sharding.stream_allgather(outs[s % ring], chunk, consume, buffer=gathered)
            eng.output_consumed(outs[s % ring], consumer)
            state["next"] += 1

    def run(k_steps):
        goal = state["next"] + k_steps
        for k in range(k_steps):
            while True:
                try:
                    q = eng.submitted()  # global step number: the consumer indexes the ring with it too
                    eng.submit(d_pk, d_sig, d_msg, witness=outs[q % ring], result=results[q % ring])
                    break
                except pkg.BlswBusy:
                    drain()
            drain()
        eng.flush()
        while state["next"] < goal:
            drain()
        consumer.synchronize()
        torch.cuda.synchronize()

    run(group)  # warm-up: communicator, buffers
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    eng.close()
    return float(t.item()), steps, {"form": "full", "world": world, "backend": dist.get_backend(), "micro_batch_instances_per_rank": chunk, "ring": ring, "group_steps": group, "consumer_mode": True,
                                    "consumer": "blsw_witness_digest over each gathered micro-batch",
                                    "bytes_received_per_gpu_per_step": (world - 1) * n * lay["n_witness"] * 48}


def allgather_leg_compact(args, pkg, sharding, dist, dev, inputs, lay, world):
    """Generation + all-gather in compact wire form: a step is submitted with submit_compact, its 2.6 MB per
    instance are all-gathered (one RCCL all-gather of [world][compact_bytes] per step), and this rank expands every rank's
    batch into a full witness tensor (expand_compact: the same expansion / placement kernels as a local step) which the
    digest kernel consumes before the next one is expanded. Per step every GPU writes and reads world x 34 MB x n of vectors."""
    import torch

    n = args.batch
    steps = min(args.allgather_steps, args.steps)
    group = max(1, min(args.allgather_group, steps // 2))
    ring = 4  # compact buffers; consumer mode: a step leaves for its buffer when the gather has released the buffer's previous user
    eng = pkg.WitnessEngine(n, 32, max_steps=group, device=dev, n_buffers=max(2, min(3, (steps + group - 1) // group)), consumer_mode=1)
    cbufs = eng.new_compact_buffer(ring)
    results = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(ring)]
    wit = eng.new_witness_tensor()
    dig = torch.empty((n, 2), dtype=torch.int64, device=dev)
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    # the consumer's stream in the HIGH-priority pool of hardware queues (with the engine's sha / expand / place streams): the runtime backs each
    # priority level with four hardware queues, and a fifth normal-priority stream (null stream + three group buffers' main streams + this one) shares
    # a queue with the null stream — every submit's input-ready marker then queues behind the consumer's waiting digests and the next launch group
    # starts ~70 ms late (profiles/r04_consumer_timeline.txt)
    consumer = torch.cuda.Stream(device=dev, priority=-1)
    comm = torch.cuda.Stream(device=dev)
    d_pk, d_msg, d_sig = inputs
    state = {"next": 0}

    def expand(c):
        eng.expand_compact(c, wit, stream=consumer)
        return wit

    def consume(w, r, k):
        acc.add_(pkg.witness_digest(w, out=dig, stream=consumer).sum(dim=0))

    # double-buffered: the RCCL all-gather of step k + 1 runs on the communication stream while every rank's part of step k is
    # expanded and digested on the consumer stream (two gathered buffers)
    overlap = (world > 1) if args.allgather_overlap < 0 else bool(args.allgather_overlap)
    pipe = sharding.CompactGatherPipeline(world, eng.compact_bytes(), dev, expand, consume, comm_stream=comm if overlap else consumer, consumer_stream=consumer)

    def drain():
        while state["next"] < eng.materialised():
            s = state["next"]
            buf = cbufs[s % ring]
            pipe.push(buf, before=lambda st, s=s: eng.wait_step(s, st), after=lambda st, buf=buf: eng.output_consumed(buf, st))
            state["next"] += 1

    def run(k_steps):
        goal = state["next"] + k_steps
        for k in range(k_steps):
            while True:
                try:
                    q = eng.submitted()  # global step number: the consumer indexes the ring with it too
                    eng.submit_compact(d_pk, d_sig, d_msg, cbufs[q % ring], result=results[q % ring])
                    break
                except pkg.BlswBusy:
                    drain()
            drain()
        eng.flush()
        while state["next"] < goal:
            drain()
        pipe.flush()
        comm.synchronize()
        consumer.synchronize()
        torch.cuda.synchronize()

    run(group)  # warm-up: communicator, buffers
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    cb = eng.compact_bytes()
    eng.close()
    return float(t.item()), steps, {"form": "compact", "world": world, "backend": dist.get_backend(), "ranks": dist.get_world_size(), "group_steps": group, "ring": ring, "consumer_mode": True,
                                    "overlap": "all-gather of step k + 1 on a communication stream beside expansion + digest of step k (two gathered buffers)" if overlap else "none (one stream: world 1, the gather is a local copy)",
                                    "wire_bytes_per_instance": cb / n,
                                    "bytes_received_per_gpu_per_step": (world - 1) * cb,
                                    "bytes_expanded_per_gpu_per_step": world * n * lay["n_witness"] * 48,
                                    "consumer": "blsw_witness_digest over every expanded batch (all ranks' batches, on every rank)"}


def main():
    args = parse_args()
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        spawn_ranks(args)  # does not return
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    dist = None
    # BLSW_TEST_BACKEND=gloo: rehearsal of world > 1 on a one-GPU box (every rank on device 0, which RCCL refuses); never set by the driver
    backend = os.environ.get("BLSW_TEST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if under_launcher:  # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist

        # RCCL prints a version banner on stdout when the communicator is created: keep stdout for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        assert dist.get_world_size() == args.gpus
    pkg = importlib.import_module("bls-verify-gadget_amd")
    sharding = importlib.import_module("bls-verify-gadget_amd.sharding")
    workload = importlib.import_module("bls-verify-gadget_amd.workload")
    pkg.lib()

    n = args.batch
    # n DISTINCT instances (16 keys, n uniform messages, every 16th tampered), this rank's block of the
    # global batch, minted on the GPU by the product's own signer (blsw_sign_batch)
    d_pk, d_msg, d_sig, expect = workload.make_batch(pkg, n, seed=0x5EED, device=dev, start=rank * n)
    lay = pkg.layout(32)
    # engine: up to `--coalesce` submitted batches are fused into one launch group (fills the SIMDs); witness tensors
    # are written per step, in order, into a ring of `--outputs` output tensors (a consumer would drain them in order)
    out_bytes = n * lay["n_witness"] * 48
    free_b, _ = torch.cuda.mem_get_info(dev)
    n_out = max(1, min(args.outputs, args.steps + args.warmup))
    coalesce = balanced_coalesce(args.steps, max(1, args.coalesce))
    buffers = max(1, min(args.buffers, (args.steps + coalesce - 1) // coalesce))
    # leave >= 25 % of HBM to the runtime (per-queue scratch = stacks of the chain kernels)
    while buffers > 1 and pkg.engine_workspace_bytes(n, 32, coalesce, buffers) + n_out * out_bytes > args.mem_frac * free_b:
        buffers -= 1
    eng_opts = {}
    if args.expand_variant >= 0:
        eng_opts["expand_variant"] = args.expand_variant
    eng = pkg.WitnessEngine(n, 32, max_steps=coalesce, device=dev, n_buffers=buffers, **eng_opts)
    outs = [eng.new_witness_tensor() for _ in range(n_out)]
    results = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(n_out)]
    torch.cuda.synchronize()

    def step(k):
        eng.submit(d_pk, d_sig, d_msg, witness=outs[k % n_out], result=results[k % n_out])

    # same-box yardstick: what this box's memory system gives a plain fill of one witness tensor in the expansion's store geometry (boxes of the pool
    # differ by 20-30 %; the tensor is overwritten by the warm-up steps)
    fill_bps = pkg.fill_rate(outs[0], reps=2)
    for k in range(args.warmup):
        step(k)
    eng.flush()
    torch.cuda.synchronize()
    eng.expand_stats()  # reset: only launches of the timed region are averaged
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    eng.flush()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # live measurement of the dominant-by-bytes kernel (k_sha_expand): HIP events recorded around it on the stream it
    # ran on, for every launch of the timed region (at most 1024)
    exp_count, exp_avg_ms = eng.expand_stats()
    res = torch.stack(results)
    ok = bool((res.cpu().numpy().astype(bool) == expect[None, :]).all())
    # what the timed region left in the ring: every step submitted the same batch, so each ring tensor must hold exactly the
    # witness vectors a direct-mode engine (chains write in place, no staging, no expansion ordering) produces for that batch
    ring_digests = [pkg.witness_digest(o).cpu() for o in outs]
    gathered_ok = None
    per_rank_dt = [dt]
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)  # each rank's own clock over its timed region: a straggler shows as min << max
        per_rank_dt = [float(x.item()) for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        allres = sharding.all_gather_results(results[0], n * world)  # result shards only (multi-GPU)
        gathered_ok = bool(allres.numel() == n * world)
    eng.close()
    del eng
    direct = pkg.WitnessEngine(n, 32, max_steps=1, device=dev, n_buffers=1)
    direct.submit(d_pk, d_sig, d_msg, witness=outs[0], result=results[0])
    direct.flush()
    torch.cuda.synchronize()
    ref_digest = pkg.witness_digest(outs[0]).cpu()
    direct.close()
    witness_ok = bool(all(torch.equal(d, ref_digest) for d in ring_digests)) and bool(ref_digest.abs().sum().item() != 0)
    del direct, outs
    torch.cuda.empty_cache()

    consumer = None
    if world == 1 and args.consumer_shard >= 2 * n:  # the real-use rate: a consumer reads every tensor before it is overwritten
        try:
            shard = args.consumer_shard // n * n
            stream_shard = importlib.import_module("tools.shard_rehearsal").stream_shard  # the measurement harness of the consumer legs (not product code)
            # ONE engine for the warm-up and the timed runs, as a consumer that streams shard after shard uses it: an engine's fresh streams pay the
            # runtime's one-time costs (per-queue scratch for every chain kernel of this group shape, first dispatches): the first shard on a new engine
            # took 0.19-2.9 s against 0.14 s for the shards after it. Warm-up: the same shard once, untimed. The shard is a 0.14 s job, so it then
            # runs `--consumer-reps` times; value = the MEDIAN run, every run's seconds are in the line
            kept = {}
            stream_shard(pkg, shard, n, 2, 0, 1, device=dev, keep=kept)
            runs = []
            for _ in range(max(1, args.consumer_reps)):
                cs = stream_shard(pkg, shard, n, 2, 0, 1, device=dev, keep=kept)
                runs.append((cs["seconds"], cs))
                cs = None
            kept["eng"].close()
            kept.clear()
            torch.cuda.empty_cache()
            runs_seconds = [r[0] for r in runs]
            runs_ok = all(bool(r[1]["results_ok"]) for r in runs)
            cs = sorted(runs, key=lambda r: r[0])[len(runs) // 2][1]
            del runs
            consumer = {"value": cs["instances_per_s"], "unit": "instances/s", "shard_instances": shard, "steps": cs["steps"], "ring": 2, "group_steps": cs["group_steps"],
                        "seconds": cs["seconds"], "first_step_ms": cs["first_step_ms"], "runs_seconds": runs_seconds, "best_run_value": shard / min(runs_seconds), "statistic": "median of %d runs" % len(runs_seconds), "results_ok": runs_ok,
                        "digests_equal_free_running": bool((cs["digests"][:n].view("int64") == ref_digest.numpy()).all()) if cs["first_instance"] == 0 else None,
                        "consumer": "blsw_witness_digest reads every witness tensor before the engine may overwrite it (consumer-mode engine: late materialisation)",
                        "hbm_bytes_per_instance": 2 * lay["n_witness"] * 48}
            del cs
            if args.consumer_steady_shard >= 4 * n:
                torch.cuda.empty_cache()
                kept = {}
                stream_shard(pkg, args.consumer_steady_shard // n * n, n, 2, 0, 1, device=dev, keep=kept)  # the engine's one-time costs (see above)
                cs = stream_shard(pkg, args.consumer_steady_shard // n * n, n, 2, 0, 1, device=dev, keep=kept)
                kept["eng"].close()
                kept.clear()
                consumer["steady"] = {"value": cs["instances_per_s"], "shard_instances": cs["n_shard"], "steps": cs["steps"], "group_steps": cs["group_steps"], "seconds": cs["seconds"],
                                      "results_ok": cs["results_ok"], "hbm_GBps_written_plus_read": cs["instances_per_s"] * 2 * lay["n_witness"] * 48 / 1e9}
                del cs
        except Exception as exc:  # noqa: BLE001 (reported in the JSON line)
            consumer = dict(consumer or {}, error="%s: %s" % (type(exc).__name__, exc))
        torch.cuda.empty_cache()

    ag = None
    if dist and args.allgather_steps > 0:
        leg = allgather_leg_compact if (args.allgather_form == "compact" and n % 64 == 0) else allgather_leg
        try:  # the second leg never takes the headline line down with it
            ag = leg(args, pkg, sharding, dist, dev, (d_pk, d_msg, d_sig), lay, world)
        except Exception as exc:  # noqa: BLE001 (reported in the JSON line)
            ag = (None, 0, {"form": args.allgather_form, "backend": dist.get_backend(), "error": "%s: %s" % (type(exc).__name__, exc)})
    dist_backend = dist.get_backend() if dist else None
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    value = n * world * args.steps / dt
    expand_bytes = n * lay["sha_bits"] * 48  # bytes one k_sha_expand launch must write
    achieved = expand_bytes / (exp_avg_ms * 1e-3) / 1e9 if exp_avg_ms else 0.0
    bytes_per_instance = 48 * (lay["n_witness"] + lay["n_instance_vars"]) + 320
    opc = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_opcount.json")))
    # HBM bytes per k_sha_expand launch from the PMC passes committed under profiles/ (rocprofv3 --pmc cannot be combined
    # with the timed run); only quoted for the workload it was collected on
    traffic, traffic_source, whole_step, whole_step_file = None, None, None, None
    for name in TRAFFIC_FILES:
        try:
            if whole_step is None:
                whole_step = json.load(open(os.path.join(ROOT, "profiles", name))).get("whole_step")
                whole_step_file = name if whole_step else None
            tr = json.load(open(os.path.join(ROOT, "profiles", name)))["k_sha_expand"]
            if tr["instances_per_launch"] == n and lay["msg_len"] == 32:
                traffic = (tr["write_kib"] + tr["fetch_kib"]) * 1024.0
                traffic_source = "profiles/%s (rocprofv3 --pmc WRITE_SIZE and FETCH_SIZE passes of this command, per launch, bytes; FETCH_SIZE doubled as the gfx950 guide prescribes)" % name
                break
        except (OSError, KeyError, ValueError):
            continue
    mad_per_instance = (opc["fp_mul"] + opc["fp_inv"] * FPMUL_PER_INV) * MAD_PER_FPMUL
    executed_fpmul = opc["fp_mul"] + opc["fp_inv"] * FPMUL_PER_INV_EXECUTED
    mad_peak = pkg.microbench(0, iters=8192, blocks=8192)
    fpmul_peak = pkg.microbench(1, iters=512, blocks=8192)
    out = {
        "metric": "BLS-verify witness instances/sec (full pairing circuit)",
        "value": value,
        "unit": "instances/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # generation only: no data-path collective inside the timed region (the all-gather leg is value_with_allgather below)
        "value_generation_only": value,
        "per_rank": {"instances_per_s_min": n * args.steps / max(per_rank_dt), "instances_per_s_max": n * args.steps / min(per_rank_dt),
                     "seconds": per_rank_dt, "backend": dist_backend},
        "dtype": "u32 (381-bit Montgomery integers mod the BLS12-381 prime as 12 x 32-bit limbs in memory, products on 14 x 28-bit limbs with 64-bit columns; SHA-256 words)",
        "data": "synthetic",
        "config": {"workload": "configs[1]: batch of 1024 independent BLS-verify instances per GPU per step (1024 distinct messages, 16 keys, every 16th tampered), 32-byte messages, full witness vectors written",
                   "instances_per_gpu_per_step": n, "batches_fused_per_launch_group": coalesce, "groups_in_flight": buffers, "output_ring": n_out, "n_witness": lay["n_witness"], "results_ok": ok, "witness_ok": witness_ok,
                   "result_shards_gathered": gathered_ok, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
        "roofline": {"bound": "hbm", "kernel": "k_sha_expand", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": expand_bytes, "avg_launch_ms": exp_avg_ms, "launches_timed": exp_count,
                     # the same-box yardstick: a plain fill of one witness tensor in the kernel's store geometry, measured before the timed region
                     "fill_GBps": fill_bps / 1e9, "frac_of_fill": achieved / (fill_bps / 1e9) if fill_bps else None},
        "roofline_whole_path": {"bound": "hbm", "algorithmic_bytes_per_instance": bytes_per_instance,
                                "achieved": value / world * bytes_per_instance / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": value / world * bytes_per_instance / 1e9 / HBM_PEAK_GBPS},
        "roofline_valu": {"bound": "valu-int32-mad", "algorithmic_mad_per_instance": mad_per_instance, "measured_peak_mad_per_s": mad_peak,
                          "measured_peak_fpmul_per_s": fpmul_peak, "achieved_mad_per_s": value / world * mad_per_instance,
                          "frac": value / world * mad_per_instance / mad_peak},
        # utilisation, not Fermat-priced throughput: Fp products the kernels actually execute per instance (products of the
        # op counter + 27 product-times per safegcd inversion) over the measured Fp-product rate of the chip
        "roofline_valu_executed": {"bound": "valu-fp-mul", "executed_fpmul_per_instance": executed_fpmul, "achieved_fpmul_per_s": value / world * executed_fpmul,
                                   "measured_peak_fpmul_per_s": fpmul_peak, "frac": value / world * executed_fpmul / fpmul_peak},
    }
    if whole_step and whole_step["instances_per_step"] == n and lay["msg_len"] == 32:
        # every byte the HBM moves per step (PMC, all kernels) over the measured step time: what the memory system delivers to
        # the path as a whole — the expansion's own figure above is lower because the other kernels' traffic runs beside it
        tot = whole_step["write_bytes"] + whole_step["fetch_bytes"]
        out["roofline_hbm_total"] = {"bound": "hbm", "traffic_per_step": tot, "achieved": tot / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                     "frac": tot / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS, "source": "profiles/%s whole_step (rocprofv3 --pmc passes)" % whole_step_file}
    out["witness_ok"] = witness_ok  # digests of the ring tensors after the timed region == a direct-mode engine's on the same batch
    if consumer:
        out["value_consumer_mode"] = consumer.get("value")
        out["value_consumer_mode_steady"] = (consumer.get("steady") or {}).get("value")
        out["consumer_mode"] = consumer
    if ag:
        ag_dt, ag_steps, ag_info = ag
        if ag_dt:
            out["value_with_allgather"] = n * world * ag_steps / ag_dt
        out["allgather"] = dict({"steps": ag_steps, "seconds": ag_dt}, **ag_info)
    if world == 1 and args.side_legs:
        out["side"] = side_legs(args, pkg, workload, dev, fpmul_peak, opc)
    if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only: the other ranks of a multi-GPU run would wait ~15 s in the final barrier
        out["cpu_baseline"] = cpu_baseline(args, d_pk, d_msg, d_sig, n)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
