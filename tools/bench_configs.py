#!/usr/bin/env python3
"""Side measurements for SURVEY.md §8d configs 4 and 5 and the entry points around the hot path (one JSON line each):
hash-to-G2 over 1 M messages, same-message aggregate_verify with 128 keys, input decode, signer. Not the headline bench."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hash-n", type=int, default=1 << 20)
    ap.add_argument("--hash-chunk", type=int, default=1 << 18)  # >= 2 waves per SIMD for the value-only cofactor ladder
    ap.add_argument("--agg-n", type=int, default=1024)
    ap.add_argument("--agg-keys", type=int, default=128)
    ap.add_argument("--sign-n", type=int, default=16384)
    ap.add_argument("--agg-steps", type=int, default=9)
    ap.add_argument("--agg-coalesce", type=int, default=3)
    ap.add_argument("--multi-pairs", type=int, default=128)
    ap.add_argument("--multi-engine-n", type=int, default=16, help="instances per step of the grouped-engine configs[3] run (ring of two output tensors of n x 4.19 GB)")
    ap.add_argument("--multi-engine-steps", type=int, default=24)
    ap.add_argument("--multi-engine-coalesce", type=int, default=4)
    ap.add_argument("--multi-n", type=int, default=48)  # 48 x 4.19 GB of witnesses per call (the rest of HBM stays free for the runtime's per-queue scratch)
    ap.add_argument("--only", default="", help="comma-separated sections: side, multi-direct, multi-engine, multi-consumer (default: all)")
    args = ap.parse_args()
    run = set(args.only.split(",")) if args.only else {"side", "multi-direct", "multi-engine", "multi-consumer"}
    import torch

    dev = torch.device("cuda:0")
    pkg = importlib.import_module("bls-verify-gadget_amd")
    workload = importlib.import_module("bls-verify-gadget_amd.workload")
    pkg.lib()

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r

    if "side" in run:
        # config 5: msg_i = SHA-256(seed || "h" || i); instance 0 is the 32-zero-byte message of bls.rs:645 by convention
        chunk = min(args.hash_chunk, args.hash_n)
        out = torch.empty((chunk, 24), dtype=torch.int64, device=dev)
        msgs = workload.messages(0x5EED, 0, chunk, tag=b"h")
        msgs[0] = 0
        d = torch.from_numpy(msgs).to(dev)
        pkg.hash_to_g2_batch(d, out=out)  # warm-up
        torch.cuda.synchronize()
        first = out[0].cpu().numpy().view(np.uint64).tolist()
        total = 0.0
        done = 0
        while done < args.hash_n:
            dt, _ = timed(lambda: pkg.hash_to_g2_batch(d, out=out))
            total += dt
            done += chunk
        print(json.dumps({"workload": "configs[4] hash-to-G2 (SURVEY 8d config 5)", "messages": done, "chunk": chunk, "seconds": total, "value": done / total,
                          "unit": "messages/s", "first_output_x_c0_limbs": first[:6]}))

        # signer + key derivation
        sk = np.frombuffer(b"".join((k.to_bytes(32, "little")) for k in workload.secret_keys(0x5EED, 16)) * (args.sign_n // 16), dtype=np.uint8).reshape(-1, 32).copy()
        m = workload.messages(0x5EED, 0, sk.shape[0])
        dsk, dm = torch.from_numpy(sk).to(dev), torch.from_numpy(m).to(dev)
        pkg.sign_batch(dsk, dm)
        dt, r = timed(lambda: pkg.sign_batch(dsk, dm))
        print(json.dumps({"workload": "sign + sk->pk (bls.rs:411-425)", "instances": sk.shape[0], "seconds": dt, "value": sk.shape[0] / dt, "unit": "signatures/s"}))
        # decode of the compressed outputs
        pkg.decode_batch(r["pk48"], r["sig96"])
        dt, _ = timed(lambda: pkg.decode_batch(r["pk48"], r["sig96"]))
        print(json.dumps({"workload": "decode pk48 + sig96 incl. subgroup checks (bls.rs:219-242, 316-339)", "instances": sk.shape[0], "seconds": dt,
                          "value": sk.shape[0] / dt, "unit": "instances/s"}))

        # Signature::aggregate / PublicKey::aggregate: lists of 8 of the signatures / keys just minted (decode with subgroup check + sum + serialise)
        k_agg = 8
        lists = sk.shape[0] // k_agg
        s96, p48 = r["sig96"][: lists * k_agg].reshape(lists, k_agg, 96).contiguous(), r["pk48"][: lists * k_agg].reshape(lists, k_agg, 48).contiguous()
        for name, fn, pts in (("Signature::aggregate (bls.rs:288-300)", pkg.aggregate_signatures, s96), ("PublicKey::aggregate (bls.rs:183-195)", pkg.aggregate_public_keys, p48)):
            fn(pts)
            dt, (o, stt) = timed(lambda: fn(pts))
            print(json.dumps({"workload": name + ", lists of %d compressed points, decode incl. subgroup checks + sum + serialise" % k_agg, "lists": lists, "points": lists * k_agg,
                              "seconds": dt, "value": lists * k_agg / dt, "unit": "points/s", "all_ok": bool((stt == 0).all().item())}))

        # the single-key path with the steps leaving in compact wire form (bit-packed SHA witnesses + field witnesses, 2.6 MB per
        # instance; what a sharded job ships): generation without the 34 MB-per-instance expansion
        nb = 1024
        cpk, cmsg, csig, cexp = workload.make_batch(pkg, nb, seed=0x5EED, device=dev)
        eng = pkg.WitnessEngine(nb, 32, max_steps=16, device=dev, n_buffers=3)
        cbufs = eng.new_compact_buffer(4)
        cres = [torch.empty(nb, dtype=torch.int32, device=dev) for _ in range(4)]

        def run_compact(k_steps):
            for k in range(k_steps):
                eng.submit_compact(cpk, csig, cmsg, cbufs[k % 4], result=cres[k % 4])
            eng.flush()

        run_compact(48)
        dt, _ = timed(lambda: run_compact(192))
        print(json.dumps({"workload": "configs[1] instances leaving in compact wire form (no expansion; groups of 16, 3 in flight, ring of 4 compact buffers, free running)",
                          "instances": nb * 192, "seconds": dt, "value": nb * 192 / dt, "unit": "instances/s", "compact_bytes_per_instance": eng.compact_bytes() / nb,
                          "results_ok": bool(np.array_equal(cres[0].cpu().numpy().astype(bool), cexp))}))
        eng.close()
        del eng, cbufs
        torch.cuda.empty_cache()

        # config 4 (reference-defined variant): same-message aggregate_verify, K keys, all-ones bitmap
        n, K = args.agg_n, args.agg_keys
        sks = workload.secret_keys(0x5EED, K)
        ksk = np.frombuffer(b"".join(s.to_bytes(32, "little") for s in sks), dtype=np.uint8).reshape(K, 32).copy()
        msg1 = workload.messages(0x5EED, 7, 1)
        kr = pkg.sign_batch(torch.from_numpy(ksk).to(dev), torch.from_numpy(np.repeat(msg1, K, 0)).to(dev), want_bytes=False)
        agg_sk = sum(sks) % workload.R_MOD
        ar = pkg.sign_batch(torch.from_numpy(np.frombuffer(agg_sk.to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32).copy()).to(dev), torch.from_numpy(msg1).to(dev),
                            want_bytes=False)
        pks = kr["pk_xy"].unsqueeze(0).repeat(n, 1, 1).contiguous()
        bitmap = torch.ones((n, K), dtype=torch.uint8, device=dev)
        sig = ar["sig_xy"].repeat(n, 1).contiguous()
        dmsg = torch.from_numpy(np.repeat(msg1, n, 0)).to(dev)
        for want in (False, True):
            pkg.aggregate_verify(pkg.ParametersVar(), pkg.PublicKeyVar(pks), bitmap, dmsg, pkg.SignatureVar(sig), want_witness=want)
            dt, (res, cnt, _) = timed(lambda: pkg.aggregate_verify(pkg.ParametersVar(), pkg.PublicKeyVar(pks), bitmap, dmsg, pkg.SignatureVar(sig), want_witness=want))
            print(json.dumps({"workload": "aggregate_verify, %d keys, all-ones bitmap (constraints.rs:153-191), one direct call" % K, "instances": n, "witness_written": want,
                              "seconds": dt, "value": n / dt, "unit": "instances/s", "all_true": bool((res == 1).all().item()), "count_ok": bool((cnt == K).all().item())}))
        torch.cuda.empty_cache()
        # the same through the grouped engine (options.n_keys): groups of `agg_coalesce` batches, two groups in flight, ring of two tensors
        steps, coal = args.agg_steps, args.agg_coalesce
        eng = pkg.WitnessEngine(n, 32, max_steps=coal, device=dev, n_buffers=2, n_keys=K)
        outs = [eng.new_witness_tensor() for _ in range(2)]
        ress = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
        cnts = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]

        def run_engine(k_steps):
            for k in range(k_steps):
                eng.submit_aggregate(pks, bitmap, sig, dmsg, witness=outs[k % 2], result=ress[k % 2], count=cnts[k % 2])
            eng.flush()

        run_engine(coal)
        dt, _ = timed(lambda: run_engine(steps))
        print(json.dumps({"workload": "aggregate_verify, %d keys, grouped engine (%d batches per group, 2 groups in flight), witness tensors written" % (K, coal), "instances": n * steps,
                          "seconds": dt, "value": n * steps / dt, "unit": "instances/s", "n_witness": eng.n_witness, "all_true": bool((ress[0] == 1).all().item()),
                          "count_ok": bool((cnts[0] == K).all().item())}))
        eng.close()
        del eng, outs
        torch.cuda.empty_cache()

    # BASELINE configs[3]: ONE signature over K (pk, msg) pairs, K + 1 pairs in the Miller product (blsw_verify_multi_batch)
    Kp, nm = args.multi_pairs, args.multi_n
    msk = workload.secret_keys(0x5EED, 16)
    mm = workload.messages(0x5EED, 1000, Kp, tag=b"mm")
    mr = pkg.sign_batch(torch.from_numpy(np.frombuffer(b"".join(msk[j % 16].to_bytes(32, "little") for j in range(Kp)), dtype=np.uint8).reshape(Kp, 32).copy()).to(dev),
                        torch.from_numpy(mm).to(dev))
    # sigma = sum of the K signatures: summed on the host from the compressed outputs by the product's own decode + a G2 sum is not
    # exposed, so the aggregate signature is produced as sk_total * H(m) only when all messages are equal; with distinct messages
    # the bench uses the signature of pair 0 (result false) — the witness work is identical, only the output Boolean differs
    mpks = mr["pk_xy"].unsqueeze(0).repeat(nm, 1, 1).contiguous()
    mmsg = torch.from_numpy(mm).to(dev).unsqueeze(0).repeat(nm, 1, 1).contiguous()
    msig = mr["sig_xy"][0:1].repeat(nm, 1).contiguous()
    if "multi-direct" in run:
        for want in (False, True):
            pkg.verify_multi(pkg.ParametersVar(), pkg.PublicKeyVar(mpks), mmsg, pkg.SignatureVar(msig), want_witness=want)
            dt, (res, _) = timed(lambda: pkg.verify_multi(pkg.ParametersVar(), pkg.PublicKeyVar(mpks), mmsg, pkg.SignatureVar(msig), want_witness=want))
            print(json.dumps({"workload": "configs[3]: one signature over %d (pk, msg) pairs, %d-pair Miller product (blsw_verify_multi_batch)" % (Kp, Kp + 1), "instances": nm,
                              "pairs": Kp * nm, "witness_written": want, "n_witness": pkg.layout_multi(32, Kp)["n_witness"], "seconds": dt, "value": nm / dt, "unit": "instances/s",
                              "pairs_per_s": Kp * nm / dt}))

        del res, _  # the direct call's 201 GB witness tensor
        torch.cuda.empty_cache()
    # the same circuit through the grouped engine (options.n_pairs): groups of `coalesce` steps of n instances, three groups in flight,
    # ring of two output tensors, free running (every step writes its n x 4.19 GB) — more instances in flight than output tensors
    ne, st, co = args.multi_engine_n, args.multi_engine_steps, args.multi_engine_coalesce
    ress = [torch.empty(ne, dtype=torch.int32, device=dev) for _ in range(2)]
    epks = mr["pk_xy"].unsqueeze(0).repeat(ne, 1, 1).contiguous()
    emsg = torch.from_numpy(mm).to(dev).unsqueeze(0).repeat(ne, 1, 1).contiguous()
    esig = mr["sig_xy"][0:1].repeat(ne, 1).contiguous()
    if "multi-engine" in run:
        eng = pkg.WitnessEngine(ne, 32, max_steps=co, device=dev, n_buffers=3, n_pairs=Kp)
        outs = [eng.new_witness_tensor() for _ in range(2)]

        def run_multi(k_steps):
            for k in range(k_steps):
                eng.submit_multi(epks, emsg, esig, witness=outs[k % 2], result=ress[k % 2])
            eng.flush()

        run_multi(co)
        dt, _ = timed(lambda: run_multi(st))
        print(json.dumps({"workload": "configs[3] through the grouped engine: one signature over %d pairs, %d instances per step, %d steps per group, 3 groups in flight, ring of 2 tensors, free running" % (Kp, ne, co),
                          "instances": ne * st, "pairs": Kp * ne * st, "witness_written": True, "seconds": dt, "value": ne * st / dt, "unit": "instances/s", "pairs_per_s": Kp * ne * st / dt,
                          "output_GBps": ne * st * eng.n_witness * 48 / dt / 1e9}))
        eng.close()
        del eng, outs
        torch.cuda.empty_cache()

    # configs[3] with a CONSUMER (round 4): a consumer-mode engine (late materialisation out of the 180 MB per instance of staging), ring of two
    # n x 4.19 GB tensors, the digest kernel reads every tensor before the engine may overwrite it; and the same steps leaving in COMPACT form
    # (blsw_engine_submit_multi_compact, ~180 MB per instance) through a ring of four buffers, each expanded (blsw_engine_expand_compact) into
    # one tensor and digested — what a sharded run of this circuit would ship and what its receiver would do. The consumer's stream is in the
    # high-priority queue pool (profiles/r04_consumer_timeline.txt).
    if "multi-consumer" not in run:
        return
    consumer = torch.cuda.Stream(device=dev, priority=-1)

    def drive(eng, n_steps, ring_outs, submit, consume):
        """submit(k, out) / consume(s, out, stream): steps through a ring with BUSY -> drain, as tools/shard_rehearsal.py's stream_shard does (the engine's step
        numbers run on across calls: this call's steps are base .. base + n_steps - 1)"""
        base = eng.submitted()
        assert base == eng.materialised()
        state = {"next": base}

        def drain():
            while state["next"] < eng.materialised():
                s0 = state["next"]
                out = ring_outs[s0 % len(ring_outs)]
                eng.wait_step(s0, consumer)
                consume(s0, out, consumer)
                eng.output_consumed(out, consumer)
                state["next"] += 1

        for k in range(base, base + n_steps):
            while True:
                try:
                    submit(k, ring_outs[k % len(ring_outs)])
                    break
                except pkg.BlswBusy:
                    drain()
            drain()
        eng.flush()
        while state["next"] < base + n_steps:
            drain()
        consumer.synchronize()
        torch.cuda.synchronize()

    dig = torch.empty((ne, 2), dtype=torch.int64, device=dev)
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    eng = pkg.WitnessEngine(ne, 32, max_steps=co, device=dev, n_buffers=3, n_pairs=Kp, consumer_mode=1)
    outs = [eng.new_witness_tensor() for _ in range(2)]

    def consume_tensor(s0, out, stream):
        with torch.cuda.stream(stream):
            acc.add_(pkg.witness_digest(out, out=dig, stream=stream).sum(dim=0))

    submit_full = lambda k, out: eng.submit_multi(epks, emsg, esig, witness=out, result=ress[k % 2])
    drive(eng, co, outs, submit_full, consume_tensor)
    ref = acc.clone()
    acc.zero_()
    dt, _ = timed(lambda: drive(eng, st, outs, submit_full, consume_tensor))
    # every step has the same inputs: the digest sum of the warm-up's `co` steps and of the `st` timed steps are co x D and st x D (mod 2^64)
    same = all((int(a) * co - int(r) * st) % (1 << 64) == 0 for a, r in zip(acc.cpu().tolist(), ref.cpu().tolist())) and bool(ref.abs().sum().item() != 0)
    print(json.dumps({"workload": "configs[3] through the grouped engine in CONSUMER mode: one signature over %d pairs, %d instances per step, %d steps per group, ring of 2 tensors, "
                                  "blsw_witness_digest reads every tensor before it may be overwritten" % (Kp, ne, co),
                      "instances": ne * st, "pairs": Kp * ne * st, "seconds": dt, "value": ne * st / dt, "unit": "instances/s", "pairs_per_s": Kp * ne * st / dt,
                      "hbm_GBps_written_plus_read": 2 * ne * st * eng.n_witness * 48 / dt / 1e9,
                      "digest_sums_consistent": same}))
    del outs
    torch.cuda.empty_cache()
    if eng.n * Kp % 64 == 0:
        cb = eng.compact_bytes()
        cbufs = list(eng.new_compact_buffer(4))
        wit = eng.new_witness_tensor()
        submit_c = lambda k, out: eng.submit_multi_compact(epks, emsg, esig, out, result=ress[k % 2])

        def consume_compact(s0, out, stream):
            eng.expand_compact(out, wit, stream=stream)
            with torch.cuda.stream(stream):
                acc.add_(pkg.witness_digest(wit, out=dig, stream=stream).sum(dim=0))

        for label, consume in (("compact form only (no expansion): the sender's side of a sharded run", lambda s0, out, stream: None),
                               ("compact form, expanded on the 'receiver' (blsw_engine_expand_compact) and digested", consume_compact)):
            drive(eng, co, cbufs, submit_c, consume)
            acc.zero_()
            dt, _ = timed(lambda: drive(eng, st, cbufs, submit_c, consume))
            print(json.dumps({"workload": "configs[3], consumer-mode engine, steps leave in %s: %d pairs, %d instances per step, ring of 4 compact buffers" % (label, Kp, ne),
                              "instances": ne * st, "pairs": Kp * ne * st, "seconds": dt, "value": ne * st / dt, "unit": "instances/s", "pairs_per_s": Kp * ne * st / dt,
                              "wire_bytes_per_instance": cb / ne, "vector_bytes_per_instance": eng.n_witness * 48}))
    eng.close()


if __name__ == "__main__":
    main()
