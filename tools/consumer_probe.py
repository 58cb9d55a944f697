#!/usr/bin/env python3
"""The consumer-mode shard legs of bench.py (tools/shard_rehearsal.py: stream_shard) repeated in one process — run-to-run spread, the
group ramp on / off, group sizes. One line per run."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# as bench.py: more hardware queues per priority level than the runtime's default of 4, so that the engine's streams and the consumer's do not share one
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    import torch

    pkg = importlib.import_module("bls-verify-gadget_amd")
    sharding = importlib.import_module("tools.shard_rehearsal")
    dev = torch.device("cuda:0")
    n = 1024
    sharding.stream_shard(pkg, 2 * n, n, 2, 0, 1, device=dev)
    torch.cuda.empty_cache()
    plans = [(8192, None, True), (8192, None, False), (8192, None, True), (8192, None, False), (8192, 6, True), (8192, 8, True), (8192, 2, False),
             (32768, None, True), (32768, None, False), (32768, None, True), (32768, 8, True), (32768, 10, True)]
    if len(sys.argv) > 1:
        plans = [tuple(json.loads(a)) for a in sys.argv[1:]]
    kept, kept_plan = {}, None
    for plan in plans:
        shard, group, ramp = plan[:3]
        ring = plan[3] if len(plan) > 3 else 2
        if kept_plan != (shard, group, ramp, ring):  # consecutive runs of one plan share an engine (as a consumer streaming shards would)
            if kept:
                kept["eng"].close()
            kept, kept_plan = {}, (shard, group, ramp, ring)
            torch.cuda.empty_cache()
        cs = sharding.stream_shard(pkg, shard, n, ring, 0, 1, device=dev, group=group, ramp=ramp, keep=kept)
        print(json.dumps({"shard": shard, "ring": ring, "group": cs["group_steps"], "ramp": cs["group_ramp"], "instances_per_s": round(cs["instances_per_s"]), "seconds": round(cs["seconds"], 4), "first_step_ms": round(cs["first_step_ms"], 2),
                          "results_ok": cs["results_ok"]}), flush=True)
        del cs
        torch.cuda.empty_cache()
    if kept:
        kept["eng"].close()


if __name__ == "__main__":
    main()
