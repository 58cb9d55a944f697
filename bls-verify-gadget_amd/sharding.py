"""Instance sharding across ranks (one process per GPU). Instances are fully independent: contiguous blocks per
rank, and GENERATION needs no collective. What is exchanged, with torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU):
the per-instance result booleans (all_gather_results), and — the north star's all-gather of witness shards — every step's batch,
either as full witness tensors in micro-batches (stream_allgather) or in the compact wire form that each receiver expands
(stream_allgather_compact, CompactGatherPipeline: the gather of step k + 1 beside expansion + consumption of step k)."""


def shard_range(n_total, rank, world):
    """Contiguous block [lo, hi) of instances owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_results(local_results, n_total, group=None):
    """Gathers the int32 result shards of all ranks into one [n_total] tensor (ragged shards are padded)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    m = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(m, dtype=local_results.dtype, device=local_results.device)
    pad[: local_results.numel()] = local_results
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)])


def all_gather_witness_chunk(local_chunk, group=None, out=None):
    """RCCL all-gather of one equal-sized witness micro-batch chunk ([m, k, 6] int64 per rank -> [world*m, k, 6]).
    The full gathered witness (n x 34 MB) does not fit one GPU at config-3 scale, so callers gather micro-batches
    and hand each to its consumer before the next."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * local_chunk.shape[0],) + tuple(local_chunk.shape[1:]), dtype=local_chunk.dtype, device=local_chunk.device)
    assert out.shape[0] == world * local_chunk.shape[0] and out.is_contiguous()
    dist.all_gather_into_tensor(out, local_chunk.contiguous(), group=group)
    return out


def stream_allgather(local_tensor, chunk, consume, group=None, buffer=None):
    """All-gathers `local_tensor` ([m, k, 6] per rank) in micro-batches of `chunk` rows per rank and hands every gathered
    micro-batch ([world * rows, k, 6], rank-major) to `consume` before the next one is gathered — the full gathered tensor
    never exists. `buffer` ([world * chunk, k, 6]) is reused for the full-size micro-batches.
    Returns the number of micro-batches. Used by bench.py's all-gather leg (RCCL) and by the gloo test (CPU)."""
    m = local_tensor.shape[0]
    count = 0
    for c0 in range(0, m, chunk):
        part = local_tensor[c0:c0 + chunk]
        g = all_gather_witness_chunk(part, group=group, out=buffer if (buffer is not None and part.shape[0] == chunk) else None)
        consume(g, c0, part.shape[0])
        count += 1
    return count


def stream_allgather_compact(local_compact, expand, consume, group=None, buffer=None):
    """All-gathers one batch in COMPACT wire form (WitnessEngine.submit_compact: bit-packed SHA witnesses + staged field
    witnesses, ~2.6 MB per instance instead of 34 MB) and, rank by rank, turns it back into witness vectors on the receiver:
    `expand(compact_r)` -> witness tensor (WitnessEngine.expand_compact into a scratch tensor), then `consume(witness, r)`
    before the next rank's batch is expanded — the gathered vectors never exist at once. `local_compact`: uint8 [bytes];
    `buffer`: uint8 [world, bytes], reused. Returns the number of batches consumed (= world)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    flat = local_compact.reshape(-1)
    if buffer is None:
        buffer = torch.empty((world, flat.numel()), dtype=flat.dtype, device=flat.device)
    assert buffer.shape == (world, flat.numel()) and buffer.is_contiguous()
    src, dst = flat, buffer.reshape(-1)
    if flat.dtype.itemsize == 1 and flat.numel() % 8 == 0 and flat.data_ptr() % 8 == 0 and dst.data_ptr() % 8 == 0:
        src, dst = flat.view(torch.int64), dst.view(torch.int64)  # a batch is > 2^31 bytes: keep the element count small
    dist.all_gather_into_tensor(dst, src, group=group)
    for r in range(world):
        consume(expand(buffer[r]), r)
    return world


class CompactGatherPipeline:
    """Double-buffered all-gather of batches in compact wire form: the gather of batch k + 1 (communication stream) runs while
    every rank's part of batch k is expanded and consumed (consumer stream) — two `gathered` buffers, one event pair per buffer.

        pipe = CompactGatherPipeline(world, nbytes, device, expand, consume, comm_stream=..., consumer_stream=...)
        pipe.push(local_compact, before=lambda comm: engine.wait_step(s, comm), after=lambda comm: engine.output_consumed(buf, comm))
        ...
        pipe.flush()

    push() issues the gather of its batch FIRST and only then the expansion + consumption of the previous batch, so the device
    has both in flight. `expand(compact_r)` -> witness tensor, `consume(witness, rank, batch_index)`; both run under the consumer
    stream. Both streams or neither: None, None for CPU tensors with gloo (everything is synchronous, the order of operations is the
    same); one stream without the other would leave the gather and the expansion unordered and is refused."""

    def __init__(self, world, nbytes, device, expand, consume, group=None, comm_stream=None, consumer_stream=None):
        import torch

        if (comm_stream is None) != (consumer_stream is None):
            raise ValueError("CompactGatherPipeline: pass both comm_stream and consumer_stream (they may be the same stream), or neither")
        self.torch, self.world, self.group = torch, world, group
        self.expand, self.consume = expand, consume
        self.comm, self.consumer = comm_stream, consumer_stream
        self.gathered = [torch.empty((world, nbytes), dtype=torch.uint8, device=device) for _ in range(2)]
        cuda = comm_stream is not None
        self.ev_gathered = [torch.cuda.Event() for _ in range(2)] if cuda else None
        self.ev_free = [torch.cuda.Event() for _ in range(2)] if cuda else None
        self.used = [False, False]
        self.pending = None  # (slot, batch index)
        self.k = 0
        self.order = []  # ("gather", k) / ("consume", k): what was issued, in order (tests)

    def _ctx(self, stream):
        import contextlib

        return self.torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def push(self, local_compact, before=None, after=None):
        import torch.distributed as dist

        torch = self.torch
        slot = self.k % 2
        flat = local_compact.reshape(-1)
        dst = self.gathered[slot].reshape(-1)
        src = flat
        if flat.dtype.itemsize == 1 and flat.numel() % 8 == 0 and flat.data_ptr() % 8 == 0 and dst.data_ptr() % 8 == 0:
            src, dst = flat.view(torch.int64), dst.view(torch.int64)  # a batch is > 2^31 bytes: keep the element count small
        if before is not None:
            before(self.comm)  # e.g. the engine's wait_step on the communication stream
        with self._ctx(self.comm):
            if self.comm is not None and self.used[slot]:
                self.comm.wait_event(self.ev_free[slot])  # the buffer's previous batch has been consumed
            dist.all_gather_into_tensor(dst, src, group=self.group)
            if self.comm is not None:
                self.ev_gathered[slot].record(self.comm)
        self.order.append(("gather", self.k))
        if after is not None:
            after(self.comm)  # e.g. release the local compact buffer to the engine
        prev, self.pending = self.pending, (slot, self.k)
        self.used[slot] = True
        self.k += 1
        if prev is not None:
            self._consume(*prev)

    def _consume(self, slot, index):
        with self._ctx(self.consumer):
            if self.consumer is not None:
                self.consumer.wait_event(self.ev_gathered[slot])
            for r in range(self.world):
                self.consume(self.expand(self.gathered[slot][r]), r, index)
            if self.consumer is not None:
                self.ev_free[slot].record(self.consumer)
        self.order.append(("consume", index))

    def flush(self):
        if self.pending is not None:
            prev, self.pending = self.pending, None
            self._consume(*prev)
