"""Data-parallel layer for MMVit4: one process per GPU, bucketed gradient all-reduce over RCCL (xGMI).

The reference is single-process (no NCCL/MPI/DDP anywhere), so this is new surface: it reproduces what
wrapping the reference in DistributedDataParallel would compute - every rank runs the reference semantics on its own
shard of the batch (BatchNorm statistics and the inter-modal re-view stay per-rank) and parameter
gradients are averaged across ranks.

Design for xGMI (point-to-point links, ring collectives are per-link bound): few large buckets (default 64 MB) issued in
reverse-forward order from autograd post-accumulate hooks on a side stream, so the all-reduce of the decoder's gradients
overlaps the encoders' backward; the 18 parameters that never receive a gradient are never communicated; gradients live
as views of the flat buckets so no flatten/unflatten copies are made.
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, model, bucket_bytes=64 << 20, process_group=None, overlap=True, force_collective=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = force_collective and dist.is_initialized()      # run the collectives even with one rank (single-GPU rehearsal)
        self.overlap = overlap
        self._limit = max(1, bucket_bytes // 4)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.buckets = None          # built lazily after the first backward (only then is the grad-less set known)
        self._pending = []
        self._hooks = []
        self._stream = None

    # ---------------------------------------------------------------- bucket construction
    def _build(self):
        live = [p for p in self.params if p.grad is not None]
        live.reverse()               # reverse registration order ~ order in which backward produces them
        self.buckets, cur, size = [], [], 0
        limit = self._bucket_elems()
        for p in live:
            cur.append(p)
            size += p.numel()
            if size >= limit:
                self.buckets.append(self._make_bucket(cur))
                cur, size = [], 0
        if cur:
            self.buckets.append(self._make_bucket(cur))
        self._index = {}
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._index[p] = bi
        if self.overlap and live and live[0].is_cuda:
            self._stream = torch.cuda.Stream()
            for p in live:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _bucket_elems(self):
        return self._limit

    def _make_bucket(self, params):
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        views, o = [], 0
        for p in params:
            v = flat[o:o + p.numel()].view_as(p)
            v.copy_(p.grad)
            p.grad = v               # gradient now lives inside the bucket (autograd accumulates in place)
            views.append(v)
            o += p.numel()
        return {"params": params, "flat": flat, "ready": 0, "work": None, "events": []}

    # ---------------------------------------------------------------- hooks / reduction
    def _on_grad(self, p):
        b = self.buckets[self._index[p]]
        b["ready"] += 1
        if self._stream is not None:
            # gradients of one bucket may be accumulated on different streams (the model runs its three modality branches
            # on three streams): remember where each one was produced
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            b["events"].append(ev)
        if b["ready"] == len(b["params"]):
            self._launch(b)

    def _launch(self, b):
        if self.world == 1 and not self.force:
            b["events"] = []
            return
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            for ev in b["events"]:
                self._stream.wait_event(ev)
            b["events"] = []
            with torch.cuda.stream(self._stream):
                b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def zero_grad(self):
        """keeps the bucket views alive (set_to_none would drop them)"""
        if self.buckets is None:
            for p in self.params:
                p.grad = None
            return
        for b in self.buckets:
            b["flat"].zero_()
            b["ready"] = 0
            b["work"] = None
            b["events"] = []

    def finish(self):
        """call after backward(): waits for / issues the all-reduces and averages.  After it returns every rank holds the
        mean gradient in p.grad."""
        first = self.buckets is None
        if first:
            self._build()
        for b in self.buckets:
            if (self.world > 1 or self.force) and (b["work"] is None):
                self._launch(b)      # first step (hooks not yet installed) or overlap disabled
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
            if self.world > 1 or self.force:
                if self._stream is not None:
                    torch.cuda.current_stream().wait_stream(self._stream)
                b["flat"].mul_(1.0 / self.world)
            b["ready"] = 0

    def communicated_elements(self):
        return 0 if self.buckets is None else sum(b["flat"].numel() for b in self.buckets)


def broadcast_module_state(model, src=0, group=None):
    """identical initial parameters and buffers on every rank (rank `src` wins)"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src, group=group)


def shard_batch(x, rank, world):
    """rank r takes samples [r*b, (r+1)*b) of the global batch"""
    b = x.shape[0] // world
    return x[rank * b:(rank + 1) * b]
