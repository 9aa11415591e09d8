"""Data-parallel layer for MMVit4: one process per GPU, bucketed gradient all-reduce over RCCL (xGMI).

The reference is single-process (no NCCL/MPI/DDP anywhere), so this is new surface: it reproduces what
wrapping the reference in DistributedDataParallel would compute - every rank runs the reference semantics on its own
shard of the batch (BatchNorm statistics and the inter-modal re-view stay per-rank) and parameter
gradients are averaged across ranks.

Design for xGMI (point-to-point links, ring collectives are per-link bound): few large buckets (default 64 MB) issued in
reverse-forward order from autograd post-accumulate hooks on a side stream, so the all-reduce of the decoder's gradients
overlaps the encoders' backward; the parameters that never receive a gradient (the model names them: `NOGRAD_PREFIXES`,
18 tensors for MMVit4) are never communicated; gradients live as views of the flat buckets so no flatten/unflatten copies
are made.  Buckets exist before the first backward (the grad-less set is known from the model), so step 1 overlaps like
every other step; collectives are issued strictly in bucket-index order on every rank (a bucket that completes early waits
for its predecessors), so the order never depends on the autograd engine's scheduling; the 1/world averaging runs on the
communication stream right behind each all-reduce (own kernel, no ATen launch).

BatchNorm buffers (running statistics, counters) stay per rank during training; `sync_buffers` / `save_checkpoint`
implement the SURVEY section 8(e) policy: rank 0's buffers are broadcast before a checkpoint is written, and rank 0 writes it.
"""
import torch
import torch.distributed as dist


def _scale_(flat, alpha):
    if flat.is_cuda:
        import corrif_hip as H
        H.check(H.lib().corrif_scale(flat.data_ptr(), flat.data_ptr(), flat.numel(), alpha, H.stream()), "corrif_scale")
    else:                                  # gloo rehearsal on CPU tensors (tests): host arithmetic
        flat.mul_(alpha)


def _zero_(flat):
    if flat.is_cuda:
        import corrif_hip as H
        H.check(H.lib().corrif_fill(flat.data_ptr(), flat.numel(), 0.0, H.stream()), "corrif_fill")
    else:
        flat.zero_()


class GradAllReducer:
    def __init__(self, model, bucket_bytes=64 << 20, process_group=None, overlap=True, force_collective=False, skip_prefixes=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = force_collective and dist.is_initialized()      # run the collectives even with one rank (single-GPU rehearsal)
        self.active = self.world > 1 or self.force
        self.overlap = overlap
        self._limit = max(1, bucket_bytes // 4)
        if skip_prefixes is None:
            skip_prefixes = getattr(model, "NOGRAD_PREFIXES", None)
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
        self.buckets = None
        self._hooks = []
        self._stream = None
        self._next = 0
        if not self.active:
            # one rank, no collective to run: no buckets at all.  zero_grad() drops the gradients (p.grad = None), so autograd's
            # AccumulateGrad adopts each freshly produced gradient tensor instead of adding it into a zeroed bucket view - that add was
            # one stock ATen launch per parameter and step (645 of them in the single-GPU step of round 1).
            return
        if skip_prefixes is not None:        # the grad-less set is known: buckets (and hooks) exist before the first backward
            self._build([p for n, p in named if not n.startswith(tuple(skip_prefixes))])
        # else: built lazily after the first backward, from the parameters that did receive a gradient

    # ---------------------------------------------------------------- bucket construction
    def _build(self, live):
        live = list(live)
        live.reverse()               # reverse registration order ~ order in which backward produces them
        self.buckets, cur, size = [], [], 0
        for p in live:
            cur.append(p)
            size += p.numel()
            if size >= self._limit:
                self.buckets.append(self._make_bucket(cur))
                cur, size = [], 0
        if cur:
            self.buckets.append(self._make_bucket(cur))
        self._index = {}
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._index[p] = bi
        if self.active and self.overlap and live and live[0].is_cuda:
            self._stream = torch.cuda.Stream()
            for p in live:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _make_bucket(self, params):
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        o = 0
        for p in params:
            v = flat[o:o + p.numel()].view_as(p)
            if p.grad is not None:
                v.copy_(p.grad)
            p.grad = v               # gradient now lives inside the bucket (autograd accumulates in place)
            o += p.numel()
        return {"params": params, "flat": flat, "ready": 0, "work": None, "events": [], "launched": False}

    # ---------------------------------------------------------------- hooks / reduction
    def _on_grad(self, p):
        """post-accumulate hook (only installed when collectives will run): count the bucket's parameters; remember on which
        stream each gradient was produced (the model runs its modality branches / sample-group lanes on several streams)"""
        b = self.buckets[self._index[p]]
        b["ready"] += 1
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        b["events"].append(ev)
        self._launch_ready()

    def _launch_ready(self):
        """issue every complete bucket whose predecessors have all been issued: strict index order on every rank"""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b["ready"] < len(b["params"]):
                return
            self._launch(b)
            self._next += 1

    def _launch(self, b):
        b["launched"] = True
        if not self.active:
            b["events"] = []
            return
        alpha = 1.0 / self.world
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            for ev in b["events"]:
                self._stream.wait_event(ev)
            b["events"] = []
            with torch.cuda.stream(self._stream):
                b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.world > 1:
                    b["work"].wait()             # stream-level dependency only (does not block the host for RCCL)
                    _scale_(b["flat"], alpha)    # average on the communication stream, behind the collective
                    b["scaled"] = True
        else:
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def zero_grad(self):
        """keeps the bucket views alive (set_to_none would drop them)"""
        if self.buckets is None or not self.active:
            for p in self.params:
                p.grad = None
            return
        for b in self.buckets:
            _zero_(b["flat"])
            b["ready"] = 0
            b["work"] = None
            b["events"] = []
            b["launched"] = False
            b["scaled"] = False
        self._next = 0

    def finish(self):
        """call after backward(): issues whatever the hooks have not issued (in index order), waits, averages.  After it returns
        every rank holds the mean gradient in p.grad."""
        if not self.active:
            return
        if self.buckets is None:
            self._build([p for p in self.params if p.grad is not None])
        for b in self.buckets:           # first lazy step / overlap disabled / a bucket whose hooks did not all fire
            if not b["launched"]:
                self._launch(b)
        self._next = len(self.buckets)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
            if self.world > 1 and not b.get("scaled"):
                _scale_(b["flat"], 1.0 / self.world)
                b["scaled"] = True
            b["ready"] = 0
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def communicated_elements(self):
        return 0 if self.buckets is None else sum(b["flat"].numel() for b in self.buckets)


def broadcast_module_state(model, src=0, group=None):
    """identical initial parameters and buffers on every rank (rank `src` wins)"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src, group=group)


def sync_buffers(model, src=0, group=None):
    """SURVEY section 8(e) buffer policy: BatchNorm running statistics / counters are per rank during training (as under
    DistributedDataParallel with broadcast_buffers=False); before a checkpoint is taken every rank adopts rank `src`'s buffers"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in model.buffers():            # MMVit4.buffers() brings the host-side BatchNorm counters up to date first
        dist.broadcast(t.data, src, group=group)


def save_checkpoint(model, path, src=0, group=None):
    """`torch.save(model.state_dict(), path)` (F4_TRAIN.py:84,86) for a data-parallel job: buffers synchronised from rank `src`,
    written by rank `src` only, every rank returns after the file exists"""
    sync_buffers(model, src, group)
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if rank == src:
        torch.save(model.state_dict(), path)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group=group)


def shard_batch(x, rank, world):
    """rank r takes samples [r*b, (r+1)*b) of the global batch"""
    b = x.shape[0] // world
    return x[rank * b:(rank + 1) * b]
