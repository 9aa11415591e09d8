"""Training / evaluation harness around the MI355X `MMVit4`: the callers of the hot path.

Restates the loops of the reference (which cannot be imported: they pull in modules that are not in the repository):
  * `train_epoch`  - F4_TRAIN.py:41-86: `scheduler.step()` BEFORE the epoch's optimiser steps (reference quirk, F4_TRAIN.py:46),
                     per batch `zero_grad -> model -> BCEWithLogitsLoss (on the sigmoided output) -> backward -> Adam.step`,
                     `loss.item()` per step, `jI += Jaccard2(mask[:,0], pred[:,0]) * n`, checkpoint per epoch (F4_TRAIN.py:84).
  * `evaluate`     - F4_TRAIN.py:181-208 / F7_TEST2.py:131-184: `model.eval()`, `torch.no_grad()`, BatchNorm running statistics.
  * `per_image_metrics` - allJaccardResults_irem_f1_jcrd.py:201-222: Jaccard2 / F1 per image at batch 1.
Loss, metric and the Adam update run as gfx950 kernels (ops.bce_with_logits_mean, ops.jaccard_all, corrif_adam_multi).
"""
import struct

import torch

import mmvit4
import ops
from corrif_hip import check, lib, stream


class FusedAdam:
    """torch.optim.Adam(params, lr) semantics (betas 0.9/0.999, eps 1e-8, no amsgrad; F2_MAIN.py:168-169) in ONE launch per step.
    Parameters whose .grad is None are skipped, exactly like torch.optim.Adam (the 18 grad-less tensors of MMVit4)."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.state = {}
        self.t = 0
        self._sig = self._ptr_sig = None

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def _tables(self, live):
        """Device tables of the one-launch update.  The block lists (which 1024-element block belongs to which tensor) depend on the
        SET of live parameters only and are built once; the 40-byte pointer records are refreshed whenever a gradient lives at a new
        address (`zero_grad()` drops the gradients like torch.optim's set_to_none, so the allocator may hand out other blocks next
        step): 1122 records = 45 kB through a pinned staging buffer, instead of re-building an 83k-entry block table per step."""
        dev = live[0].device
        shape_sig = tuple(id(p) for p in live)
        if shape_sig != self._sig:
            bt, bo = [], []
            for ti, p in enumerate(live):
                if p not in self.state:
                    self.state[p] = (torch.zeros_like(p), torch.zeros_like(p))
                nb = (p.numel() + 1023) // 1024
                bt.append(torch.full((nb,), ti, dtype=torch.int32))
                bo.append(torch.arange(nb, dtype=torch.int64) * 1024)
            self._bt = torch.cat(bt).to(dev)
            self._bo = torch.cat(bo).to(dev)
            self._host = [torch.empty(40 * len(live), dtype=torch.uint8).pin_memory() for _ in range(2)]     # staging, double-buffered
            self._host_ev = [None, None]
            self._flip = 0
            self._table = torch.empty(40 * len(live), dtype=torch.uint8, device=dev)
            self._sig, self._ptr_sig = shape_sig, None
        ptr_sig = tuple((p.data_ptr(), p.grad.data_ptr()) for p in live)
        if ptr_sig != self._ptr_sig:
            raw = bytearray()
            for p in live:
                st = self.state[p]
                raw += struct.pack("<QQQQq", p.data_ptr(), p.grad.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), p.numel())
            k = self._flip = self._flip ^ 1
            if self._host_ev[k] is not None:
                self._host_ev[k].synchronize()          # the copy that last read this staging buffer (two steps ago) has completed
            self._host[k].copy_(torch.frombuffer(raw, dtype=torch.uint8))
            self._table.copy_(self._host[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._host_ev[k] = ev
            self._ptr_sig = ptr_sig

    @torch.no_grad()
    def step(self):
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        for p in live:
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
        self.t += 1
        self._tables(live)
        check(lib().corrif_adam_multi(self._table.data_ptr(), self._bt.data_ptr(), self._bo.data_ptr(), self._bt.numel(), self.lr,
                                      self.betas[0], self.betas[1], self.eps, self.wd, self.t, stream()), "corrif_adam_multi")


class StepLR:
    """torch.optim.lr_scheduler.StepLR(step_size, gamma) driving FusedAdam.lr"""

    def __init__(self, optim, step_size, gamma):
        self.optim, self.step_size, self.gamma, self.base, self.epoch = optim, step_size, gamma, optim.lr, 0

    def step(self):
        self.epoch += 1
        self.optim.lr = self.base * self.gamma ** (self.epoch // self.step_size)

    def get_lr(self):
        return [self.optim.lr]


def train_step(model, optim, images, masks, reducer=None):
    """F4_TRAIN.py:54-71 for one batch; returns (loss, Jaccard2 * n, n) as device tensors / ints, no host sync"""
    (reducer or optim).zero_grad()
    pred = model(images)
    loss = ops.bce_with_logits_mean(pred, masks)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    optim.step()
    n = masks.shape[0] * masks.shape[-1] * masks.shape[-2]
    jac = mmvit4.Jaccard2(masks[:, 0].reshape(n, 1), pred.detach()[:, 0].reshape(n, 1)) * n
    return loss.detach(), jac, n


def checkpoint_paths(pathm, i):
    """the reference's checkpoint files: `iremmodel{i}.pt` rewritten after every epoch (F4_TRAIN.py:84) and
    `Finaliremmodel{i}.pt` after the last one (F4_TRAIN.py:86); `validate` / `test_model` reload them (F4_TRAIN.py:180, F7_TEST2.py:126)"""
    import os
    return os.path.join(pathm, "iremmodel{}.pt".format(i)), os.path.join(pathm, "Finaliremmodel{}.pt".format(i))


def train_model(n_epochs, model, scheduler, training_generator, optim, device, pathm, i, validation_generator=None, reducer=None):
    """F4_TRAIN.py:39-86 restated: per epoch `train_epoch` (scheduler first), the per-epoch checkpoint `iremmodel{i}.pt`, validation;
    `Finaliremmodel{i}.pt` at the end.  In a data-parallel job rank 0's BatchNorm buffers are broadcast before each save and rank 0
    writes the file (data_parallel.save_checkpoint).  Returns the per-epoch (train loss, train Jaccard, val loss,
    val Jaccard) the reference writes to its log files."""
    from data_parallel import save_checkpoint
    per_epoch, final = checkpoint_paths(pathm, i)
    log = []
    for _ in range(n_epochs):
        tl, tj = train_epoch(model, optim, scheduler, training_generator, device, reducer)
        save_checkpoint(model, per_epoch)
        vl = vj = None
        if validation_generator is not None:
            vl, vj = evaluate(model, validation_generator, device)
        log.append((tl, tj, vl, vj))
    save_checkpoint(model, final)
    return log


def train_epoch(model, optim, scheduler, loader, device, reducer=None, checkpoint=None):
    model.train()
    scheduler.step()                                   # before the optimiser, as the reference does (F4_TRAIN.py:46)
    losses, jI, total = [], 0.0, 0
    for images, masks in loader:
        loss, jac, n = train_step(model, optim, images.to(device), masks.to(device), reducer)
        losses.append(loss.item())                     # the reference syncs here every step too (F4_TRAIN.py:64)
        jI += jac.item()
        total += n
    if checkpoint:
        from data_parallel import save_checkpoint
        save_checkpoint(model, checkpoint)             # F4_TRAIN.py:84 (rank 0's buffers, written by rank 0)
    return sum(losses) / max(len(losses), 1), jI / max(total, 1)


@torch.no_grad()
def evaluate(model, loader, device):
    """validation / test pass: mean loss and batch-weighted soft Jaccard (F4_TRAIN.py:181-208)"""
    model.eval()
    losses, jI, total = [], 0.0, 0
    for images, masks in loader:
        images, masks = images.to(device), masks.to(device)
        pred = model(images)
        losses.append(ops.bce_with_logits_mean(pred, masks).item())
        n = masks.shape[0] * masks.shape[-1] * masks.shape[-2]
        jI += (mmvit4.Jaccard2(masks[:, 0].reshape(n, 1), pred[:, 0].reshape(n, 1)) * n).item()
        total += n
    return sum(losses) / max(len(losses), 1), jI / max(total, 1)


class GraphedForward:
    """Eval-mode forward captured ONCE per input shape in a HIP graph and replayed (the per-image metric loop runs at batch 1,
    where the ~2000 launches of a forward cost more host time than the GPU needs: a step is launch-bound, not MFMA-bound).
    The capture takes the module's real schedule - the three modality-branch streams and the decoder's skip stream fork from the
    capturing stream and re-join it (the sample-group lanes are left out of a capture: mmvit4._run_lanes) - so the replay keeps their
    concurrency (1.9x faster than eager at
    batch 1 in round 1, 1.24x for a single-stream capture).  Round 1 had to capture single-stream because the process crashed after
    a multi-stream capture; the cause was events destroyed while their stream was still capturing (mmvit4._Edges explains it) and
    is fixed by the model's persistent fork/join events.  `single_stream=True` still forces the one-stream schedule.
    Results are bit-identical to the eager forward."""

    def __init__(self, model, example, warmup=2, single_stream=False):
        model.eval()
        self.model = model
        self.static_in = example.detach().clone()
        saved = []
        if single_stream:
            for obj, name, off in ((model, "concurrent_branches", False), (model, "decoder_split", 0),
                                   (getattr(model, "decoder_fuse", None), "concurrent_skips", False)):
                if obj is not None and hasattr(obj, name):
                    saved.append((obj, name, getattr(obj, name)))
                    setattr(obj, name, off)
        try:
            side = torch.cuda.Stream(device=example.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():      # lazy initialisation (kernel attributes, streams, fork/join events, allocator pool) outside the capture
                for _ in range(warmup):
                    model(self.static_in)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.static_out = model(self.static_in)
        finally:
            for obj, name, val in saved:
                setattr(obj, name, val)

    def __call__(self, x):
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out


@torch.no_grad()
def per_image_metrics(model, images, masks):
    """Jaccard2 and F1 per image at batch 1 (allJaccardResults_irem_f1_jcrd.py:201-222); returns two lists of floats"""
    model.eval()
    js, fs = [], []
    for i in range(images.shape[0]):
        pred = model(images[i:i + 1])
        n = masks.shape[-1] * masks.shape[-2]
        out = ops.jaccard_all(masks[i, 0].reshape(n, 1), pred[0, 0].reshape(n, 1))
        js.append(out[0].item())
        fs.append(out[2].item())
    return js, fs
