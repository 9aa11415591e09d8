"""Throughput of the MMVit4 hot path (forward + loss + backward) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One JSON line on rank 0.  metric = images/s for one forward + backward of MMVit4 (BASELINE.json), workload = config 2:
4 bands per modality group, 224x224, batch 32 per GPU, fp32, train mode (batch-stat BatchNorm, dropout on), synthetic
inputs resident in HBM, random-init weights.  N > 1: weak scaling, one replica per GPU on its own 32-image shard, gradients
averaged with bucketed RCCL all-reduce (inside the timed region).

roofline: the dominant kernel family is the fp32 MFMA implicit-GEMM / patch-conv family; `achieved` is the algorithmic FLOPs
of those launches (2*M*N*K each, from the launch descriptors) divided by their summed duration measured with HIP events
around every launch, in extra steps right after the timed region with the three modality branches serialised (the timed
region itself runs them on three concurrent streams, where an event pair would also span other streams' kernels);
`whole_step_frac` prices the WHOLE timed step (645.9 GFLOP/image) against the same peak = 157.3 TFLOP/s (fp32 matrix).
cpu_baseline: the CPU oracle (stock-PyTorch restatement, bit-identical to the reference on CPU) timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "corrifnet-correlation-aware-interactive-fusion-multimodal-learning-for-multispectral-images_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA (MI355X_MICROARCH.md): the instruction peak of the split-bf16 main loops
FLOP_PER_IMAGE_FWD_BWD = 645.9e9          # 215.3 GFLOP forward x 3 at D=4, 224^2


class MfmaTimer:
    """records (flops, start, stop) for every MFMA GEMM launch by wrapping ops.gemm / ops.wgrad"""

    def __init__(self, ops):
        self.ops, self.rec, self.on = ops, [], False
        self._gemm, self._wgrad, self._patch, self._pwg = ops.gemm, ops.wgrad, ops.conv3_patch, ops.conv3_patch_wgrad
        ops.gemm, ops.wgrad, ops.conv3_patch, ops.conv3_patch_wgrad = self.gemm, self.wgrad, self.patch, self.patch_wgrad
        self._ffwd, self._fbwd = ops.flash_fwd, ops.flash_bwd
        ops.flash_fwd, ops.flash_bwd = self.flash_fwd, self.flash_bwd
        self._sfwd, self._swg = ops.stem_fwd, ops.stem_wgrad
        ops.stem_fwd, ops.stem_wgrad = self.stem_fwd, self.stem_wgrad

    def _timed(self, fn, flops, key, a, kw):
        if not self.on:
            return fn(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.ops.LAST_SPLIT = 0
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        # split: the launch ran the split-bf16 main loop (six bf16 MFMAs per fp32 product) - only gemm / wgrad launches can
        self.rec.append((flops, e0, e1, key, bool(self.ops.LAST_SPLIT) and key[0] in ("gemm", "conv", "dgrad", "wgrad")))
        return r

    def gemm(self, *a, **kw):       # (A, lda, B, ldb, b_layout, C, ldc, M, N, K, Cs, geom, ...)
        key = ("gemm" if a[11].is_gemm else ("dgrad" if a[11].dir < 0 else "conv"), a[7], a[8], a[9], kw.get("Z", 1))
        return self._timed(self._gemm, 2.0 * a[7] * a[8] * a[9] * kw.get("Z", 1), key, a, kw)

    def wgrad(self, *a, **kw):      # (A, lda, B, ldb, Cs, C, ldc, R, M, N, geom, ...)
        key = ("wgrad", a[7], a[8], a[9], kw.get("Z", 1))
        return self._timed(self._wgrad, 2.0 * a[7] * a[8] * a[9] * kw.get("Z", 1), key, a, kw)

    def patch(self, *a, **kw):      # (x, ldx, wp, y, ldy, bias, B, S, O, Ci, Co, pad, clamp, cc)
        M = a[6] * a[8][0] * a[8][1] * a[8][2]
        key = ("patch_dgrad" if a[11] == 2 or a[5] == 0 else "patch_conv", M, a[10], 27 * a[9], 1)
        return self._timed(self._patch, 2.0 * M * a[10] * 27 * a[9], key, a, kw)

    def patch_wgrad(self, *a, **kw):   # (x, ldx, gy, ldg, gwp, B, S, O, Ci, Co, clamp, dev)
        M = a[5] * a[7][0] * a[7][1] * a[7][2]
        return self._timed(self._pwg, 2.0 * M * a[9] * 27 * a[8], ("patch_wgrad", M, a[9], 27 * a[8], 1), a, kw)

    # flash attention: ALGORITHMIC products only (q.k^T and p.v forward; dv, dp, dq, dk backward = 2 + 4 products of 2*N*N*64 flop per
    # head); the backward kernels recompute q.k^T twice and dp once on top of that, which is not credited
    def flash_fwd(self, *a, **kw):   # (qkv, out, lse, mask, B, N, heads, ...)
        fl = 2 * 2.0 * a[4] * a[6] * a[5] * a[5] * 64
        return self._timed(self._ffwd, fl, ("flash_fwd", a[5], a[5], 64, a[4] * a[6]), a, kw)

    def flash_bwd(self, *a, **kw):   # (qkv, out, lse, mask, go, dvec, dqkv, B, N, heads, ...)
        fl = 4 * 2.0 * a[7] * a[9] * a[8] * a[8] * 64
        return self._timed(self._fbwd, fl, ("flash_bwd", a[8], a[8], 64, a[7] * a[9]), a, kw)

    def stem_fwd(self, *a, **kw):    # (x, batch_pitch, wp, y, ldy, B, D, H, W)
        M = a[5] * a[6] * ((a[7] - 1) // 2 + 1) * ((a[8] - 1) // 2 + 1)
        return self._timed(self._sfwd, 2.0 * M * 64 * 147, ("stem_conv", M, 64, 147, 1), a, kw)

    def stem_wgrad(self, *a, **kw):  # (x, batch_pitch, gy, ldg, gwp, B, D, H, W)
        M = a[5] * a[6] * ((a[7] - 1) // 2 + 1) * ((a[8] - 1) // 2 + 1)
        return self._timed(self._swg, 2.0 * M * 64 * 147, ("stem_wgrad", M, 64, 147, 1), a, kw)

    def by_shape(self):
        agg = {}
        for fl, e0, e1, key, _sp in self.rec:
            d = agg.setdefault(key, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
            d[2] += fl
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        return [{"kind": k[0], "M_or_R": k[1], "N_or_M": k[2], "K_or_N": k[3], "Z": k[4], "calls": v[0], "ms": round(v[1], 3),
                 "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2) if v[1] > 0 else 0} for k, v in rows]

    @staticmethod
    def pair_overhead_ms(n=200):
        """elapsed time an EMPTY event pair reports on the busy stream's queue (two barrier packets): subtracted per launch, so the
        sum of launch durations matches rocprofv3's kernel durations instead of carrying ~10 us of marker time per launch"""
        torch.cuda.synchronize()
        ds = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            ds.append((e0, e1))
        torch.cuda.synchronize()
        v = sorted(a.elapsed_time(b) for a, b in ds)
        return v[len(v) // 2]

    def by_kind(self, overhead_ms=0.0, steps=1):
        """per kernel family: ms per step and achieved TFLOP/s (conv / dgrad / gemm = gemm_fwd_kernel in its three roles)"""
        agg = {}
        for fl, e0, e1, key, _sp in self.rec:
            d = agg.setdefault(key[0], [0.0, 0.0, 0])
            d[0] += max(e0.elapsed_time(e1) - overhead_ms, 0.0)
            d[1] += fl
            d[2] += 1
        return {k: {"ms_per_step": round(v[0] / steps, 2), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] > 0 else 0.0,
                    "launches_per_step": v[2] // steps} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}

    def summary(self, overhead_ms=0.0):
        fl = sum(r[0] for r in self.rec)
        ms = sum(max(r[1].elapsed_time(r[2]) - overhead_ms, 0.0) for r in self.rec)
        return fl, ms, len(self.rec)

    def executed(self, overhead_ms=0.0):
        """The matrix-core work as ISSUED: a split launch executes 6 bf16 MFMA products per fp32 product (dense bf16 peak 2500 TFLOP/s), the
        others v_mfma_f32_* (157.3).  Returns (fp32-equivalent flops of split launches, their ms, flops of fp32-input launches, their ms)."""
        fs = ms_s = ff = ms_f = 0.0
        for fl, e0, e1, key, sp in self.rec:
            t = max(e0.elapsed_time(e1) - overhead_ms, 0.0)
            if sp:
                fs += fl; ms_s += t
            else:
                ff += fl; ms_f += t
        return fs, ms_s, ff, ms_f


class HbmTimer:
    """per-launch HIP-event timing of the HBM-bound kernel families (normalisation passes, resampling, pooling, the inter-modal
    correlation, dropout, ...) in the same single-stream pass MfmaTimer uses: a proxy in front of the ctypes library records (bytes,
    start, stop) for the entry points below.  `bytes` = ALGORITHMIC bytes of the call (every tensor it has to read or write, once), so
    GB/s against the 8 TB/s HBM3E peak is the roofline fraction of these kernels (BASELINE north_star: "achieved HBM GB/s")."""

    @staticmethod
    def _n(v):
        return 0 if v is None else int(getattr(v, "value", v) or 0)

    def __init__(self, H):
        n = self._n
        self.H, self.real, self.rec, self.on = H, H.lib(), [], False

        def norm_apply(a):       # (x, ldx, mean, rstd, gamma, beta, residual, ldr, y, ldy, rpg, G, C, ...)
            return 4 * n(a[10]) * n(a[11]) * n(a[12]) * (2 + (1 if n(a[6]) else 0))

        def norm_bwd(a, pre):    # (dy, lddy, y, ldy, x, ldx, mean, rstd, gamma, dx, lddx, dres, lddres, dgamma, dbeta, rpg, G, C, ...)
            if pre and len(a) == 22:                        # corrif_norm_bwd_pre: (..., rows, C, flags, part, chunks, ws, stream)
                e = n(a[15]) * n(a[16])
            else:
                e = n(a[15]) * n(a[16]) * n(a[17])
            y = 1 if n(a[2]) else 0
            apply_ = 2 + y + (1 if n(a[9]) else 0) + (1 if n(a[11]) else 0)
            return 4 * e * (apply_ + (0 if pre else 2 + y))

        def resample(a):         # (x, ldx, y, ldy, B, C, Di, Hi, Wi, Do, Ho, Wo, stream)
            return 4 * n(a[4]) * n(a[5]) * (n(a[6]) * n(a[7]) * n(a[8]) + n(a[9]) * n(a[10]) * n(a[11]))

        self.models = {
            "corrif_norm_apply": ("norm_apply", norm_apply), "corrif_norm_apply_g": ("norm_apply", norm_apply),
            "corrif_norm_bwd": ("norm_bwd", lambda a: norm_bwd(a, False)), "corrif_norm_bwd_g": ("norm_bwd", lambda a: norm_bwd(a, False)),
            "corrif_norm_bwd_pre": ("norm_bwd", lambda a: norm_bwd(a, True)), "corrif_norm_bwd_pre_g": ("norm_bwd", lambda a: norm_bwd(a, True)),
            "corrif_norm_stats": ("norm_stats", lambda a: 4 * n(a[2]) * n(a[3]) * n(a[4])),
            "corrif_norm_stats_g": ("norm_stats", lambda a: 4 * n(a[2]) * n(a[3]) * n(a[4])),
            "corrif_trilinear_fwd": ("trilinear", resample), "corrif_trilinear_bwd": ("trilinear", resample),
            "corrif_nearest_fwd": ("nearest", resample), "corrif_nearest_bwd": ("nearest", resample),
            "corrif_maxpool133_fwd": ("maxpool", lambda a: n(a[3]) * n(a[4]) * n(a[7]) * (4 * n(a[5]) * n(a[6]) + 5 * ((n(a[5]) - 1) // 2 + 1) * ((n(a[6]) - 1) // 2 + 1))),
            "corrif_maxpool133_bwd": ("maxpool", lambda a: n(a[3]) * n(a[4]) * n(a[7]) * (4 * n(a[5]) * n(a[6]) + 5 * ((n(a[5]) - 1) // 2 + 1) * ((n(a[6]) - 1) // 2 + 1))),
            "corrif_intercorr_fwd": ("inter_corr", lambda a: 4 * 12 * n(a[8]) * n(a[9]) * n(a[10])),
            "corrif_intercorr_bwd": ("inter_corr", lambda a: 4 * 21 * n(a[11]) * n(a[12]) * n(a[13])),
            "corrif_dropout": ("dropout", lambda a: 8 * n(a[2])),
            "corrif_add": ("add", lambda a: 12 * n(a[3])),
            "corrif_copy2d": ("copy", lambda a: 8 * n(a[4]) * n(a[5])),
            "corrif_layernorm_fwd": ("layernorm", lambda a: 4 * n(a[9]) * n(a[10]) * (2 + (1 if n(a[3]) else 0))),
            "corrif_layernorm_bwd": ("layernorm", lambda a: 12 * n(a[9]) * n(a[10])),
            "corrif_gelu_fwd": ("gelu", lambda a: 8 * n(a[2])), "corrif_gelu_bwd": ("gelu", lambda a: 12 * n(a[3])),
            "corrif_depth_bcast_add": ("depth_class", lambda a: 8 * n(a[4]) * n(a[5]) * n(a[6]) * n(a[7])),
            "corrif_depth_class_reduce": ("depth_class", lambda a: 4 * n(a[4]) * n(a[5]) * n(a[6]) * n(a[7])),
            "corrif_conv1x1_small_fwd": ("conv1x1_small", lambda a: 4 * n(a[7]) * (n(a[8]) + n(a[9]))),
            "corrif_head_fwd": ("head", lambda a: 4 * n(a[4]) * n(a[5]) * 11), "corrif_head_bwd": ("head", lambda a: 4 * n(a[8]) * n(a[9]) * 22),
        }
        outer = self

        class Proxy:
            def __getattr__(self, name):
                fn = getattr(outer.real, name)
                m = outer.models.get(name)
                if m is None:
                    return fn

                def timed(*a):
                    if not outer.on:
                        return fn(*a)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    r = fn(*a)
                    e1.record()
                    outer.rec.append((m[0], m[1](a), e0, e1))
                    return r
                return timed
        H._lib = Proxy()

    def close(self):
        self.H._lib = self.real

    def by_family(self, overhead_ms, steps):
        agg = {}
        for fam, nbytes, e0, e1 in self.rec:
            d = agg.setdefault(fam, [0.0, 0.0, 0])
            d[0] += max(e0.elapsed_time(e1) - overhead_ms, 0.0)
            d[1] += nbytes
            d[2] += 1
        out = {k: {"ms_per_step": round(v[0] / steps, 2), "GB_per_step": round(v[1] / steps / 1e9, 2),
                   "GBps": round(v[1] / (v[0] * 1e-3) / 1e9, 0) if v[0] > 0 else 0.0,
                   "frac_of_8TBps": round(v[1] / (v[0] * 1e-3) / 8e12, 3) if v[0] > 0 else 0.0, "launches_per_step": v[2] // steps}
               for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}
        t = sum(v[0] for v in agg.values())
        b = sum(v[1] for v in agg.values())
        out["_all"] = {"ms_per_step": round(t / steps, 2), "GB_per_step": round(b / steps / 1e9, 1),
                       "GBps": round(b / (t * 1e-3) / 1e9, 0) if t > 0 else 0.0, "frac_of_8TBps": round(b / (t * 1e-3) / 8e12, 3) if t > 0 else 0.0}
        return out


def helpers_free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(B=4, D=4, HW=224, iters=3):
    """Bounded sample of the same workload on the host cores (SURVEY section 8d / BASELINE.md section 3): the oracle's forward + loss +
    backward at batch 4 (= BASELINE configs[0], the reference's own CPU-runnable case; 18.5 GB of host memory), train mode with
    dropout on, 1 warm-up + `iters` timed steps, best and median."""
    from oracle import mmvit4_oracle as O
    import helpers
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(cores, int(os.environ.get("CORRIF_CPU_THREADS", "16"))))   # a 1-GPU box owns a 16-core CPU share
    torch.set_num_threads(cores)
    model_name = _cpu_model()
    print("[bench] timing the CPU oracle on %d host threads of '%s' (B=%d, %d fwd+bwd steps) ..." % (cores, model_name, B, iters + 1),
          file=sys.stderr, flush=True)
    torch.manual_seed(0)
    m = O.MMVit4().train()
    x, mask = helpers.make_inputs(B, D, HW, HW)
    ts = []
    for i in range(iters + 1):
        t0 = time.time()
        m.zero_grad(set_to_none=True)
        O.train_step_loss(m(x), mask).backward()
        ts.append(time.time() - t0)
        print("[bench]   cpu step %d: %.1f s" % (i, ts[-1]), file=sys.stderr, flush=True)
    timed = sorted(ts[1:])
    best, med = timed[0], timed[len(timed) // 2]
    return {"value": round(B / best, 4), "median": round(B / med, 4), "unit": "images/s", "cores": cores, "cpu": model_name, "kind": "port",
            "sample": "oracle (CPU restatement of mmvit4.MMVit4, bit-identical to the reference on CPU) fwd+loss+bwd, train mode "
                      "(dropout on), B=%d D=%d %dx%d fp32 (BASELINE configs[0]), 1 warm-up + %d timed steps: best %.1f s, median %.1f s "
                      "per step" % (B, D, HW, HW, iters, best, med)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (BASELINE config 2: 32)")
    ap.add_argument("--bands", type=int, default=4)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--dump-shapes", default=None, help="write per-shape MFMA launch timings (JSON lines) to this file")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("CORRIF_DIST_BACKEND", "nccl")     # "gloo": control-flow rehearsal of N ranks on a box with fewer GPUs
        local = local % torch.cuda.device_count() if backend != "nccl" else local
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        if os.environ.get("CORRIF_FORCE_COLLECTIVE") == "1":      # rehearsal: the N > 1 gradient path (bucket gather + RCCL all-reduce) with one rank
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(helpers_free_port()))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", local if world > 1 else 0)

    import mmvit4
    import ops
    import corrif_hip
    import helpers
    from data_parallel import GradAllReducer, broadcast_module_state
    ops.lib()                                           # fail loudly if the HIP library is missing
    timer = None if args.no_kernel_timing else MfmaTimer(ops)
    hbm = None if args.no_kernel_timing else HbmTimer(corrif_hip)

    torch.manual_seed(0)
    model = mmvit4.MMVit4().to(dev).train()
    if os.environ.get("CORRIF_GRAD_TAP") is not None:        # A/B switch
        mmvit4.GRAD_TAP = os.environ["CORRIF_GRAD_TAP"] == "1"
    if os.environ.get("CORRIF_DECODER_SPLIT") is not None:   # A/B switch
        model.decoder_split = int(os.environ["CORRIF_DECODER_SPLIT"])
    if os.environ.get("CORRIF_STREAM_K") is not None:        # A/B switch: 0 = every GEMM as one workgroup per tile
        ops.STREAM_K = os.environ["CORRIF_STREAM_K"] == "1"
    if os.environ.get("CORRIF_STREAM_K_LONG") is not None:   # A/B switch: 0 = no stream-K split for the long-K, few-tile GEMMs either
        ops.STREAM_K_LONG = os.environ["CORRIF_STREAM_K_LONG"] == "1"
    if os.environ.get("CORRIF_INTERLEAVE") is not None:      # A/B switch: 0 = the host enqueues one modality branch after the other
        model.interleave_branches = os.environ["CORRIF_INTERLEAVE"] == "1"
    if os.environ.get("CORRIF_BWD_STATS") is not None:       # A/B switch: 0 = every BatchNorm backward runs its own reduction pass
        ops.BWD_STATS = os.environ["CORRIF_BWD_STATS"] == "1"
    if os.environ.get("CORRIF_SIDE_WGRAD") is not None:      # A/B switch: 0 = the encoders' weight gradients stay on their branch stream
        ops.SIDE_WGRAD = os.environ["CORRIF_SIDE_WGRAD"] == "1"
    if os.environ.get("CORRIF_AUTO_STREAMS") is not None:    # A/B switch: 0 = keep the multi-stream schedule whatever the memory estimate says
        model.auto_streams = os.environ["CORRIF_AUTO_STREAMS"] == "1"
    if os.environ.get("CORRIF_FLASH") is not None:           # A/B switch: 0 = materialised attention scores
        ops.FLASH_ATTENTION = os.environ["CORRIF_FLASH"] == "1"
    if os.environ.get("CORRIF_SPLIT_BF16") is not None:      # A/B switch: 0 = the fp32-input MFMA main loop in every GEMM (rounds 1-3)
        ops.SPLIT_BF16 = os.environ["CORRIF_SPLIT_BF16"] == "1"
    if os.environ.get("CORRIF_TRILINEAR_SEP") is not None:   # A/B switch: 0 = the up-samplings' adjoint as the one-pass gather
        ops.TRILINEAR_SEPARABLE = os.environ["CORRIF_TRILINEAR_SEP"] == "1"
    if os.environ.get("CORRIF_GROUPED") is not None:         # A/B switch: 0 = one Encoder.forward per modality (three launches per twin layer)
        model.grouped_encoders = os.environ["CORRIF_GROUPED"] == "1"
    if os.environ.get("CORRIF_SIDE_WGRAD_G") is not None:    # A/B switch: 0 = the grouped encoder's weight gradients stay on the main stream
        ops.SIDE_WGRAD_GROUPED = os.environ["CORRIF_SIDE_WGRAD_G"] == "1"
    if os.environ.get("CORRIF_STREAM_K_G") is not None:      # A/B switch: 0 = no stream-K split for the grouped long-K launches
        ops.STREAM_K_GROUPED = os.environ["CORRIF_STREAM_K_G"] == "1"
    if os.environ.get("CORRIF_SERIAL") == "1":          # profiling aid: one stream, clean per-kernel attribution
        ops.SIDE_WGRAD = False
        ops.SIDE_WGRAD_GROUPED = False
        model.concurrent_branches = False
        model.decoder_fuse.concurrent_skips = False
        model.decoder_split = 0
    broadcast_module_state(model)
    reducer = GradAllReducer(model, force_collective=os.environ.get("CORRIF_FORCE_COLLECTIVE") == "1")
    B = args.batch
    xg, maskg = helpers.make_inputs(B, args.bands, args.size, args.size, seed=1234 + rank)   # rank r's own shard
    x, mask = xg.to(dev), maskg.to(dev)
    ops.manual_seed(1234 + rank)

    def step():
        reducer.zero_grad()
        pred = model(x)
        loss = ops.bce_with_logits_mean(pred, mask)     # F4_TRAIN.py:58-60
        loss.backward()
        reducer.finish()
        return loss.detach()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    warm_peak = torch.cuda.max_memory_allocated()
    torch.cuda.reset_peak_memory_stats()             # peak_mem_GB below = the timed steps' own peak (the bench's statistic, not the model's)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    peak_mem = torch.cuda.max_memory_allocated()     # of the timed steps
    kt_steps = 0
    if timer:
        # EVERY rank runs this pass (its steps contain the gradient all-reduces: a rank-0-only pass would leave the collectives
        # unmatched and hang the job); only rank 0 reports.
        # per-launch durations of the MFMA kernels: HIP events around every launch, in an extra pass right after the timed
        # region with the three modality branches serialised on one stream (with concurrent streams an event pair also
        # spans the other streams' kernels, so per-kernel durations are only meaningful one stream at a time)
        grouped_was, model.grouped_encoders = model.grouped_encoders, model.encoders_grouped_for(x)     # the timed region's encoder schedule
        model.concurrent_branches = False
        side_was, ops.SIDE_WGRAD = ops.SIDE_WGRAD, False
        sideg_was, ops.SIDE_WGRAD_GROUPED = ops.SIDE_WGRAD_GROUPED, False
        split_was, model.decoder_split = model.decoder_split, 0
        skips_was, model.decoder_fuse.concurrent_skips = model.decoder_fuse.concurrent_skips, False
        kt_steps = min(2, args.steps)
        # ~100 ms of queued GEMMs ahead of every instrumented step keep the host ahead of the GPU, so an event pair never spans
        # a moment where the queue ran dry (it would then measure host launch latency, not the kernel)
        ga, gb, gc = (torch.empty(4096, 4096, device=dev).normal_() for _ in range(3))
        gg = corrif_hip.gemm_geom()
        timer.on = False
        for _ in range(kt_steps):
            for _ in range(80):
                timer._gemm(ga.data_ptr(), 4096, gb.data_ptr(), 4096, 1, gc.data_ptr(), 4096, 4096, 4096, 4096, 4096, gg)
            timer.on = hbm.on = ops.TRACK_SPLIT = True
            step()
            timer.on = hbm.on = ops.TRACK_SPLIT = False
        torch.cuda.synchronize()
        del ga, gb, gc
        model.concurrent_branches = True
        model.grouped_encoders = grouped_was
        ops.SIDE_WGRAD = side_was
        ops.SIDE_WGRAD_GROUPED = sideg_was
        model.decoder_split = split_was
        model.decoder_fuse.concurrent_skips = skips_was
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    if rank == 0:
        std = B == 32 and args.bands == 4 and args.size == 224        # BASELINE configs[1] (= the per-GPU shard of configs[3])
        label = "(BASELINE configs[1])" if std else "(NOT the BASELINE headline workload: --batch/--bands/--size override)"
        out = {"metric": "images/sec fwd+bwd, %d-band %dx%d bs%d" % (args.bands, args.size, args.size, B), "value": round(value, 3),
               "unit": "images/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "mmvit4 CorrIFNet fwd+loss+bwd, %d bands/modality, %dx%d, batch %d per GPU, train mode %s"
                                      % (args.bands, args.size, args.size, B, label),
                          "global_batch": world * B, "parallelism": "dp%d" % world + (" (one rank, collectives forced)" if reducer.force else ""),
                          "loss": float(loss.item()),
                          "peak_mem_GB": round(peak_mem / 1e9, 2), "peak_mem_warmup_GB": round(warm_peak / 1e9, 2),
                          "encoder_schedule": "grouped (one launch per twin layer of the three modality encoders)"
                                              if model.encoders_grouped_for(x) else "per modality on three streams"}}
        # HBM bytes of the MFMA kernel family for ONE step of this workload, from the newest committed rocprofv3 PMC passes
        # (profiles/r??_mfma_traffic.json).  Quoted only while the kernel sources still are the ones the counters were taken with
        # (sha1 of csrc/ stored by tools/reduce_traffic.py); otherwise null - a stale number is worse than none.
        traffic, traffic_src = None, None
        import glob
        import hashlib
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_mfma_traffic.json")))
        if cands and B == 32 and args.bands == 4 and args.size == 224:
            try:
                tj = json.load(open(cands[-1]))
                h = hashlib.sha1()
                for f in sorted(glob.glob(os.path.join(PKG, "csrc", "*"))):
                    h.update(open(f, "rb").read())
                traffic_src = os.path.basename(cands[-1])
                if tj.get("csrc_sha1") == h.hexdigest():
                    traffic = float(tj["mfma_family_hbm_bytes_per_step"])
                else:
                    traffic_src += " (stale: csrc/ changed since the counters were collected - not quoted)"
            except Exception:
                traffic = None
        if timer:
            ovh = MfmaTimer.pair_overhead_ms()
            fl, ms, n = timer.summary(ovh)
            per_step_ms = ms / max(kt_steps, 1)
            # achieved = the flops the launches of the family actually carry (sum of 2*M*N*K over the launch descriptors) / the time the
            # family needs for one step: the kernels' own hardware utilisation, the definition of round 1 (the round-2 line
            # divided the reference's DENSE flops by that time and so credited work the compact skip branch no longer issues).  The
            # algorithmic rate (645.9 GFLOP per image forward + backward) is reported beside it under its own names,
            # `algorithmic_achieved` / `algorithmic_frac` (against the family's time) and `whole_step_frac` (against the whole timed step).
            algo = FLOP_PER_IMAGE_FWD_BWD * B if std else None
            launched = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            ach = launched
            fs, ms_s, ff, ms_f = timer.executed(ovh)
            # fraction of the INSTRUCTION peaks: the time the issued MFMA work would need at its own dense peak (6 bf16 products per fp32
            # product of a split launch at 2500 TFLOP/s; fp32-input MFMA at 157.3) over the time the family took
            t_roof_ms = (6.0 * fs / (PEAK_BF16_MFMA_TFLOPS * 1e12) + ff / (PEAK_FP32_MFMA_TFLOPS * 1e12)) * 1e3
            algo_ach = (algo / (per_step_ms * 1e-3) / 1e12) if (algo and per_step_ms > 0) else None
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                               "algorithmic_achieved": round(algo_ach, 3) if algo_ach else None,
                               "algorithmic_frac": round(algo_ach / PEAK_FP32_MFMA_TFLOPS, 4) if algo_ach else None,
                               "traffic_note": "HBM bytes per step of the same launches (sum over the MFMA family), rocprofv3 FETCH_SIZE x2 + "
                                               "WRITE_SIZE from separate --pmc passes (profiles/%s); algorithmic = 243 GB" % traffic_src,
                               "kernel": "gemm_fwd_kernel+wgrad_split_kernel (split-bf16 main loops) + conv3_patch_kernel+stem_*+flash_* (fp32-input MFMA)",
                               "issued": {"note": "gemm_fwd / wgrad launches on 128-row tiles form every fp32 product from six bf16 MFMA products of exactly "
                                                  "split operands (fp32-grade result, DESIGN section 4); achieved / peak above stay in fp32-equivalent "
                                                  "flops against the fp32-input MFMA peak; this block prices the work as issued",
                                          "split_bf16x6": {"fp32_equiv_tflops": round(fs / (ms_s * 1e-3) / 1e12, 1) if ms_s > 0 else None,
                                                           "bf16_tflops_issued": round(6 * fs / (ms_s * 1e-3) / 1e12, 1) if ms_s > 0 else None,
                                                           "peak_bf16": PEAK_BF16_MFMA_TFLOPS, "ms_per_step": round(ms_s / max(kt_steps, 1), 2),
                                                           "frac_of_bf16_peak": round(6 * fs / (ms_s * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4) if ms_s > 0 else None},
                                          "fp32_input_mfma": {"tflops": round(ff / (ms_f * 1e-3) / 1e12, 1) if ms_f > 0 else None,
                                                              "ms_per_step": round(ms_f / max(kt_steps, 1), 2),
                                                              "frac_of_fp32_peak": round(ff / (ms_f * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if ms_f > 0 else None},
                                          "frac_of_instruction_peaks": round(t_roof_ms / ms, 4) if ms > 0 else None},
                               "launches_per_step": n // max(kt_steps, 1), "mfma_ms_per_step": round(per_step_ms, 3),
                               "algorithmic_gflop_per_step": round(algo / 1e9, 1) if algo else None,
                               "launched_gflop_per_step": round(fl / max(kt_steps, 1) / 1e9, 1),
                               "launched": {"gflop_per_step": round(fl / max(kt_steps, 1) / 1e9, 1), "tflops": round(launched, 3),
                                            "frac": round(launched / PEAK_FP32_MFMA_TFLOPS, 4)},
                               "method": "HIP events around each MFMA launch (weight gradients incl. their fixed-order slab reduce), %d extra "
                                         "single-stream steps after the timed region; the empty-event-pair time is subtracted per launch; "
                                         "achieved = launched flops of a step / that kernel time" % kt_steps,
                               "event_pair_overhead_us": round(ovh * 1e3, 2),
                               "families": timer.by_kind(ovh, max(kt_steps, 1)),
                               # the HBM-bound kernel families: algorithmic bytes of every launch / its HIP-event duration, against 8 TB/s
                               "hbm_families": hbm.by_family(ovh, max(kt_steps, 1)),
                               "whole_step_frac": round(algo / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if algo else None}
        if timer and args.dump_shapes:
            with open(args.dump_shapes, "w") as f:
                for r in timer.by_shape():
                    f.write(json.dumps(r) + "\n")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1 or dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
