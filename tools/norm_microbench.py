"""Achieved HBM bandwidth of the normalisation kernels on the big decoder / encoder shapes (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, ops, corrif_hip as H
from corrif_hip import lib, P, check, stream
dev = "cuda:0"
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
# (rows_per_group, G, C): instance norm at the decoder levels (G = batch), batch norm in the encoders (G = 1)
for (rpg, G, C) in [(2097152, 32, 8), (2097152, 32, 16), (262144, 32, 16), (262144, 32, 32), (401408 * 4, 1, 64), (401408, 1, 256), (100352, 1, 512), (25088, 1, 1024)]:
    x = torch.randn(G * rpg, C, device=dev); y = torch.empty_like(x); gy = torch.randn_like(x); gx = torch.empty_like(x)
    mean = torch.zeros(G * C, device=dev); rstd = torch.ones(G * C, device=dev)
    ws = torch.empty(lib().corrif_norm_workspace(rpg, G, C) // 8 + 16, dtype=torch.float64, device=dev)
    n = x.numel() * 4 / 1e9
    t1 = bench(lambda: check(lib().corrif_norm_stats(P(x), C, rpg, G, C, 1, 1e-5, P(mean), P(rstd), None, None, 0.0, P(ws), stream()), "stats"))
    t2 = bench(lambda: check(lib().corrif_norm_apply(P(x), C, P(mean), P(rstd), None, None, None, 0, P(y), C, rpg, G, C, 1, stream()), "apply"))
    t3 = bench(lambda: check(lib().corrif_norm_bwd(P(gy), C, None, 0, P(x), C, P(mean), P(rstd), None, P(gx), C, None, 0, None, None, rpg, G, C, 1, 0, P(ws), stream()), "bwd"))
    print("rows/g %8d G %2d C %4d (%.2f GB): stats %.3f ms %.2f TB/s | apply %.3f ms %.2f TB/s | bwd(partial+apply) %.3f ms %.2f TB/s" %
          (rpg, G, C, n, t1, n / t1, t2, 2 * n / t2, t3, 5 * n / t3), flush=True)
    del x, y, gy, gx
