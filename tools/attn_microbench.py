"""Attention core fwd + bwd at the step's shapes: flash kernels vs the materialised-score path (GPU box).
    python tools/attn_microbench.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402,F401
import torch  # noqa: E402

import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
heads, C = 8, 512
for N in (2048, 512):
    qkv = (torch.randn(B, N, 3 * C, device=dev) * 0.5)
    go = torch.randn(B, N, C, device=dev)
    flops = 4.0 * B * heads * N * N * 64          # q.k^T + p.v forward
    for flash in (True, False):
        ops.FLASH_ATTENTION = flash
        for p in (0.1, 0.0):
            def run():
                q = qkv.clone().requires_grad_()
                o = ops.attention(q, heads, p, True)
                o.backward(go)
            run()
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            e[0].record()
            for _ in range(5):
                run()
            e[1].record()
            torch.cuda.synchronize()
            ms = e[0].elapsed_time(e[1]) / 5
            mem = torch.cuda.max_memory_allocated() / 1e9
            print("N=%4d %-12s p=%.1f  fwd+bwd %7.2f ms  (%5.1f TFLOP/s on the 6 algorithmic products)  peak mem %.1f GB"
                  % (N, "flash" if flash else "materialised", p, ms, 3 * flops / ms / 1e9, mem), flush=True)
            torch.cuda.reset_peak_memory_stats()
