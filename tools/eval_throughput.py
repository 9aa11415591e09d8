"""Eval-mode (no-grad, BatchNorm running statistics) forward throughput of MMVit4.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit4
dev = "cuda:0"
torch.manual_seed(0)
model = mmvit4.MMVit4().to(dev).eval()
for B in (1, 8, 32):
    x, _ = helpers.make_inputs(B, 4, 224, 224); x = x.to(dev)
    with torch.no_grad():
        for _ in range(2): model(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 5
        for _ in range(n): model(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("eval forward B=%2d: %.1f ms, %.1f images/s, peak %.1f GB" % (B, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 1e9), flush=True)
