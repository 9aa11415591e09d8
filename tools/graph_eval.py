"""Batch-1 eval forward: eager vs HIP-graph replay (train.GraphedForward).  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit4, train
dev = "cuda:0"
torch.manual_seed(0)
model = mmvit4.MMVit4().to(dev).eval()
for B in (1, 4):
    x, _ = helpers.make_inputs(B, 4, 224, 224); x = x.to(dev)
    with torch.no_grad():
        for _ in range(3): ref = model(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ref = model(x)
        torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 10
    g = train.GraphedForward(model, x)
    out = g(x); torch.cuda.synchronize()
    same = torch.equal(out, ref)
    x2 = torch.randn_like(x)
    with torch.no_grad(): ref2 = model(x2).clone()
    same = same and torch.equal(g(x2), ref2) and torch.equal(g(x), ref)
    t0 = time.perf_counter()
    for _ in range(10): out = g(x)
    torch.cuda.synchronize(); graphed = (time.perf_counter() - t0) / 10
    print("B=%d eval forward: eager %.2f ms, HIP graph %.2f ms (%.2fx), bit-identical %s" % (B, eager * 1e3, graphed * 1e3, eager / graphed, same), flush=True)
