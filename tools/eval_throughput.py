"""Eval-mode (no-grad, BatchNorm running statistics) forward throughput of MMVit4: the TORCH_LIBRARY binding
(torch.ops.corrif.*, descriptors filled in C++) against the ctypes binding of the same kernels, and the captured HIP graph.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit4, ops, train
dev = "cuda:0"
torch.manual_seed(0)
model = mmvit4.MMVit4().to(dev).eval()


def timeit(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for B in (1, 2, 4, 8, 32):
    x, _ = helpers.make_inputs(B, 4, 224, 224); x = x.to(dev)
    res = {}
    with torch.no_grad():
        for label, on in (("torch.ops.corrif (C++ descriptors)", True), ("ctypes binding", False)):
            ops.USE_TORCH_LIBRARY = on
            res[label] = timeit(lambda: model(x), 20 if B <= 8 else 5)
        ops.USE_TORCH_LIBRARY = True
    if B <= 8:
        g = train.GraphedForward(model, x)
        res["HIP graph replay"] = timeit(lambda: g(x), 20)
    print("eval forward B=%2d (%s encoders): " % (B, "grouped" if model.encoders_grouped_for(x) else "per-modality") +
          "  |  ".join("%s %.2f ms = %.1f images/s" % (k, v * 1e3, B / v) for k, v in res.items()), flush=True)
