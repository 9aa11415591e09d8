"""How long does the HOST take to enqueue one MMVit4 step (forward + loss + backward) vs. how long the GPU takes to run it?
If the host time approaches the GPU time, parts of the step are launch-bound.  GPU box: python tools/cpu_enqueue.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit4, ops
dev = "cuda:0"
torch.manual_seed(0)
B = int(os.environ.get("B", "32"))
model = mmvit4.MMVit4().to(dev).train()
x, mask = helpers.make_inputs(B, 4, 224, 224); x, mask = x.to(dev), mask.to(dev)
for it in range(5):
    model.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pred = model(x)
    loss = ops.bce_with_logits_mean(pred, mask)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t1s = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("step %d: fwd enqueue %.1f ms (gpu done at %.1f) | bwd enqueue %.1f ms (gpu done at %.1f)" %
          (it, (t1 - t0) * 1e3, (t1s - t0) * 1e3, (t2 - t1s) * 1e3, (t3 - t1s) * 1e3), flush=True)
