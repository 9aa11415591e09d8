"""MMVit2 fwd+loss+bwd timing at the headline input shape (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit2, ops
dev = "cuda:0"
B = int(os.environ.get("B", "16"))
torch.manual_seed(0)
model = mmvit2.MMVit2().to(dev).train()
if os.environ.get("SPLIT") is not None: model.decoder_split = int(os.environ["SPLIT"])
x, mask = helpers.make_inputs(B, 4, 224, 224); x, mask = x.to(dev), mask.to(dev)
def step():
    model.zero_grad(set_to_none=True)
    loss = ops.bce_with_logits_mean(model(x), mask); loss.backward(); return loss
for _ in range(2): l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print("MMVit2 B=%d D=4 224x224: %.1f ms/step, %.1f images/s, loss %.5f, peak mem %.1f GB" % (B, dt * 1e3, B / dt, l.item(), torch.cuda.max_memory_allocated() / 1e9), flush=True)
