"""Which gradients does autograd COPY instead of adopting (the ~300 __amd_rocclr_copyBuffer per training step)?  (GPU box)
Profiles one backward with shapes and lists the aten::copy_ / aten::clone / aten::add_ calls by tensor shape."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit4, ops
dev = "cuda:0"
torch.manual_seed(0)
model = mmvit4.MMVit4().to(dev).train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x, mask = helpers.make_inputs(B, 4, 224, 224); x, mask = x.to(dev), mask.to(dev)
for _ in range(2):
    for p in model.parameters(): p.grad = None
    ops.bce_with_logits_mean(model(x), mask).backward()
torch.cuda.synchronize()
for p in model.parameters(): p.grad = None
loss = ops.bce_with_logits_mean(model(x), mask)
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True) as prof:
    loss.backward()
torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::add_", "aten::add", "aten::contiguous", "aten::zero_", "aten::fill_"):
        c[(e.name, str(e.input_shapes)[:90])] += 1
tot = collections.Counter()
for (n, s), k in c.items(): tot[n] += k
print(dict(tot))
for (n, s), k in c.most_common(60): print("%4d  %-16s %s" % (k, n, s))
shapes = collections.Counter(tuple(p.shape) for p in model.parameters() if p.grad is not None)
print("parameter shapes with gradients:", len(shapes), "distinct;", sum(shapes.values()), "tensors")
