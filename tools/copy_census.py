"""Which host-side operations issue the ~300 __amd_rocclr_copyBuffer (D2D hipMemcpyAsync) per training step?  (GPU box)
Profiles one forward + backward with CPU and device activities and lists, per CPU operator, the Memcpy / copyBuffer device events it
launched (correlated through the profiler's linked kernels)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit4, ops
dev = "cuda:0"
torch.manual_seed(0)
model = mmvit4.MMVit4().to(dev).train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x, mask = helpers.make_inputs(B, 4, 224, 224); x, mask = x.to(dev), mask.to(dev)
for _ in range(2):
    for p in model.parameters(): p.grad = None
    ops.bce_with_logits_mean(model(x), mask).backward()
torch.cuda.synchronize()
for p in model.parameters(): p.grad = None
A = torch.profiler.ProfilerActivity
with torch.profiler.profile(activities=[A.CPU, A.CUDA], record_shapes=True) as prof:
    ops.bce_with_logits_mean(model(x), mask).backward()
    torch.cuda.synchronize()
c = collections.Counter()
names = collections.Counter()
for e in prof.events():
    nm = e.name
    if "Memcpy" in nm or "copyBuffer" in nm or "memcpy" in nm:
        names[nm] += 1
    ks = getattr(e, "kernels", None) or []
    for k in ks:
        if "copyBuffer" in k.name or "Memcpy" in k.name or "memcpy" in k.name:
            c[(nm, str(e.input_shapes)[:80])] += 1
print("device copy events by name:", dict(names))
for (n, s), k in c.most_common(40): print("%4d  %-28s %s" % (k, n, s))
tot = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::detach", "aten::add_", "aten::add"):
        tot[e.name] += 1
print("aten ops:", dict(tot))
