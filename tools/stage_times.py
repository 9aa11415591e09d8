"""Coarse wall-time split of one MMVit4 step on the GPU box (forward stages via events; backward = total - forward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit4, ops
dev = "cuda:0"
torch.manual_seed(0)
model = mmvit4.MMVit4().to(dev).train()
x, mask = helpers.make_inputs(32, 4, 224, 224); x, mask = x.to(dev), mask.to(dev)
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
dec = model.decoder_fuse
orig_dec = dec.forward
def dec_fwd(*a):
    mark("pre_decoder"); r = orig_dec(*a); mark("post_decoder"); return r
dec.forward = dec_fwd
mt = model.multimodal_transformer
orig_mt = mt.forward
def mt_fwd(*a):
    mark("pre_mm_transformer"); r = orig_mt(*a); mark("post_mm_transformer"); return r
mt.forward = mt_fwd
for it in range(3):
    marks.clear()
    model.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); mark("start")
    pred = model(x)
    loss = ops.bce_with_logits_mean(pred, mask); mark("fwd_done")
    loss.backward(); mark("bwd_done")
    torch.cuda.synchronize()
t0 = marks[0][1]
for n, e in marks[1:]:
    print("%-22s %8.1f ms" % (n, t0.elapsed_time(e)))
