"""Peak memory of one training step under the product's switches (GPU box): python tools/mem_probe.py B D HW"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit4, ops
dev = "cuda:0"
B, D, HW = (int(v) for v in sys.argv[1:4])
x, mask = helpers.make_inputs(B, D, HW, HW); x, mask = x.to(dev), mask.to(dev)


def run(label, single=False, **sw):
    for k, v in sw.items():
        setattr(ops, k, v)
    torch.manual_seed(0)
    model = mmvit4.MMVit4().to(dev).train()
    model.auto_streams = False
    if single:
        model.concurrent_branches, model.decoder_split, model.decoder_fuse.concurrent_skips = False, 0, False
    if "grouped" in label:
        model.grouped_encoders = True
    elif "twins" in label:
        model.grouped_encoders = False
    peaks = []
    for step in range(3):
        for p in model.parameters(): p.grad = None
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
        pred = model(x)
        fwd = torch.cuda.max_memory_allocated()
        ops.bce_with_logits_mean(pred, mask).backward()
        torch.cuda.synchronize()
        peaks.append((fwd / 1e9, torch.cuda.max_memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
        del pred
    print("%-44s fwd-peak / step-peak / reserved GB per step: %s" % (label, ["%.1f/%.1f/%.1f" % p for p in peaks]), flush=True)
    for k in sw:
        setattr(ops, k, True)
    del model
    torch.cuda.empty_cache()


run("multi-stream twins")
run("single-stream twins", single=True)
run("single-stream grouped", single=True)
run("single-stream grouped, no side wgrad", single=True, SIDE_WGRAD_GROUPED=False)
run("single-stream twins, PATCH_STATS off", single=True, PATCH_STATS=False)
