"""Time the patch-staged 3x3x3 kernels (forward / data-gradient form and weight gradient) on the decoder shapes (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, ops, corrif_hip as H
from corrif_hip import lib
dev = "cuda:0"
REPS = int(os.environ.get("REPS", "5"))
def bench(fn, n=REPS):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
# (B, (D,H,W), Ci, Co, with_bias)
for (B, S, Ci, Co, wb) in [(32, (128, 128, 128), 32, 8, True), (32, (128, 128, 128), 16, 8, True), (32, (128, 128, 128), 8, 32, False), (32, (128, 128, 128), 8, 16, False),
                           (32, (64, 64, 64), 64, 16, True), (32, (64, 64, 64), 16, 32, False), (32, (4, 56, 56), 24, 24, True), (32, (4, 56, 56), 24, 24, False)][:int(os.environ.get("NSHAPES", "8"))]:
    cc = lib().corrif_conv3_patch_cc(Ci, Co)
    if not cc: continue
    M = B * S[0] * S[1] * S[2]
    x = torch.randn(M, Ci, device=dev); y = torch.empty(M, Co, device=dev)
    wp = torch.randn(Ci // cc, Co, 27, cc, device=dev); bias = torch.randn(Co, device=dev)
    ms = bench(lambda: ops.conv3_patch(x.data_ptr(), Ci, wp.data_ptr(), y.data_ptr(), Co, bias.data_ptr() if wb else 0, B, S, S, Ci, Co, 1, False, cc))
    fl = 2.0 * M * Co * 27 * Ci / 1e9
    line = "patch B%d %s Ci %2d Co %2d bias %d: %7.3f ms %6.1f TF/s" % (B, S, Ci, Co, wb, ms, fl / ms)
    if lib().corrif_conv3_patch_wgrad_slots(Ci, Co):
        gw = torch.empty(Co, 27, Ci, device=dev)
        ms2 = bench(lambda: ops.conv3_patch_wgrad(x.data_ptr(), Ci, y.data_ptr(), Co, gw.data_ptr(), B, S, S, Ci, Co, False, dev))
        line += " | wgrad %7.3f ms %6.1f TF/s" % (ms2, fl / ms2)
    print(line, flush=True)
    del x, y
