"""mmvit2 fixture case mm2_b2_d4_32: gradient-norm deviation from the fp64 fixture per tensor, split-bf16 loop on / off (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa
import numpy as np
import torch
import ops
import test_mmvit2_gpu as T

name = sys.argv[1] if len(sys.argv) > 1 else "mm2_b2_d4_32"
g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
for split in (True, False):
    ops.SPLIT_BF16 = split
    model, pred, mask, loss, _ = T.run_hip(T.CASES[name])
    params = dict(model.named_parameters())
    rows = []
    for k in helpers.GRAD_KEYS_MMVIT2:
        nr = float(g["f64/grad_norm/" + k]); n32 = abs(float(g["f32/grad_norm/" + k]) - nr)
        d = abs(params[k].grad.double().norm().item() - nr)
        rows.append((d / max(1e-2 * nr, 10 * n32), k, d / nr, n32 / nr))
    rows.sort(reverse=True)
    print("SPLIT_BF16 =", split, " loss %.7f (f64 %.7f, f32 %.7f)" % (loss.item(), float(g["f64/loss"]), float(g["f32/loss"])))
    for r in rows[:6]:
        print("   bar use %.2f  %-50s rel norm dev %.3e (fp32 reference's own %.3e)" % r)
