"""test_full_gradient_against_oracle's case under kernel switches: which tensors leave the 4x bracket (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa
import torch
import ops
import test_model_gpu as T
from oracle import mmvit4_oracle as O

case = dict(B=2, D=3, H=64, W=64, mode="train_nodrop", conv_gain=1.0, wseed=11)
x, mask = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
sd = helpers.make_state_dict(O.MMVit4().state_dict(), seed=case["wseed"], conv_gain=case["conv_gain"])
res = {}
for dt in (torch.float32, torch.float64):
    ref = O.MMVit4(); ref.load_state_dict(sd); ref = ref.to(dt).train(); O.set_dropout(ref, False)
    pr = ref(x.to(dt)); O.train_step_loss(pr, mask.to(dt)).backward()
    res[dt] = {k: p.grad.double() for k, p in ref.named_parameters() if p.grad is not None}
import mmvit4
KEY = "decoder_fuse.d1_out.conv.weight"
variants = [("default", {}, None), ("SPLIT_BF16 off", {"SPLIT_BF16": False}, None),
            ("split, per-modality encoders", {}, False), ("split off, per-modality encoders", {"SPLIT_BF16": False}, False),
            ("split, grouped encoders", {}, True), ("split off, grouped encoders", {"SPLIT_BF16": False}, True),
            ("split, InstanceNorm statistics as their own pass", {"PATCH_STATS": False}, None),
            ("split off, InstanceNorm statistics as their own pass", {"SPLIT_BF16": False, "PATCH_STATS": False}, None),
            ("split, narrow 3x3x3 layers through the implicit GEMM", {"USE_PATCH": False}, None),
            ("split off, narrow 3x3x3 layers through the implicit GEMM", {"SPLIT_BF16": False, "USE_PATCH": False}, None)]
for name, sw, grouped in variants:
    for k, v in (("SPLIT_BF16", True), ("PATCH_STATS", True), ("USE_PATCH", True)):
        setattr(ops, k, sw.get(k, v))
    mmvit4.MMVit4.grouped_encoders = grouped
    model, pred, _, loss, _ = T.run_hip(case)
    rows = []
    for k, p in model.named_parameters():
        if p.grad is None or k not in res[torch.float64]:
            continue
        t = res[torch.float64][k]
        nrm = t.norm().clamp_min(1e-30)
        e = ((p.grad.double().cpu() - t).norm() / nrm).item()
        e32 = ((res[torch.float32][k] - t).norm() / nrm).item()
        rows.append((e / max(e32, 5e-5), e, e32, k))
    rows.sort(reverse=True)
    r = sorted(x[0] for x in rows)
    mine = [q for q in rows if q[3] == KEY][0]
    print("== %-58s median %.2f | %s: ratio %.1f (%.2e vs %.2e) | worst: %.1f %s, %.1f %s" % (name, r[len(r) // 2], KEY.split(".")[1], mine[0], mine[1], mine[2],
          rows[0][0], rows[0][3], rows[1][0], rows[1][3]), flush=True)
