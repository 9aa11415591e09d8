"""test_full_gradient_against_oracle's case over several weight seeds, split-bf16 loop on / off: is a tensor's excess systematic or one
rounding realisation?  (GPU box)   python tools/fullgrad_seeds.py 11 4 7"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa
import torch
import ops
import test_model_gpu as T
from oracle import mmvit4_oracle as O

KEYS = ["decoder_fuse.d1_out.conv.weight", "decoder_fuse.d1_out.conv.bias", "decoder_fuse.d1_c2.conv.weight", "fusion4.conv.bias"]
DEV = torch.device("cuda:0")
for wseed in [int(v) for v in sys.argv[1:]] or [11, 4, 7]:
    case = dict(B=2, D=3, H=64, W=64, mode="train_nodrop", conv_gain=1.0, wseed=wseed)
    x, mask = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
    sd = helpers.make_state_dict(O.MMVit4().state_dict(), seed=wseed, conv_gain=1.0)
    res = {}
    for dt in (torch.float32, torch.float64):          # the oracle's modules on the device (fp64 = truth, fp32 = the bracket), MIOpen off
        ref = O.MMVit4(); ref.load_state_dict(sd); ref = ref.to(device=DEV, dtype=dt).train(); O.set_dropout(ref, False)
        with torch.backends.cudnn.flags(enabled=False):
            O.train_step_loss(ref(x.to(device=DEV, dtype=dt)), mask.to(device=DEV, dtype=dt)).backward()
        res[dt] = {k: p.grad.double().cpu() for k, p in ref.named_parameters() if p.grad is not None}
        del ref
    for split in (True, False):
        ops.SPLIT_BF16 = split
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
        r = sorted(q[0] for q in rows)
        d = {q[3]: q for q in rows}
        print("seed %2d split %-5s median %.2f  n>4x %d  worst %.1f %-40s | " % (wseed, split, r[len(r) // 2], sum(1 for q in rows if q[1] > max(4 * q[2], 2e-4)), rows[0][0], rows[0][3][:40])
              + "  ".join("%s %.1f (%.1e/%.1e)" % (k.split(".")[-3] + "." + k.split(".")[-1], d[k][0], d[k][1], d[k][2]) for k in KEYS), flush=True)
        del model
