"""Generate the MMVit2 golden fixtures from the UPSTREAM reference (development container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mmvit2.py

Imports /root/reference/mmmvit2.py (read-only tree; pure torch/numpy, no stand-ins needed; no bytecode written), loads the
deterministic state-dict of tests/helpers.make_state_dict into it and stores numeric inputs/outputs only - never reference
source or bytecode.  The decoder always works on 16^3..128^3 grids whatever the input size, so the cases keep B small.
"""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import mmmvit2 as ref  # noqa: E402  (the upstream module)

torch.set_num_threads(8)

CASES = [
    # name, B, D, H, W, conv_gain, weight seed     (train mode, dropout probability 0; the model has no BatchNorm, so eval == this)
    ("mm2_b2_d4_32", 2, 4, 32, 32, 1.0, 4),
    ("mm2_b1_d3_40x24", 1, 3, 40, 24, 1.0, 5),
]


def sample(t, n=64):
    f = t.detach().reshape(-1)
    n = min(n, f.numel())
    idx = (torch.arange(n, dtype=torch.int64) * (f.numel() - 1)) // max(n - 1, 1)
    return f[idx].double().numpy()


def run_case(name, B, D, H, W, gain, wseed, dtype):
    torch.manual_seed(0)
    model = ref.MMVit2()
    sd = helpers.make_state_dict(model.state_dict(), seed=wseed, conv_gain=gain)
    model.load_state_dict(sd)
    model = model.to(dtype).train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.eval()
    x, mask = helpers.make_inputs(B, D, H, W)
    x, mask = x.to(dtype), mask.to(dtype)
    taps = {}

    def grab(key):
        def hook(mod, inp, out):
            taps[key] = sample(out[4] if isinstance(out, tuple) else out, 256)
        return hook

    hs = [model.RGB_encoder.e1_c3.register_forward_hook(grab("RGB_e1_c3")),
          model.NIR_encoder.e2_c1.register_forward_hook(grab("NIR_e2_c1")),
          model.SWIR_encoder.register_forward_hook(grab("SWIR_x5")),
          model.RGB_encoder.conv.register_forward_hook(grab("RGB_x6")),
          model.NIR_transformer.register_forward_hook(grab("NIR_transformer")),
          model.multimodal_transformer.register_forward_hook(grab("mm_transformer")),
          model.multimodal_decode_conv.register_forward_hook(grab("x6_inter")),
          model.decoder_fuse.d4_c1.register_forward_hook(grab("d4_c1")),
          model.decoder_fuse.d1_out.register_forward_hook(grab("d1_out"))]
    t0 = time.time()
    pred = model(x)
    out = {"pred_sample": pred.detach()[:, :, 0, ::4, ::4].double().numpy(),
           "pred_sum": np.float64(pred.detach().double().sum().item()),
           "pred_sqsum": np.float64((pred.detach().double() ** 2).sum().item())}
    for k, v in taps.items():
        out["tap_" + k] = v
    loss = F.binary_cross_entropy_with_logits(pred, mask)      # the training loss of F4_TRAIN.py:58-60
    loss.backward()
    out["loss"] = np.float64(loss.item())
    for k, p in model.named_parameters():
        if k in helpers.GRAD_KEYS_MMVIT2:
            assert p.grad is not None, k
            out["grad_sample/" + k] = sample(p.grad)
            out["grad_norm/" + k] = np.float64(p.grad.double().norm().item())
    nog = [k for k, p in model.named_parameters() if p.grad is None]
    out["nograd_count"] = np.int64(len(nog))
    assert all(k.startswith(helpers.NOGRAD_PREFIXES_MMVIT2) for k in nog), nog
    for h in hs:
        h.remove()
    print("  %s %s: %.1fs" % (name, str(dtype).split(".")[-1], time.time() - t0), flush=True)
    return out


def main():
    meta = {"torch": torch.__version__, "cases": []}
    for (name, B, D, H, W, gain, wseed) in CASES:
        blob = {}
        for dtype in (torch.float32, torch.float64):
            r = run_case(name, B, D, H, W, gain, wseed, dtype)
            tag = "f32" if dtype == torch.float32 else "f64"
            for k, v in r.items():
                blob[tag + "/" + k] = v
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **blob)
        meta["cases"].append({"name": name, "B": B, "D": D, "H": H, "W": W, "mode": "train_nodrop", "conv_gain": gain, "wseed": wseed})
    torch.manual_seed(0)
    m = ref.MMVit2()
    inv = {k: [list(v.shape), str(v.dtype).split(".")[-1]] for k, v in m.state_dict().items()}
    meta["n_params"] = int(sum(p.numel() for p in m.parameters()))
    with open(os.path.join(HERE, "state_dict_inventory_mmvit2.json"), "w") as f:
        json.dump(inv, f)
    with open(os.path.join(HERE, "meta_mmvit2.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("n keys", len(inv), "params", meta["n_params"])


if __name__ == "__main__":
    main()
