"""Generate the golden fixtures under tests/golden/ from the UPSTREAM reference (development container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [case-name | train_trace ...]     (no argument: everything)

Imports /root/reference/mmvit4.py and F5_JACCARD2.py (read-only tree; no bytecode written),
with the topology-only torchvision stand-in of tests/golden/_tv_standin on sys.path ahead of it
(torchvision is not installed and resnet50() is used for topology only).
Only numeric inputs/outputs are stored - never reference source or bytecode.
"""
import json
import math
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "_tv_standin"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import mmvit4 as ref_mmvit4  # noqa: E402  (the upstream module)
import F5_JACCARD2 as ref_j2  # noqa: E402
import F5_JACCARD as ref_j1  # noqa: E402

torch.set_num_threads(8)

CASES = [
    # name, B, D, H, W, mode, conv_gain, weight seed
    ("tame_train_b2_d3_64", 2, 3, 64, 64, "train_nodrop", 1.0, 0),
    ("tame_eval_b3_d3_64", 3, 3, 64, 64, "eval", 1.0, 1),
    ("tame_train_b1_d4_224", 1, 4, 224, 224, "train_nodrop", 1.0, 2),
    ("kaiming_train_b2_d3_96", 2, 3, 96, 96, "train_nodrop", math.sqrt(2.0), 3),
    # round 2: the BASELINE sizes.  configs[0] = batch 4, 4 bands, 224^2 (the one configuration the reference itself runs,
    # F4_TRAIN.py:52-61); B = 3 at 224^2 (B | 3: the other branch of the inter-modal re-view); the reference's real init scale
    # (kaiming_normal_, mmvit4.py:437-439) at 224^2 with B = 2 (non-trivial re-view and BatchNorm batch statistics)
    # (fp32 only: the fp64 run of this case needs > 62 GB, more than the development container has; its fp64 truth is computed by
    # the GPU tests with the oracle's modules on the device, which tests/test_model_gpu.py pins against the fp64 fixtures below)
    ("tame_train_b4_d4_224", 4, 4, 224, 224, "train_nodrop", 1.0, 4),
    ("tame_eval_b3_d3_224", 3, 3, 224, 224, "eval", 1.0, 5),
    ("kaiming_train_b2_d4_224", 2, 4, 224, 224, "train_nodrop", math.sqrt(2.0), 6),
]


FP32_ONLY = {"tame_train_b4_d4_224"}


def sample(t, n=64):
    f = t.detach().reshape(-1)
    n = min(n, f.numel())
    idx = (torch.arange(n, dtype=torch.int64) * (f.numel() - 1)) // max(n - 1, 1)      # exact integer spacing
    return f[idx].double().numpy()


def run_case(name, B, D, H, W, mode, gain, wseed, dtype):
    torch.manual_seed(0)
    model = ref_mmvit4.MMVit4()
    sd = helpers.make_state_dict(model.state_dict(), seed=wseed, conv_gain=gain)
    model.load_state_dict(sd)
    model = model.to(dtype)
    if mode == "eval":
        model.eval()
    else:
        model.train()
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.eval()
    x, mask = helpers.make_inputs(B, D, H, W)
    x, mask = x.to(dtype), mask.to(dtype)
    taps = {}

    def grab(key):
        def hook(mod, inp, out):
            taps[key] = sample(out, 256)
        return hook

    hs = [model.RGB_encoder.e2.register_forward_hook(grab("RGB_e2")),
          model.SWIR_encoder.e5.register_forward_hook(grab("SWIR_e5")),
          model.fusion3.register_forward_hook(grab("fusion3")),
          model.NIR_transformer.register_forward_hook(grab("NIR_transformer")),
          model.qkv_RGB.register_forward_hook(grab("qkv_RGB")),
          model.multimodal_transformer.register_forward_hook(grab("mm_transformer")),
          model.multimodal_decode_conv.register_forward_hook(grab("x6_inter")),
          model.decoder_fuse.d4_c2.register_forward_hook(grab("d4_c2")),
          model.decoder_fuse.d1_out.register_forward_hook(grab("d1_out"))]
    mm_in = {}

    def pre(mod, inp):
        mm_in["x"] = sample(inp[0], 256)          # returns None: the input is left untouched

    hs.append(model.multimodal_transformer.register_forward_pre_hook(pre))
    t0 = time.time()
    with torch.set_grad_enabled(mode != "eval"):       # eval cases need no autograd state (B = 3 at 224^2 in fp64 would not fit otherwise)
        pred = model(x)
    out = {"pred_sample": pred.detach()[:, :, 0, ::4, ::4].double().numpy(),
           "pred_sum": np.float64(pred.detach().double().sum().item()),
           "pred_sqsum": np.float64((pred.detach().double() ** 2).sum().item()),
           "mm_in": mm_in["x"]}
    for k, v in taps.items():
        out["tap_" + k] = v
    n = B * 224 * 224
    out["jaccard2"] = ref_j2.Jaccard2(mask[:, 0].reshape(n, 1), pred.detach()[:, 0].reshape(n, 1)).double().numpy()
    if mode != "eval":
        loss = F.binary_cross_entropy_with_logits(pred, mask)      # F4_TRAIN.py:58-60
        loss.backward()
        out["loss"] = np.float64(loss.item())
        for k, p in model.named_parameters():
            if k in helpers.GRAD_KEYS:
                assert p.grad is not None, k
                out["grad_sample/" + k] = sample(p.grad)
                out["grad_norm/" + k] = np.float64(p.grad.double().norm().item())
        nog = [k for k, p in model.named_parameters() if p.grad is None]
        out["nograd_count"] = np.int64(len(nog))
        assert all(k.startswith(helpers.NOGRAD_PREFIXES) for k in nog), nog
        sd2 = model.state_dict()
        for k in ("RGB_encoder.e1_bn.running_mean", "RGB_encoder.e1_bn.running_var", "SWIR_encoder.e5.2.bn3.running_var",
                  "NIR_encoder.e3.0.downsample.1.running_mean"):
            out["buf/" + k] = sample(sd2[k])
    for h in hs:
        h.remove()
    print("  %s %s: %.1fs" % (name, str(dtype).split(".")[-1], time.time() - t0), flush=True)
    return out


def jaccard_cases():
    out = {}
    g = torch.Generator().manual_seed(7)
    n = 2 * 224 * 224
    y = (torch.rand(n, 1, generator=g) > 0.6).float()
    soft = torch.rand(n, 1, generator=g)
    hard = (soft > 0.5).float()
    zero = torch.zeros(n, 1)
    for nm, (a, b) in {"soft": (y, soft), "hard": (y, hard), "allzero_mask": (zero, soft), "allzero_hard": (zero, hard),
                       "perfect": (y, y.clone())}.items():
        out["j2_" + nm] = ref_j2.Jaccard2(a, b).numpy()
        out["j1_" + nm] = ref_j1.Jaccard(a, b).numpy()
        out["f1_" + nm] = ref_j2.JaccardAndF1(a, b).numpy()
    return out


def train_trace(dtype, B=2, D=3, HW=64, wseed=7, steps=3, lr=1e-4, step_size=1, gamma=0.5):
    """The training step of F4_TRAIN.py:41-71 with the optimiser of F2_MAIN.py:168-173 (torch.optim.Adam + StepLR, the scheduler
    stepped BEFORE the optimiser, F4_TRAIN.py:46), on the upstream model, dropout modules in eval (masks cannot be matched).
    step_size = 1 makes the scheduler quirk visible: the run uses lr * gamma from its first step on."""
    torch.manual_seed(0)
    model = ref_mmvit4.MMVit4()
    model.load_state_dict(helpers.make_state_dict(model.state_dict(), seed=wseed, conv_gain=1.0))
    model = model.to(dtype).train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.eval()
    optim = torch.optim.Adam(model.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.StepLR(optim, step_size=step_size, gamma=gamma)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sched.step()                                   # F4_TRAIN.py:46 - before any optimiser step
    x, mask = helpers.make_inputs(B, D, HW, HW)
    x, mask = x.to(dtype), mask.to(dtype)
    n = B * 224 * 224
    out = {"loss": [], "jaccard2": [], "lr": np.float64(optim.param_groups[0]["lr"])}
    for _ in range(steps):
        optim.zero_grad()
        pred = model(x)
        loss = nn.BCEWithLogitsLoss()(pred, mask)
        loss.backward()
        optim.step()
        out["loss"].append(loss.item())
        out["jaccard2"].append(ref_j2.Jaccard2(mask[:, 0].reshape(n, 1), pred.detach()[:, 0].reshape(n, 1)).double().item())
    out["loss"], out["jaccard2"] = np.array(out["loss"]), np.array(out["jaccard2"])
    sd = model.state_dict()
    tot = 0.0
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            tot += v.double().abs().sum().item()
    out["abs_sum_all"] = np.float64(tot)
    for k in helpers.GRAD_KEYS + ["RGB_encoder.e1_bn.running_mean", "SWIR_encoder.e5.2.bn3.running_var"]:
        out["param_sample/" + k] = sample(sd[k])
        out["param_norm/" + k] = np.float64(sd[k].double().norm().item())
    out["nbt"] = np.int64(int(sd["RGB_encoder.e1_bn.num_batches_tracked"]))
    frozen = helpers.make_state_dict(sd, seed=wseed, conv_gain=1.0)
    out["untouched"] = np.int64(sum(1 for k, v in model.named_parameters() if torch.equal(v.detach().float(), frozen[k])))
    return out


def main():
    only = set(sys.argv[1:])
    meta_path = os.path.join(HERE, "meta.json")
    meta = json.load(open(meta_path)) if only and os.path.exists(meta_path) else {"torch": torch.__version__, "cases": []}
    if not only or "train_trace" in only:
        blob = {}
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            for k, v in train_trace(dtype).items():
                blob[tag + "/" + k] = v
        np.savez_compressed(os.path.join(HERE, "train_trace_b2_d3_64.npz"), **blob)
        print("train trace", blob["f32/loss"], blob["f64/loss"], flush=True)
    for (name, B, D, H, W, mode, gain, wseed) in CASES:
        if only and name not in only:
            continue
        blob = {}
        for dtype in (torch.float32,) if name in FP32_ONLY else (torch.float32, torch.float64):
            r = run_case(name, B, D, H, W, mode, gain, wseed, dtype)
            tag = "f32" if dtype == torch.float32 else "f64"
            for k, v in r.items():
                blob[tag + "/" + k] = v
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **blob)
        meta["cases"] = [c for c in meta["cases"] if c["name"] != name]
        meta["cases"].append({"name": name, "B": B, "D": D, "H": H, "W": W, "mode": mode, "conv_gain": gain, "wseed": wseed,
                              "dtypes": ["f32"] if name in FP32_ONLY else ["f32", "f64"]})
        with open(meta_path, "w") as f:
            json.dump(meta, f, indent=1)
    if only:
        return
    np.savez_compressed(os.path.join(HERE, "jaccard.npz"), **jaccard_cases())
    # key / shape inventory of the reference state-dict (the drop-in contract)
    torch.manual_seed(0)
    m = ref_mmvit4.MMVit4()
    inv = {k: [list(v.shape), str(v.dtype).split(".")[-1]] for k, v in m.state_dict().items()}
    meta["n_params"] = int(sum(p.numel() for p in m.parameters()))
    with open(os.path.join(HERE, "state_dict_inventory.json"), "w") as f:
        json.dump(inv, f)
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("n keys", len(inv), "params", meta["n_params"])


if __name__ == "__main__":
    main()
