"""Fixtures for the LARGE BASELINE geometries (configs[2]: 8 bands 256^2, configs[4]: 12 bands 512^2) at batch 2, forward + backward.

    python tests/golden/make_golden_large.py <out-dir> [case ...]

These sizes do not fit the development container (62 GB; the upstream model holds 85 GB of autograd state at 12 bands 512^2, B = 2, in
fp32), so they are produced on the GPU box's HOST cores by the CPU oracle (oracle/mmvit4_oracle.py) - which the committed reference
fixtures pin bit-exactly to the upstream arithmetic (tests/test_oracle_golden.py) - in fp32 (= the reference's arithmetic) and fp64
(truth), and committed like the reference fixtures.  tests/test_model_gpu.py::test_large_baseline_configs_fwd_bwd compares the HIP path
with them.  The address space is capped so that an over-estimate ends in a MemoryError of this process, not in a dead box.
"""
import os
import resource
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import mmvit4_oracle as O  # noqa: E402

CASES = helpers.LARGE_CASES      # name -> (B, D, H, W, weight seed)


def sample(t, n=64):
    f = t.detach().reshape(-1)
    n = min(n, f.numel())
    idx = (torch.arange(n, dtype=torch.int64) * (f.numel() - 1)) // max(n - 1, 1)
    return f[idx].double().numpy()


def run(name, dtype):
    B, D, H, W, wseed = CASES[name]
    torch.manual_seed(0)
    model = O.MMVit4()
    model.load_state_dict(helpers.make_state_dict(model.state_dict(), seed=wseed, conv_gain=1.0))
    model = model.to(dtype).train()
    O.set_dropout(model, False)                       # train-nodrop: batch statistics, dropout off (masks cannot be matched)
    x, mask = helpers.make_inputs(B, D, H, W)
    x, mask = x.to(dtype), mask.to(dtype)
    t0 = time.time()
    pred = model(x)
    loss = O.train_step_loss(pred, mask)
    print("  %s %s forward %.0f s" % (name, dtype, time.time() - t0), flush=True)
    loss.backward()
    print("  %s %s forward+backward %.0f s" % (name, dtype, time.time() - t0), flush=True)
    n = B * 224 * 224
    out = {"pred_sample": pred.detach()[:, :, 0, ::4, ::4].double().numpy(), "pred_sum": np.float64(pred.detach().double().sum().item()),
           "loss": np.float64(loss.item()),
           "jaccard2": helpers.jaccard2_ref(mask[:, 0].reshape(n, 1), pred.detach()[:, 0].reshape(n, 1)).double().numpy()}
    for k, p in model.named_parameters():
        if k in helpers.GRAD_KEYS:
            out["grad_sample/" + k] = sample(p.grad)
            out["grad_norm/" + k] = np.float64(p.grad.double().norm().item())
    sd = model.state_dict()
    for k in ("RGB_encoder.e1_bn.running_mean", "RGB_encoder.e1_bn.running_var", "SWIR_encoder.e5.2.bn3.running_var",
              "NIR_encoder.e3.0.downsample.1.running_mean"):
        out["buf/" + k] = sample(sd[k])
    return out


def main():
    outdir = sys.argv[1]
    names = sys.argv[2:] or list(CASES)
    os.makedirs(outdir, exist_ok=True)
    cap = int(os.environ.get("CORRIF_GOLDEN_MEM_GB", "230")) << 30
    resource.setrlimit(resource.RLIMIT_AS, (cap, cap))
    torch.set_num_threads(int(os.environ.get("CORRIF_CPU_THREADS", str(len(os.sched_getaffinity(0))))))
    for name in names:
        blob = {}
        want = os.environ.get("CORRIF_GOLDEN_DTYPES", "f32,f64").split(",")      # "f32": the reference's arithmetic only (fp64 truth from the device oracle)
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            if tag not in want:
                continue
            try:
                r = run(name, dtype)
            except (MemoryError, RuntimeError) as e:             # address-space cap reached: keep what exists, say so
                print("  %s %s FAILED: %s" % (name, tag, str(e)[:200]), flush=True)
                continue
            for k, v in r.items():
                blob[tag + "/" + k] = v
            np.savez_compressed(os.path.join(outdir, name + ".npz"), **blob)
        print("wrote", name, sorted(set(k.split("/")[0] for k in blob)), flush=True)


if __name__ == "__main__":
    main()
