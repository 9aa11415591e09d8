"""CPU: the MMVit2 oracle restatement (oracle/mmvit2_oracle.py) against the fixtures captured from the
upstream reference's mmmvit2.py by tests/golden/make_golden_mmvit2.py."""
import json
import os

import numpy as np
import pytest
import torch

import helpers
from oracle import mmvit2_oracle as O2
from oracle import mmvit4_oracle as O

META = json.load(open(os.path.join(helpers.GOLDEN, "meta_mmvit2.json")))
CASES = {c["name"]: c for c in META["cases"]}


def sample(t, n=64):
    f = t.detach().reshape(-1)
    n = min(n, f.numel())
    idx = (torch.arange(n, dtype=torch.int64) * (f.numel() - 1)) // max(n - 1, 1)
    return f[idx].double().numpy()


def test_state_dict_inventory_matches_reference():
    inv = json.load(open(os.path.join(helpers.GOLDEN, "state_dict_inventory_mmvit2.json")))
    m = O2.MMVit2()
    sd = m.state_dict()
    assert list(sd.keys()) == list(inv.keys())            # same keys, same order
    for k, v in sd.items():
        assert list(v.shape) == inv[k][0], k
        assert str(v.dtype).split(".")[-1] == inv[k][1], k
    assert sum(p.numel() for p in m.parameters()) == META["n_params"]


@pytest.mark.parametrize("name", ["mm2_b1_d3_40x24"])
def test_oracle_matches_reference_fixture(name):
    case = CASES[name]
    g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
    torch.manual_seed(0)
    model = O2.MMVit2()
    model.load_state_dict(helpers.make_state_dict(model.state_dict(), seed=case["wseed"], conv_gain=case["conv_gain"]))
    model.train()
    O.set_dropout(model, False)
    x, mask = helpers.make_inputs(case["B"], case["D"], case["H"], case["W"])
    pred = model(x)
    ps = pred.detach()[:, :, 0, ::4, ::4].double().numpy()
    np.testing.assert_allclose(ps, g["f32/pred_sample"], rtol=0, atol=2e-7)     # identical ATen ops in identical order
    assert abs(pred.detach().double().sum().item() - float(g["f32/pred_sum"])) < 1e-3
    loss = O.train_step_loss(pred, mask)
    loss.backward()
    assert abs(loss.item() - float(g["f32/loss"])) < 1e-6
    params = dict(model.named_parameters())
    for k in helpers.GRAD_KEYS_MMVIT2:
        ref = g["f32/grad_sample/" + k]
        got = sample(params[k].grad)
        scale = max(np.abs(ref).max(), 1e-12)
        assert np.abs(got - ref).max() <= 1e-4 * scale + 1e-9, k
    nog = [k for k, p in params.items() if p.grad is None]
    assert len(nog) == int(g["f32/nograd_count"])
    assert all(k.startswith(helpers.NOGRAD_PREFIXES_MMVIT2) for k in nog)
