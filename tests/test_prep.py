"""Input pipeline: the numpy oracle against the fixture captured from the reference's own get_images4, and (GPU) the
device kernels against the oracle."""
import os

import numpy as np
import pytest
import torch

import helpers

G = np.load(os.path.join(helpers.GOLDEN, "prep.npz"))
N = 6


def _oracle():
    from oracle import prep_oracle as P
    rgb, all20, masks = helpers.make_raw_patches(N)
    return (rgb, all20, masks), P.prepare_inputs(rgb, all20, masks, list(G["trind"]))


def test_prep_oracle_matches_reference_fixture():
    _, (images, targets, means) = _oracle()
    assert images.shape == (N, 3, 3, 224, 224) and targets.shape == (N, 3, 1, 224, 224)
    np.testing.assert_array_equal(images[:, :, :, ::32, ::32].astype(np.float64), G["img_sample"])     # same numpy ops: bit-identical
    np.testing.assert_array_equal(targets[:, :, :, ::32, ::32].astype(np.float64), G["tgt_sample"])
    np.testing.assert_allclose(means[:3].astype(np.float64), G["mean_rgb"], rtol=0, atol=0)
    assert abs(images.astype(np.float64).sum() - float(G["img_sum"])) < 1e-6 * abs(float(G["img_sqsum"])) ** 0.5 + 1e-3
    assert float(targets.sum()) == float(G["tgt_sum"])


@pytest.mark.gpu
def test_prep_kernels_match_oracle():
    import prep
    (rgb, all20, masks), (images, targets, means) = _oracle()
    dev = "cuda:0"
    im, tg, mu = prep.prepare_inputs(torch.from_numpy(rgb).to(dev), torch.from_numpy(all20).to(dev), torch.from_numpy(masks).to(dev),
                                     list(G["trind"]))
    torch.cuda.synchronize()
    # means: numpy accumulates in float32 pairwise, the kernel in double -> 1e-6 relative
    np.testing.assert_allclose(mu.cpu().numpy(), means, rtol=2e-6)
    # band values are O(100-1700): absolute tolerance follows the mean's rounding (one float32 ulp of ~1e3 is 6e-5)
    assert np.abs(im.cpu().numpy() - images).max() < 2e-3
    assert torch.equal(tg.cpu(), torch.from_numpy(targets))                       # mask copy: bit-exact
    np.testing.assert_allclose(im.cpu().numpy()[:, :, :, ::32, ::32], G["img_sample"], atol=2e-3, rtol=0)
    # with the oracle's means supplied the stacking itself is bit-exact (pure byte shuffling + one subtraction)
    im2, _, _ = prep.prepare_inputs(torch.from_numpy(rgb).to(dev), torch.from_numpy(all20).to(dev), None, None,
                                    means=torch.from_numpy(means).to(dev))
    assert torch.equal(im2.cpu(), torch.from_numpy(images))
