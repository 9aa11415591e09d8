"""ORACLE - test infrastructure only.  numpy restatement of the reference loader's arithmetic (F8_IMAGES4.py:36-88):
band selection (9-11, 12-14 of the 20-band cube), HWC -> CHW, per-band training-mean subtraction, stacking to
[N,3,3,224,224] and the 3x repeated mask.  Pinned by tests/golden/prep.npz (the reference's own get_images4 run on
synthetic patches with its file I/O patched, tests/golden/make_golden_prep.py)."""
import numpy as np


def prepare_inputs(rgb, all20, masks, trind):
    """rgb [N,H,W,3], all20 [N,H,W,20], masks [N,H,W] float32; returns images [N,3,3,H,W], targets [N,3,1,H,W], means[9]"""
    rgb = np.asarray(rgb, dtype=np.float32)
    all20 = np.asarray(all20, dtype=np.float32)
    groups = [np.moveaxis(rgb, 3, 1).copy(),                       # F8_IMAGES4.py:49-50
              np.moveaxis(all20[..., 9:12], 3, 1).copy(),          # :36-38,44,51-52
              np.moveaxis(all20[..., 12:15], 3, 1).copy()]         # :40-42,45,53-54
    means = []
    for g in groups:                                               # :57-79 per-band mean over the training indices
        for c in range(3):
            m = g[trind, c, :, :].mean()
            g[:, c, :, :] = g[:, c, :, :] - m
            means.append(m)
    images = np.stack(groups, axis=1)                              # :86
    targets = np.repeat(np.asarray(masks, dtype=np.float32)[:, None, None, :, :], 3, axis=1)   # :55,87
    return images, targets, np.asarray(means, dtype=np.float32)
