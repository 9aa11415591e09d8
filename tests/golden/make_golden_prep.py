"""Golden fixture for the input pipeline: runs the UPSTREAM `F8_IMAGES4.get_images4` (development container only) on
synthetic patches by patching its file I/O (os.listdir / scipy.io.loadmat read from Windows paths that do not exist here).
Stores only numbers (means, checksums, samples).   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_prep.py"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402
sys.path.insert(0, "/root/reference")
import numpy as np  # noqa: E402
import scipy.io as sio  # noqa: E402

N = 6
rgb, all20, masks = helpers.make_raw_patches(N)
names = ["p%02d.mat" % i for i in range(N)]
real_listdir, real_loadmat = os.listdir, sio.loadmat


def fake_listdir(path):
    return list(names) if "DSTL" in path else real_listdir(path)


def fake_loadmat(path, **kw):
    i = names.index(os.path.basename(path))
    if "RGBs" in path:
        return {"inputPatch": rgb[i]}
    if "class06_mats" in path:
        return {"inputPatch": masks[i]}
    if "all20Ch" in path:
        return {"inputPatch": all20[i]}
    return real_loadmat(path, **kw)


os.listdir, sio.loadmat = fake_listdir, fake_loadmat
import F8_IMAGES4 as ref  # noqa: E402
trind = [0, 2, 3, 5]
images, targets, mR, mG, mB = ref.get_images4(N, 0, 0, [1], trind, [4], 0)
os.listdir, sio.loadmat = real_listdir, real_loadmat
images, targets = images.numpy(), targets.numpy()
assert images.shape == (N, 3, 3, 224, 224) and targets.shape == (N, 3, 1, 224, 224)
np.savez_compressed(os.path.join(HERE, "prep.npz"), trind=np.asarray(trind), mean_rgb=np.asarray([mR, mG, mB], dtype=np.float64),
                    img_sum=np.float64(images.astype(np.float64).sum()), img_sqsum=np.float64((images.astype(np.float64) ** 2).sum()),
                    img_sample=images[:, :, :, ::32, ::32].astype(np.float64), tgt_sum=np.float64(targets.sum()),
                    tgt_sample=targets[:, :, :, ::32, ::32].astype(np.float64),
                    band_means_after=images[trind].mean(axis=(0, 3, 4)).astype(np.float64))
print("prep golden written", images.shape, mR, mG, mB)
