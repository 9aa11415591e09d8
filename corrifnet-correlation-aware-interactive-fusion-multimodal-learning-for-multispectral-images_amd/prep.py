"""Input pipeline of the reference's loader (F8_IMAGES4.py:36-88) on the GPU - SURVEY section 8(f) row N3.

    images, targets, means = prepare_inputs(rgb, all20, masks, trind)

rgb [N,224,224,3], all20 [N,224,224,20] (pixel-interleaved patches as read from the .mat files), masks [N,224,224]; `trind` = indices of
the training samples whose per-band means are removed.  Returns images [N,3,3,224,224], targets [N,3,1,224,224], means [9]
(R,G,B, N1..N3 = bands 9-11, S1..S3 = bands 12-14) - the same tensors `get_images4` hands to the DataLoader.
"""
import torch

from corrif_hip import check, lib, stream, ws_bytes


def prepare_inputs(rgb, all20, masks, trind, means=None):
    assert rgb.is_cuda and rgb.dtype == torch.float32 and rgb.shape[-1] == 3 and all20.shape[-1] == 20
    rgb, all20 = rgb.contiguous(), all20.contiguous()
    N, Hh, W = rgb.shape[:3]
    HW = Hh * W
    dev = rgb.device
    if means is None:
        tr = torch.as_tensor(trind, dtype=torch.int32, device=dev).contiguous()
        means = torch.empty(9, dtype=torch.float32, device=dev)
        ws = ws_bytes(lib().corrif_prep_workspace(tr.numel(), HW), dev)
        check(lib().corrif_prep_means(rgb.data_ptr(), all20.data_ptr(), tr.data_ptr(), tr.numel(), HW, means.data_ptr(), ws.data_ptr(), stream()),
              "corrif_prep_means")
    images = torch.empty((N, 3, 3, Hh, W), dtype=torch.float32, device=dev)
    targets = None
    mptr = 0
    if masks is not None:
        masks = masks.contiguous()
        targets = torch.empty((N, 3, 1, Hh, W), dtype=torch.float32, device=dev)
        mptr = masks.data_ptr()
    check(lib().corrif_prep_stack(rgb.data_ptr(), all20.data_ptr(), mptr, means.data_ptr(), images.data_ptr(),
                                  targets.data_ptr() if targets is not None else 0, N, HW, stream()), "corrif_prep_stack")
    return images, targets, means
