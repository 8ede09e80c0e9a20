"""Shared inputs of the augmentation parity tests (CPU and GPU run the same torch program against the scipy oracle)."""
import copy

import numpy as np
from scipy import ndimage

import mlagg_unet_amd  # noqa: F401
from mlagg_unet_amd import augmentation as AUG

B, C = 6, 2
IN, OUT = (75, 83), (48, 56)                      # loader patch (ragged, odd) -> network patch


def images(seed=0, shape=IN):
    rng = np.random.RandomState(seed)
    data = (ndimage.gaussian_filter(rng.randn(B, C, *shape), (0, 0, 2, 2)) * 5).astype(np.float32)
    seg = (ndimage.gaussian_filter(rng.randn(B, 1, *shape), (0, 0, 4, 4)) * 40).round().clip(-1, 3).astype(np.float32)
    return data, seg


def forced_params(seed=1):
    """Every transform active on some sample / channel; rotation only, scale only, both, neither all present."""
    rng = np.random.RandomState(seed)
    p = AUG.draw_params(np.random.RandomState(seed), B, C)
    for k in p:
        if k.startswith("do_"):
            p[k][:] = True
    p["do_rot"][:] = [1, 0, 1, 0, 1, 0]
    p["do_scale"][:] = [0, 1, 1, 0, 1, 1]
    p["angle"][:] = rng.uniform(-3.1, 3.1, B)
    p["scale"][:] = rng.uniform(0.7, 1.4, B)
    p["noise_std"][:] = rng.uniform(0, 0.1, B)
    p["blur_ch"][:] = rng.rand(B, C) < 0.6
    p["blur_sigma"][:] = rng.uniform(0.5, 1, (B, C))
    p["bright"][:] = rng.uniform(0.75, 1.25, (B, C))
    p["contrast"][:] = rng.uniform(0.75, 1.25, (B, C))
    p["lowres_ch"][:] = rng.rand(B, C) < 0.6
    p["lowres_zoom"][:] = rng.uniform(0.5, 1, (B, C))
    p["gamma"][:] = rng.uniform(0.7, 1.5, (B, C))
    p["gamma_inv"][:] = rng.uniform(0.7, 1.5, (B, C))
    p["mirror"][:] = rng.rand(B, 2) < 0.5
    return p


def only(p, keys):
    q = copy.deepcopy(p)
    for k in q:
        if k.startswith("do_") and k not in keys:
            q[k][:] = False
    q["mirror"][:] = False
    return q


STAGES = (["do_rot", "do_scale"], ["do_noise"], ["do_blur"], ["do_bright"], ["do_contrast"], ["do_lowres"], ["do_gamma_inv"],
          ["do_gamma"])
