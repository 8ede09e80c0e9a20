"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the sliding-window inference row of SURVEY.md section 8(f)-2:
a restatement of mlagg/nnunetv2/inference/sliding_window_prediction.py (compute_gaussian :13-28,
compute_steps_for_sliding_window :31-57, get_sliding_window_generator :60-84, maybe_mirror_and_predict
:87-115, predict_sliding_window_return_logits :118-210) on its CPU branch: network in fp32 (no autocast),
accumulators and gaussian in HALF exactly as the reference allocates them.

`pad_nd_image` comes from acvl_utils (third-party, absent offline): restated from its published behaviour
(symmetric constant padding of the trailing dims up to the tile size, slicer to crop back) -- unpinned.
Pinned by tests/golden/sliding_window.npz, produced by the reference's own function (make_golden.py)."""
import numpy as np
import torch
from scipy.ndimage import gaussian_filter


def pad_nd_image(image, new_shape, value=0.0):
    shp = list(image.shape)
    nd = len(new_shape)
    target = [max(s, n) for s, n in zip(shp[-nd:], new_shape)]
    pads, slicer = [], [slice(None)] * (len(shp) - nd)
    for s, t in zip(shp[-nd:], target):
        diff = t - s
        lo, hi = diff // 2, diff // 2 + diff % 2
        pads.append((lo, hi))
        slicer.append(slice(lo, lo + s))
    flat = []
    for lo, hi in reversed(pads):
        flat += [lo, hi]
    return torch.nn.functional.pad(image, flat, mode="constant", value=value), tuple(slicer)


def compute_gaussian(tile_size, sigma_scale=1.0 / 8, dtype=np.float16):
    tmp = np.zeros(tile_size)
    tmp[tuple(i // 2 for i in tile_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in tile_size], 0, mode="constant", cval=0)
    g = (g / np.max(g)).astype(dtype)
    g[g == 0] = np.min(g[g != 0])
    return g


def compute_steps(image_size, tile_size, tile_step_size):
    target = [i * tile_step_size for i in tile_size]
    num = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, tile_size)]
    steps = []
    for dim in range(len(tile_size)):
        mx = image_size[dim] - tile_size[dim]
        actual = mx / (num[dim] - 1) if num[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num[dim])])
    return steps


def slicers_2d_tiles(image_shape, tile_size, tile_step_size):
    """image_shape (D, X, Y) with a 2-D tile: every slice d, every (sx, sy)."""
    steps = compute_steps(image_shape[1:], tile_size, tile_step_size)
    for d in range(image_shape[0]):
        for sx in steps[0]:
            for sy in steps[1]:
                yield (slice(None), d, slice(sx, sx + tile_size[0]), slice(sy, sy + tile_size[1]))


def mirror_predict(network, x, mirror_axes):
    pred = network(x)
    if mirror_axes is not None:
        n = 2 ** len(mirror_axes)
        if 0 in mirror_axes:
            pred = pred + torch.flip(network(torch.flip(x, (2,))), (2,))
        if 1 in mirror_axes:
            pred = pred + torch.flip(network(torch.flip(x, (3,))), (3,))
        if 0 in mirror_axes and 1 in mirror_axes:
            pred = pred + torch.flip(network(torch.flip(x, (2, 3))), (2, 3))
        pred = pred / n
    return pred


def predict_sliding_window(network, image, num_heads, tile_size, mirror_axes=None, tile_step_size=0.5,
                           use_gaussian=True, accum_dtype=torch.half):
    """image (c, D, X, Y) -> logits (num_heads, D, X, Y) in accum_dtype (half = the reference)."""
    network.eval()
    with torch.no_grad():
        data, revert = pad_nd_image(image, tile_size)
        gaussian = torch.from_numpy(compute_gaussian(tile_size)).half() if use_gaussian else None
        logits = torch.zeros((num_heads, *data.shape[1:]), dtype=accum_dtype)
        npred = torch.zeros(data.shape[1:], dtype=accum_dtype)
        if gaussian is not None:
            gaussian = gaussian.to(accum_dtype)
        for sl in slicers_2d_tiles(data.shape[1:], tile_size, tile_step_size):
            pred = mirror_predict(network, data[sl][None], mirror_axes)[0]
            logits[sl] += pred * gaussian if use_gaussian else pred      # fp32 product added into the accumulator
            npred[sl[1:]] += gaussian if use_gaussian else 1
        logits /= npred
    return logits[(slice(None), *revert[1:])]


def sliding_window_case():
    """The tiny deterministic network + images behind tests/golden/sliding_window.npz."""
    g = torch.Generator().manual_seed(23)
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 6, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(6, 3, 3, padding=1))
    for p_ in net.parameters():
        p_.data = torch.randn(p_.shape, generator=g) * 0.3
    img = torch.randn(2, 3, 40, 50, generator=g)
    small = torch.randn(2, 1, 20, 70, generator=g)
    return net, img, small
