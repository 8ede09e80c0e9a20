"""Sliding-window inference for the 2-D MLAgg-UNet on MI355X, the device side of
the reference's mlagg/nnunetv2/inference/sliding_window_prediction.py:118-210 and the network restore of
predict_from_raw_data.py:96-99.

Same tiling, Gaussian importance map and mirror test-time augmentation as the reference, but
  * tiles are BATCHED through the network (the reference runs one tile, and one flip, per forward: its
    fixed 256x256 patch leaves the GPU idle between ~30 us kernels),
  * the four mirror variants of a tile batch run as one 4x larger batch,
  * logits and weights accumulate in fp32 on the device (the reference accumulates in half),
  * a training checkpoint (which holds the deep-supervision heads out_1..out_4) loads into the inference
    network built with enable_deep_supervision=False (the reference's strict load fails: SURVEY finding 7d).
"""
import numpy as np
import torch


def compute_gaussian(tile_size, sigma_scale=1.0 / 8):
    """Reference compute_gaussian (:13-28): a unit impulse at the tile centre blurred with sigma = size / 8
    (scipy's truncated kernel), normalised to max 1, zeros replaced by the smallest non-zero value."""
    from scipy.ndimage import gaussian_filter
    tmp = np.zeros(tile_size)
    tmp[tuple(i // 2 for i in tile_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in tile_size], 0, mode="constant", cval=0)
    g = (g / g.max()).astype(np.float16)              # the reference builds it in half; keep its rounding
    g[g == 0] = g[g != 0].min()
    return torch.from_numpy(g.astype(np.float32))


def compute_steps_for_sliding_window(image_size, tile_size, tile_step_size):
    """Reference :31-57."""
    if not 0 < tile_step_size <= 1:
        raise RuntimeError("step_size must be larger than 0 and smaller or equal to 1")
    steps = []
    for size, tile in zip(image_size, tile_size):
        if size < tile:
            raise RuntimeError("image size must be as large or larger than patch_size")
        num = int(np.ceil((size - tile) / (tile * tile_step_size))) + 1
        actual = (size - tile) / (num - 1) if num > 1 else 0.0
        steps.append([int(np.round(actual * i)) for i in range(num)])
    return steps


def _pad_to_tile(image, tile_size):
    """acvl_utils.pad_nd_image semantics: symmetric zero padding of the trailing dims up to the tile size."""
    pads, slicer = [], []
    for s, t in zip(image.shape[-len(tile_size):], tile_size):
        diff = max(t - s, 0)
        pads.append((diff // 2, diff // 2 + diff % 2))
        slicer.append(slice(diff // 2, diff // 2 + s))
    flat = []
    for lo, hi in reversed(pads):
        flat += [lo, hi]
    return torch.nn.functional.pad(image, flat), tuple(slicer)


def load_inference_weights(network, state_dict):
    """Load a TRAINING checkpoint's `network_weights` into a network built without deep supervision."""
    own = network.state_dict()
    kept = {k: v for k, v in state_dict.items() if k in own}
    dropped = [k for k in state_dict if k not in own]
    if any(not k.startswith(("out_1.", "out_2.", "out_3.", "out_4.")) for k in dropped):
        raise RuntimeError(f"unexpected keys in checkpoint: {dropped[:5]}")
    network.load_state_dict(kept, strict=True)
    return dropped


@torch.no_grad()
def predict_sliding_window_return_logits(network, input_image, num_segmentation_heads, tile_size, mirror_axes=None,
                                         tile_step_size=0.5, use_gaussian=True, tile_batch=8, device=None):
    """input_image (c, D, X, Y) with a 2-D tile_size -> fp32 logits (num_segmentation_heads, D, X, Y) on `device`.
    `network(x)` must return a tensor (deep supervision disabled)."""
    if input_image.dim() != 4 or len(tile_size) != 2:
        raise RuntimeError("input_image must be (c, D, X, Y) and tile_size 2-D")
    device = torch.device(device) if device is not None else next(network.parameters()).device
    network.eval()
    data, revert = _pad_to_tile(torch.as_tensor(input_image, dtype=torch.float32), tuple(tile_size))
    data = data.to(device)
    D, X, Y = data.shape[1:]
    gaussian = compute_gaussian(tuple(tile_size)).to(device) if use_gaussian else torch.ones(tuple(tile_size), device=device)
    logits = torch.zeros((num_segmentation_heads, D, X, Y), dtype=torch.float32, device=device)
    weight = torch.zeros((D, X, Y), dtype=torch.float32, device=device)
    steps = compute_steps_for_sliding_window((X, Y), tile_size, tile_step_size)
    places = [(d, sx, sy) for d in range(D) for sx in steps[0] for sy in steps[1]]
    flips = [()]
    if mirror_axes is not None:
        if max(mirror_axes) > 1:
            raise RuntimeError("mirror_axes does not match the dimension of the input")
        if 0 in mirror_axes:
            flips.append((2,))
        if 1 in mirror_axes:
            flips.append((3,))
        if 0 in mirror_axes and 1 in mirror_axes:
            flips.append((2, 3))
    tx, ty = tile_size
    for i in range(0, len(places), tile_batch):
        chunk = places[i:i + tile_batch]
        tiles = torch.stack([data[:, d, sx:sx + tx, sy:sy + ty] for d, sx, sy in chunk])       # (n, c, tx, ty)
        batch = torch.cat([torch.flip(tiles, f) if f else tiles for f in flips])              # (n * nflip, ...)
        out = network(batch)
        if isinstance(out, (list, tuple)):
            raise RuntimeError("the inference network must be built with enable_deep_supervision=False")
        pred = out[:len(chunk)].clone()
        for j, f in enumerate(flips[1:], start=1):
            pred += torch.flip(out[j * len(chunk):(j + 1) * len(chunk)], f)
        pred = pred / len(flips) * gaussian
        for (d, sx, sy), p in zip(chunk, pred):
            logits[:, d, sx:sx + tx, sy:sy + ty] += p
            weight[d, sx:sx + tx, sy:sy + ty] += gaussian
    logits /= weight
    return logits[(slice(None), slice(None), *revert)]
