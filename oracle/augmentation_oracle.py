"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's training augmentation chain
(nnUNetTrainer.get_training_transforms, nnUNetTrainer.py:645-733) for the 2-D configuration.

Only tests/ may import this module; the product path (mlagg-unet_amd/augmentation.py) must not.

The transforms themselves live in batchgenerators (setup: batchgenerators>=0.25) and skimage, both third-party and ABSENT
offline: **parity unpinned**.  What is restated is their published arithmetic, sample by sample and channel by channel in plain
numpy / scipy.ndimage loops exactly as batchgenerators iterates (augment_spatial, augment_gaussian_blur,
augment_contrast, augment_linear_downsampling_scipy, augment_gamma, augment_mirroring), with the random draws replaced
by an explicit parameter dictionary (the layout of augmentation.draw_params) so that the device path can be compared
with identical parameters.  skimage.transform.resize(order, mode="edge", anti_aliasing=False, clip=True) is
scipy.ndimage.zoom(order, mode="nearest", grid_mode=True) followed by a clip to the input range.
"""
import numpy as np
from scipy import ndimage


def zero_centered_mesh(shape):
    """batchgenerators create_zero_centered_coordinate_mesh."""
    coords = np.array(np.meshgrid(*[np.arange(i) for i in shape], indexing="ij")).astype(float)
    for d in range(len(shape)):
        coords[d] -= (np.array(shape).astype(float) - 1)[d] / 2.0
    return coords


def rotate_2d(coords, angle):
    """rotate_coords_2d: coords^T . [[cos, -sin], [sin, cos]]."""
    R = np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]])
    return np.dot(coords.reshape(2, -1).T, R).T.reshape(coords.shape)


def interpolate_img(img, coords, order, cval, is_seg=False):
    """batchgenerators interpolate_img, border mode "constant"."""
    if is_seg and order != 0:
        result = np.zeros(coords.shape[1:], img.dtype)
        for c in np.unique(img):
            r = ndimage.map_coordinates((img == c).astype(float), coords, order=order, mode="constant", cval=cval)
            result[r >= 0.5] = c
        return result
    return ndimage.map_coordinates(img.astype(float), coords, order=order, mode="constant", cval=cval).astype(img.dtype)


def spatial(data, seg, patch_size, p):
    """augment_spatial with the nnU-Net arguments (B:666-677): rotation / isotropic scale, order 3 / 1, cval 0 / -1."""
    B = data.shape[0]
    out_d = np.zeros((B, data.shape[1]) + tuple(patch_size), dtype=np.float32)
    out_s = np.zeros((B, seg.shape[1]) + tuple(patch_size), dtype=np.float32)
    for b in range(B):
        coords = zero_centered_mesh(patch_size)
        modified = False
        if p["do_rot"][b]:
            coords, modified = rotate_2d(coords, p["angle"][b]), True
        if p["do_scale"][b]:
            coords, modified = coords * p["scale"][b], True
        if modified:
            for d in range(2):
                coords[d] += data.shape[d + 2] / 2.0 - 0.5
            for c in range(data.shape[1]):
                out_d[b, c] = interpolate_img(data[b, c], coords, 3, 0.0)
            for c in range(seg.shape[1]):
                out_s[b, c] = interpolate_img(seg[b, c], coords, 1, -1.0, is_seg=True)
        else:
            lb = [(data.shape[d + 2] - patch_size[d]) // 2 for d in range(2)]
            sl = (slice(lb[0], lb[0] + patch_size[0]), slice(lb[1], lb[1] + patch_size[1]))
            out_d[b], out_s[b] = data[b][(slice(None),) + sl], seg[b][(slice(None),) + sl]
    return out_d, out_s


def resize_edge(img, shape, order):
    """skimage.transform.resize(img, shape, order, mode="edge", anti_aliasing=False) (clip=True)."""
    zoom = [n / o for n, o in zip(shape, img.shape)]
    out = ndimage.zoom(img, zoom, order=order, mode="nearest", grid_mode=True)
    return np.clip(out, img.min(), img.max()) if order > 0 else out


def apply(data, seg, patch_size, p, noise):
    """The chain B:666-695 with the parameters `p` and the unit-variance noise field `noise` (B, C, H, W)."""
    data, seg = spatial(data.astype(np.float32), seg.astype(np.float32), patch_size, p)
    B, C = data.shape[:2]
    for b in range(B):
        if p["do_noise"][b]:
            data[b] += (noise[b] * p["noise_std"][b]).astype(np.float32)
    for b in range(B):
        if p["do_blur"][b]:
            for c in range(C):
                if p["blur_ch"][b, c]:
                    data[b, c] = ndimage.gaussian_filter(data[b, c], p["blur_sigma"][b, c], order=0)
    for b in range(B):
        if p["do_bright"][b]:
            for c in range(C):
                data[b, c] *= p["bright"][b, c]
    for b in range(B):
        if p["do_contrast"][b]:
            for c in range(C):
                mn, lo, hi = data[b, c].mean(), data[b, c].min(), data[b, c].max()
                data[b, c] = np.clip((data[b, c] - mn) * p["contrast"][b, c] + mn, lo, hi)
    for b in range(B):
        if p["do_lowres"][b]:
            shp = np.array(data.shape[2:])
            for c in range(C):
                if p["lowres_ch"][b, c]:
                    target = np.round(shp * p["lowres_zoom"][b, c]).astype(int)
                    down = resize_edge(data[b, c].astype(float), target, 0)
                    data[b, c] = resize_edge(down, shp, 3)
    for key_do, key_g, invert in (("do_gamma_inv", "gamma_inv", True), ("do_gamma", "gamma", False)):
        for b in range(B):
            if p[key_do][b]:
                x = -data[b] if invert else data[b].copy()
                for c in range(C):
                    mn, sd = x[c].mean(), x[c].std()
                    lo = x[c].min()
                    rnge = x[c].max() - lo
                    x[c] = np.power((x[c] - lo) / float(rnge + 1e-7), p[key_g][b, c]) * float(rnge + 1e-7) + lo
                    x[c] = x[c] - x[c].mean()
                    x[c] = x[c] / (x[c].std() + 1e-8) * sd
                    x[c] = x[c] + mn
                data[b] = -x if invert else x
    for b in range(B):
        if p["mirror"][b, 0]:
            data[b], seg[b] = data[b][:, ::-1].copy(), seg[b][:, ::-1].copy()
        if p["mirror"][b, 1]:
            data[b], seg[b] = data[b][:, :, ::-1].copy(), seg[b][:, :, ::-1].copy()
    return data, seg
