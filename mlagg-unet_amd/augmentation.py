"""Training-time data augmentation of the 2-D path ON the MI355X (second half).

The reference assembles its augmentation from batchgenerators transforms (nnUNetTrainer.get_training_transforms,
nnUNetTrainer.py:677-733: SpatialTransform, GaussianNoise, GaussianBlur, BrightnessMultiplicative, ContrastAugmentation,
SimulateLowResolution, Gamma x2, Mirror, then RemoveLabel / DownsampleSegForDS) and runs them in CPU worker processes on
numpy batches.  batchgenerators is third-party and absent offline, so its arithmetic is RESTATED here from its published
behaviour (v0.25) -- **parity unpinned**, like every third-party boundary of this repo; the restatement is held to a
scipy / numpy oracle (oracle/augmentation_oracle.py) transform by transform, with identical drawn parameters.

Shape of the design: parameters are drawn on the host (`draw_params`: a numpy RandomState, in the order batchgenerators
draws them, one sample after the other), the pixels never leave the device: every transform is a batched torch program over
(B, C, H, W) with per-sample parameter tensors and apply-masks -- no Python loop over samples, no host copy of an image.
  * SpatialTransform (rotation + isotropic scaling, no elastic deformation; data: cubic B-spline, constant 0 outside; seg:
    linear per label, result >= 0.5 wins in ascending label order, nothing assigned outside): the spline prefilter is two
    dense matmuls with the inverse collocation matrix (mirror boundary, as scipy's spline_filter for mode "constant"), the
    evaluation 16 gathers.  Untouched samples are centre-cropped.
  * GaussianBlur: separable grouped convolution, scipy's "reflect" boundary, per-(sample, channel) sigma.
  * SimulateLowResolution: nearest down-sampling + cubic B-spline up-sampling on the half-pixel-centred grid of skimage.
  * Gamma (plain and inverted, retain_stats), Contrast (range-preserving), Brightness, Mirror, Gaussian noise: pointwise.
dataloading.to_device / PrefetchLoader(augmenter=...) run the chain between the H2D copy and the target pyramid
(dataloading.targets_from_seg: label -1 -> 0, nearest down-sampling), on the loader's side stream.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

TWO_PI = 2.0 * math.pi


def get_patch_size(final_patch_size, rot_x, rot_y, rot_z, scale_range):
    """The loader's initial patch size (reference data_augmentation/compute_initial_patch_size.py:4-25): the final patch
    grown by the worst-case rotation (capped at 90 degrees) and the smallest zoom.  2-D and 3-D."""
    def cap(r):
        r = max(np.abs(r)) if isinstance(r, (tuple, list)) else r
        return min(90 / 360 * 2.0 * np.pi, r)
    rot_x, rot_y, rot_z = cap(rot_x), cap(rot_y), cap(rot_z)
    coords = np.array(final_patch_size, dtype=np.float64)
    shape = coords.copy()

    def rot2(c, a):                                      # batchgenerators rotate_coords_2d: c . R
        R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        return c @ R

    def rot3(c, ax, ay, az):                             # rotate_coords_3d: c . (Rx Ry Rz)
        Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
        Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
        Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
        return c @ (Rx @ Ry @ Rz)
    if len(coords) == 3:
        for a in ((rot_x, 0, 0), (0, rot_y, 0), (0, 0, rot_z)):
            shape = np.max(np.vstack((np.abs(rot3(coords, *a)), shape)), 0)
    else:
        shape = np.max(np.vstack((np.abs(rot2(coords, rot_x)), shape)), 0)
    return (shape / min(scale_range)).astype(int)


def rotation_for_2d(patch_size):
    """nnUNetTrainer.configure_rotation_dummyDA_mirroring_and_inital_patch_size (B:361-377), 2-D branch."""
    deg = 15.0 if max(patch_size) / min(patch_size) > 1.5 else 180.0
    return (-deg / 360 * TWO_PI, deg / 360 * TWO_PI)


# ------------------------------------------------------------------------------------------------
# parameters (host)
# ------------------------------------------------------------------------------------------------
def _two_sided(rng, lo, hi):
    """batchgenerators' draw for multiplicative ranges around 1: below 1 with probability 1/2 (if the range reaches below)."""
    if rng.random_sample() < 0.5 and lo < 1:
        return rng.uniform(lo, 1)
    return rng.uniform(max(lo, 1), hi)


def draw_params(rng, batch, channels, rotation=(-math.pi, math.pi), mirror_axes=(0, 1)):
    """One parameter set per sample, drawn transform by transform in the order of get_training_transforms (B:690-733).
    Returns a dict of numpy arrays (`do_*`: whether the transform touches the sample / channel)."""
    B, C = batch, channels
    p = {k: np.zeros(B, dtype=bool) for k in ("do_rot", "do_scale", "do_noise", "do_blur", "do_bright", "do_contrast", "do_lowres",
                                              "do_gamma_inv", "do_gamma")}
    p.update(angle=np.zeros(B), scale=np.ones(B), noise_std=np.zeros(B), blur_ch=np.zeros((B, C), dtype=bool),
             blur_sigma=np.ones((B, C)), bright=np.ones((B, C)), contrast=np.ones((B, C)), lowres_ch=np.zeros((B, C), dtype=bool),
             lowres_zoom=np.ones((B, C)), gamma_inv=np.ones((B, C)), gamma=np.ones((B, C)), mirror=np.zeros((B, 2), dtype=bool))
    for b in range(B):                                                     # SpatialTransform: p_rot 0.2, p_scale 0.2
        if rng.uniform() < 0.2:
            if rng.uniform() <= 1.0:                                       # p_rot_per_axis = 1 (B:670): drawn, always taken
                p["angle"][b] = rng.uniform(rotation[0], rotation[1])
            p["do_rot"][b] = True
        if rng.uniform() < 0.2:
            p["do_scale"][b], p["scale"][b] = True, _two_sided(rng, 0.7, 1.4)
    for b in range(B):                                                     # GaussianNoise p 0.1, variance U(0, 0.1) used as std
        if rng.uniform() < 0.1:
            p["do_noise"][b], p["noise_std"][b] = True, rng.uniform(0, 0.1)
    for b in range(B):                                                     # GaussianBlur p 0.2, per channel p 0.5, sigma U(0.5, 1)
        if rng.uniform() < 0.2:
            p["do_blur"][b] = True
            for c in range(C):
                if rng.uniform() <= 0.5:
                    p["blur_ch"][b, c], p["blur_sigma"][b, c] = True, rng.uniform(0.5, 1.0)
    for b in range(B):                                                     # BrightnessMultiplicative p 0.15, per channel
        if rng.uniform() < 0.15:
            p["do_bright"][b] = True
            p["bright"][b] = [rng.uniform(0.75, 1.25) for _ in range(C)]
    for b in range(B):                                                     # ContrastAugmentation p 0.15, per channel
        if rng.uniform() < 0.15:
            p["do_contrast"][b] = True
            for c in range(C):
                if rng.uniform() < 1.0:
                    p["contrast"][b, c] = _two_sided(rng, 0.75, 1.25)
    for b in range(B):                                                     # SimulateLowResolution p 0.25, per channel p 0.5
        if rng.uniform() < 0.25:
            p["do_lowres"][b] = True
            for c in range(C):
                if rng.uniform() < 0.5:
                    p["lowres_ch"][b, c], p["lowres_zoom"][b, c] = True, rng.uniform(0.5, 1.0)
    for key_do, key_g, prob in (("do_gamma_inv", "gamma_inv", 0.1), ("do_gamma", "gamma", 0.3)):     # Gamma x2, per channel
        for b in range(B):
            if rng.uniform() < prob:
                p[key_do][b] = True
                p[key_g][b] = [_two_sided(rng, 0.7, 1.5) for _ in range(C)]
    for b in range(B):                                                     # Mirror: each axis with probability 1/2
        for i, ax in enumerate((0, 1)):
            if ax in mirror_axes and rng.uniform() < 0.5:
                p["mirror"][b, i] = True
    return p


# ------------------------------------------------------------------------------------------------
# cubic B-spline machinery (scipy.ndimage semantics)
# ------------------------------------------------------------------------------------------------
_POLE = math.sqrt(3.0) - 2.0                     # pole of the cubic B-spline prefilter
_TAPS = 16                                       # |pole|^16 = 7e-10: the truncated impulse response is exact in fp32
_KERNELS = {}


def _prefilter_kernel(device):
    """Impulse response of the cubic B-spline prefilter, h[k] = 6 z / (z^2 - 1) * z^|k| (z = sqrt(3) - 2), cut at +-16."""
    key = str(device)
    h = _KERNELS.get(key)
    if h is None:
        k = np.arange(-_TAPS, _TAPS + 1)
        h = torch.from_numpy(6.0 * _POLE / (_POLE * _POLE - 1.0) * _POLE ** np.abs(k)).to(device=device, dtype=torch.float32)
        _KERNELS[key] = h
    return h


def _boundary_index(i, n, kind):
    """Source index of position i of an n-long axis extended as scipy's "mirror" (d c b | a b c d | c b a) or "reflect"
    (d c b a | a b c d | d c b a; called symmetric here); any distance (tiny low-resolution images)."""
    if kind == "mirror":
        period = max(2 * (n - 1), 1)
        i = i.abs() % period
        return torch.where(i > n - 1, period - i, i)
    period = 2 * n
    i = torch.where(i < 0, -i - 1, i) % period
    return torch.where(i > n - 1, period - 1 - i, i)


def band_matrix(n, taps, kind, device):
    """(.., K) filter taps (K odd, centred) -> (.., n, n) matrices T with (T x)[i] = sum_k taps[k] x[ext(i + k - K//2)]:
    a 1-D correlation with the boundary rule folded in.  Filters then run as plain (batched) matrix products -- library
    GEMMs for any image size, no convolution-solver search for the ever-changing shapes of the low-resolution transform."""
    K = taps.shape[-1]
    src = _boundary_index(torch.arange(n, device=device).view(n, 1) + torch.arange(K, device=device).view(1, K) - K // 2, n, kind)
    lead = taps.shape[:-1]
    T = torch.zeros(*lead, n, n, device=device, dtype=torch.float32)
    return T.scatter_add_(-1, src.expand(*lead, n, K), taps.unsqueeze(-2).expand(*lead, n, K).contiguous())


_MATRICES = {}


def _prefilter_matrix(n, device):
    key = (n, str(device))
    m = _MATRICES.get(key)                       # loader worker threads share the cache: get / set only, never check-then-read
    if m is None:
        m = band_matrix(n, _prefilter_kernel(device), "mirror", device)
        if len(_MATRICES) > 64:                  # the low-resolution transform asks for ever new sizes
            _MATRICES.clear()
        _MATRICES[key] = m
    return m


def spline_coefficients(x):
    """(.., H, W) image -> cubic B-spline coefficients with scipy's "mirror" boundary (what spline_filter computes for
    mode "constant"): the recursive filter's impulse response decays as 0.268^|k|, so it is a 33-tap band matrix per axis."""
    H, W = x.shape[-2:]
    return _prefilter_matrix(H, x.device) @ x @ _prefilter_matrix(W, x.device).T


def _mirror_index(i, n):
    i = i.abs()
    return torch.where(i > n - 1, 2 * (n - 1) - i, i).clamp_(0, n - 1)


def _bspline3_weights(t):
    """(.., P) fractional offsets -> (.., 4, P) weights of the taps floor - 1 .. floor + 2."""
    t2, t3 = t * t, t * t * t
    return torch.stack([(1 - t) ** 3 / 6.0, (3 * t3 - 6 * t2 + 4) / 6.0, (-3 * t3 + 3 * t2 + 3 * t + 1) / 6.0, t3 / 6.0], -2)


def sample(img, coords, order, cval):
    """scipy.ndimage.map_coordinates(img, coords, order, mode="constant", cval) for a batch: img (B, C, H, W), coords
    (B, 2, Ho, Wo) in pixel units (row, column) -> (B, C, Ho, Wo).  order 3 expects `img` to be spline coefficients.
    All taps of all channels in one gather."""
    B, C, H, W = img.shape
    out_shape = coords.shape[2:]
    y, x = coords[:, 0].reshape(B, -1), coords[:, 1].reshape(B, -1)                # (B, P)
    inside = (y >= 0) & (y <= H - 1) & (x >= 0) & (x <= W - 1)
    fy, fx = torch.floor(y), torch.floor(x)
    ty, tx = y - fy, x - fx
    if order == 3:
        wy, wx, first, n = _bspline3_weights(ty), _bspline3_weights(tx), -1, 4      # (B, 4, P)
    else:
        wy, wx, first, n = torch.stack([1 - ty, ty], -2), torch.stack([1 - tx, tx], -2), 0, 2
    taps = torch.arange(n, device=img.device).view(1, n, 1) + first
    yi = _mirror_index(fy.long().unsqueeze(1) + taps, H)                            # (B, n, P)
    xj = _mirror_index(fx.long().unsqueeze(1) + taps, W)
    idx = (yi.unsqueeze(2) * W + xj.unsqueeze(1)).reshape(B, 1, -1)                 # (B, 1, n * n * P)
    w = (wy.unsqueeze(2) * wx.unsqueeze(1)).reshape(B, 1, n * n, -1)
    vals = img.reshape(B, C, H * W).gather(2, idx.expand(-1, C, -1)).view(B, C, n * n, -1)
    out = (vals * w).sum(2)
    out = torch.where(inside.unsqueeze(1), out, torch.full_like(out, cval))
    return out.view(B, C, *out_shape)


# ------------------------------------------------------------------------------------------------
# transforms (device; per-sample parameters as tensors)
# ------------------------------------------------------------------------------------------------
def spatial_transform(data, seg, patch_size, do, angle, scale, labels=None):
    """batchgenerators augment_spatial (2-D, no elastic deformation, random_crop False): output pixel grid centred on the
    input centre, rotated by `angle` and scaled by `scale` where `do`; other samples are centre-cropped."""
    B, C, Hi, Wi = data.shape
    Ho, Wo = patch_size
    dev = data.device
    gy = torch.arange(Ho, device=dev, dtype=torch.float32) - (Ho - 1) / 2.0
    gx = torch.arange(Wo, device=dev, dtype=torch.float32) - (Wo - 1) / 2.0
    cy, cx = torch.meshgrid(gy, gx, indexing="ij")
    cos, sin = torch.cos(angle).view(B, 1, 1), torch.sin(angle).view(B, 1, 1)
    sc = scale.view(B, 1, 1)
    # rotate_coords_2d: coords^T . [[cos, -sin], [sin, cos]]  -> y' = y cos + x sin, x' = -y sin + x cos; then * scale
    y = (cy * cos + cx * sin) * sc + (Hi / 2.0 - 0.5)
    x = (-cy * sin + cx * cos) * sc + (Wi / 2.0 - 0.5)
    coords = torch.stack([y, x], 1)
    out_d = sample(spline_coefficients(data), coords, 3, 0.0)
    # segmentation: one linear interpolation per label value, ascending; the last label whose interpolated indicator reaches
    # 0.5 is assigned, nothing where none does (interpolate_img, is_seg).  All indicator maps are channels of ONE gather.
    if labels is None:
        lab = torch.unique(seg)                                              # sorted; the dataset's label list avoids this pass
    else:
        lab = labels if torch.is_tensor(labels) else torch.tensor(sorted(labels), device=dev, dtype=seg.dtype)
    onehot = (seg == lab.view(1, -1, 1, 1)).to(torch.float32)               # seg has one channel on this path
    r = sample(onehot, coords, 1, -1.0)                                      # (B, K, Ho, Wo)
    rank = ((r >= 0.5) * torch.arange(1, len(lab) + 1, device=dev).view(1, -1, 1, 1)).amax(1, keepdim=True)
    out_s = torch.where(rank > 0, lab[(rank - 1).clamp_(min=0)], torch.zeros((), device=dev, dtype=seg.dtype))
    # untouched samples: centre crop (crop() with crop_type "center")
    y0, x0 = (Hi - Ho) // 2, (Wi - Wo) // 2
    m = do.view(B, 1, 1, 1)
    return (torch.where(m, out_d, data[:, :, y0:y0 + Ho, x0:x0 + Wo]), torch.where(m, out_s, seg[:, :, y0:y0 + Ho, x0:x0 + Wo]))


def gaussian_blur(data, do, sigma, radius):
    """scipy.ndimage.gaussian_filter(img, sigma, order=0) (truncate 4, mode "reflect") per (sample, channel) where `do`;
    `radius` = int(4 max(sigma) + 0.5), from the host-side parameters (no device read-back)."""
    B, C, H, W = data.shape
    if radius == 0:
        return data
    t = torch.arange(-radius, radius + 1, device=data.device, dtype=torch.float32).view(1, 1, -1)
    s = sigma.view(B, C, 1)
    w = torch.exp(-0.5 * (t / s) ** 2) * (t.abs() <= torch.floor(4.0 * s + 0.5))      # each channel's own truncation
    w = w / w.sum(-1, keepdim=True)
    x = band_matrix(H, w, "symmetric", data.device) @ data @ band_matrix(W, w, "symmetric", data.device).transpose(-1, -2)
    return torch.where(do.view(B, C, 1, 1), x, data)


def simulate_low_resolution(data, do, zoom):
    """SimulateLowResolutionTransform: skimage resize to round(shape * zoom) with order 0, back with order 3 (mode "edge")."""
    B, C, H, W = data.shape
    out = data.clone()
    for b, c in np.argwhere(np.asarray(do)).tolist():         # few (p 0.25 x 0.5): each (sample, channel) has its own low-res size
        z = float(zoom[b, c])
        h, w = int(round(H * z)), int(round(W * z))
        img = data[b, c]
        # order 0, half-pixel-centred grid: source index floor((i + 0.5) * H / h - 0.5 + 0.5)
        f64 = dict(device=data.device, dtype=torch.float64)
        iy = torch.floor((torch.arange(h, **f64) + 0.5) * (H / h)).long().clamp_(0, H - 1)
        ix = torch.floor((torch.arange(w, **f64) + 0.5) * (W / w)).long().clamp_(0, W - 1)
        small = img[iy][:, ix]
        # order 3, mode "edge": scipy pads 12 edge pixels before the prefilter; the half-pixel rim samples that padding
        pad = F.pad(small[None, None], (12, 12, 12, 12), mode="replicate")
        coef = spline_coefficients(pad)
        yy = (((torch.arange(H, **f64) + 0.5) * (h / H) - 0.5) + 12).float()
        xx = (((torch.arange(W, **f64) + 0.5) * (w / W) - 0.5) + 12).float()
        gy, gx = torch.meshgrid(yy, xx, indexing="ij")
        up = sample(coef, torch.stack([gy, gx])[None], 3, 0.0)[0, 0]
        out[b, c] = torch.minimum(torch.maximum(up, small.min()), small.max())      # skimage resize(clip=True)
    return out


def gamma_transform(data, do, gamma, invert):
    """augment_gamma(per_channel, retain_stats, epsilon 1e-7) where `do` (B,); `gamma` (B, C)."""
    x = -data if invert else data
    mn = x.mean((2, 3), keepdim=True)
    sd = x.std((2, 3), keepdim=True, unbiased=False)
    lo = x.amin((2, 3), keepdim=True)
    rng = x.amax((2, 3), keepdim=True) - lo
    y = torch.pow((x - lo) / (rng + 1e-7), gamma.view(*gamma.shape, 1, 1)) * (rng + 1e-7) + lo
    y = y - y.mean((2, 3), keepdim=True)
    y = y / (y.std((2, 3), keepdim=True, unbiased=False) + 1e-8) * sd + mn
    y = -y if invert else y
    return torch.where(do.view(-1, 1, 1, 1), y, data)


def contrast_transform(data, do, factor):
    """augment_contrast(preserve_range, per_channel): (x - mean) * factor + mean, clipped to the channel's old range."""
    mn = data.mean((2, 3), keepdim=True)
    lo, hi = data.amin((2, 3), keepdim=True), data.amax((2, 3), keepdim=True)
    y = torch.minimum(torch.maximum((data - mn) * factor.view(*factor.shape, 1, 1) + mn, lo), hi)
    return torch.where(do.view(-1, 1, 1, 1), y, data)


def mirror_transform(data, seg, flags):
    """MirrorTransform: flags (B, 2): flip rows / columns of that sample."""
    for i, dim in enumerate((2, 3)):
        m = flags[:, i].view(-1, 1, 1, 1)
        data = torch.where(m, data.flip(dim), data)
        seg = torch.where(m, seg.flip(dim), seg)
    return data, seg


class GpuAugmenter:
    """(loader batch on the host) -> augmented (data, [targets]) on the device: the reference's training transform chain
    (B:677-733) behind `dataloading.DataLoader2D`.  `patch_size`: the network's; the loader delivers `get_patch_size(...)`."""

    def __init__(self, patch_size, device, rotation=None, mirror_axes=(0, 1), seed=None, labels=None):
        self.patch_size = tuple(int(v) for v in patch_size)
        self.device = torch.device(device)
        self.rotation = rotation_for_2d(self.patch_size) if rotation is None else rotation
        self.mirror_axes = tuple(mirror_axes)
        # every value the loader's segmentation can hold (label_manager.all_labels and the -1 padding), ascending
        self.labels = None if labels is None else torch.tensor(sorted(set(labels) | {-1}), dtype=torch.float32, device=self.device)
        self.rng = np.random.RandomState(seed)

    def initial_patch_size(self):
        return tuple(int(v) for v in get_patch_size(self.patch_size, self.rotation, (0, 0), (0, 0), (0.85, 1.25)))

    def apply(self, data, seg, p, noise=None):
        """The transform chain with given parameters (`draw_params` layout); data (B, C, Hi, Wi) fp32, seg (B, 1, Hi, Wi)."""
        dev = data.device
        T = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a), device=dev).to(dt)      # noqa: E731
        data, seg = spatial_transform(data, seg.to(torch.float32), self.patch_size, T(p["do_rot"] | p["do_scale"], torch.bool),
                                      T(p["angle"] * p["do_rot"]), T(np.where(p["do_scale"], p["scale"], 1.0)),
                                      None if self.labels is None else self.labels.to(data.device))
        if noise is None:
            noise = torch.randn_like(data)
        data = data + noise * T(p["noise_std"] * p["do_noise"]).view(-1, 1, 1, 1)
        blur = p["blur_ch"] & p["do_blur"][:, None]
        radius = int(4.0 * float(p["blur_sigma"][blur].max()) + 0.5) if blur.any() else 0
        data = gaussian_blur(data, T(blur, torch.bool), T(p["blur_sigma"]), radius)
        data = torch.where(T(p["do_bright"], torch.bool).view(-1, 1, 1, 1), data * T(p["bright"]).view(*p["bright"].shape, 1, 1), data)
        data = contrast_transform(data, T(p["do_contrast"], torch.bool), T(p["contrast"]))
        data = simulate_low_resolution(data, p["lowres_ch"] & p["do_lowres"][:, None], p["lowres_zoom"])
        data = gamma_transform(data, T(p["do_gamma_inv"], torch.bool), T(p["gamma_inv"]), invert=True)
        data = gamma_transform(data, T(p["do_gamma"], torch.bool), T(p["gamma"]), invert=False)
        return mirror_transform(data, seg, T(p["mirror"], torch.bool))

    def clone(self, seed):
        """The same chain with its own parameter stream (one per loader worker)."""
        twin = GpuAugmenter(self.patch_size, self.device, self.rotation, self.mirror_axes, seed)
        twin.labels = self.labels
        return twin

    def __call__(self, data, seg):
        """(B, C, Hi, Wi) data and (B, 1, Hi, Wi) seg of the loader's initial patch size, on the device -> the augmented
        (B, C, H, W) / (B, 1, H, W) pair (labels still carry -1; dataloading.targets_from_seg follows)."""
        p = draw_params(self.rng, data.shape[0], data.shape[1], self.rotation, self.mirror_axes)
        return self.apply(data, seg, p)
