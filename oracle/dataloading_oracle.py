"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the real-data input path of SURVEY.md section 8(f)-1: plain restatements of
  * nnUNetDataset.load_case / get_case_identifiers        (training/dataloading/nnunet_dataset.py:80-111, utils.py:39-44)
  * nnUNetDataLoaderBase: shapes, oversampling rule, get_bbox   (training/dataloading/base_data_loader.py:10-139)
  * nnUNetDataLoader2D.generate_train_batch                 (training/dataloading/data_loader_2d.py:7-86)
  * DefaultPreprocessor._sample_foreground_locations        (preprocessing/preprocessors/default_preprocessor.py:134-160)
consuming numpy's GLOBAL random state in the reference's order, so that a seeded run reproduces the reference's batches
bit for bit (tests/golden/dataloader.npz).  `get_indices` comes from batchgenerators' DataLoader (third-party, absent
offline): restated from its published behaviour for infinite=True -- np.random.choice(indices, batch_size, replace=True, p)
-- unpinned.  The augmentation transforms (batchgenerators) are out of scope."""
import os
import pickle

import numpy as np


def sample_foreground_locations(seg, classes, seed=1234):
    num_samples, min_cov = 10000, 0.01
    rnd = np.random.RandomState(seed)
    out = {}
    for c in classes:
        locs = np.argwhere(seg == c)
        if len(locs) == 0:
            out[c] = []
            continue
        n = max(min(num_samples, len(locs)), int(np.ceil(len(locs) * min_cov)))
        out[c] = locs[rnd.choice(len(locs), n, replace=False)]
    return out


def sample_locations_of(seg, classes_or_regions, seed=1234):
    """DefaultPreprocessor._sample_foreground_locations (preprocessing/preprocessors/default_preprocessor.py:206-233): a key
    may be one label or a tuple of labels (a region, or -- with an ignore label -- all annotated labels)."""
    num_samples, min_cov = 10000, 0.01
    rnd = np.random.RandomState(seed)
    out = {}
    for c in classes_or_regions:
        k = c if not isinstance(c, list) else tuple(c)
        if isinstance(c, (tuple, list)):
            mask = np.zeros_like(seg, dtype=bool)
            for ci in c:
                mask |= seg == ci
            locs = np.argwhere(mask)
        else:
            locs = np.argwhere(seg == c)
        if len(locs) == 0:
            out[k] = []
            continue
        n = max(min(num_samples, len(locs)), int(np.ceil(len(locs) * min_cov)))
        out[k] = locs[rnd.choice(len(locs), n, replace=False)]
    return out


def write_synthetic_dataset(folder, n_cases=4, seed=3, labels=(1, 2, 3), unpack=False, small=True, ignore_label=None):
    """A tiny `nnUNet_preprocessed/<dataset>/2d`-style folder: <case>.npz {data (C, D, H, W) f32, seg (1, D, H, W) i16 with
    -1 outside the 'nonzero' region} + <case>.pkl {class_locations}.  Case 1 has no foreground; case 2 is smaller than
    the patch in one axis (when `small`)."""
    os.makedirs(folder, exist_ok=True)
    rng = np.random.RandomState(seed)
    for i in range(n_cases):
        D = 5 + i
        H, W = (44 + 6 * i, 52 - 4 * i) if small else (300 + 8 * i, 280 + 12 * i)
        if small and i == 2:
            H = 20
        data = rng.standard_normal((1, D, H, W)).astype(np.float32)
        seg = np.zeros((1, D, H, W), dtype=np.int16)
        if i != 1:
            for lab in labels:
                d, y, x = rng.randint(0, D), rng.randint(2, H - 8), rng.randint(2, W - 8)
                seg[0, d:d + 2, y:y + 6, x:x + 6] = lab
        seg[0, :, :2, :] = -1
        name = f"case_{i:03d}"
        if ignore_label is not None:
            # partially annotated case: a band of every slice carries the ignore label; class_locations also gets the
            # tuple of all annotated labels (default_preprocessor.py:118-125)
            seg[0, :, H // 2:H // 2 + 6, :] = ignore_label
            locs = sample_locations_of(seg, list(labels) + [[0] + list(labels)])
        else:
            locs = sample_foreground_locations(seg, list(labels))
        np.savez_compressed(os.path.join(folder, name + ".npz"), data=data, seg=seg)
        with open(os.path.join(folder, name + ".pkl"), "wb") as fh:
            pickle.dump({"class_locations": locs}, fh)
        if unpack:
            np.save(os.path.join(folder, name + ".npy"), data)
            np.save(os.path.join(folder, name + "_seg.npy"), seg)


class Dataset:
    def __init__(self, folder):
        ids = sorted(i[:-4] for i in os.listdir(folder) if i.endswith("npz") and i.find("segFromPrevStage") == -1)
        self.dataset = {c: {"data_file": os.path.join(folder, c + ".npz"), "properties_file": os.path.join(folder, c + ".pkl")}
                        for c in ids}

    def keys(self):
        return self.dataset.keys()

    def load_case(self, key):
        e = self.dataset[key]
        with open(e["properties_file"], "rb") as fh:
            props = pickle.load(fh)
        base = e["data_file"][:-4]
        data = np.load(base + ".npy", "r") if os.path.isfile(base + ".npy") else np.load(e["data_file"])["data"]
        seg = np.load(base + "_seg.npy", "r") if os.path.isfile(base + "_seg.npy") else np.load(e["data_file"])["seg"]
        return data, seg, props


class DataLoader2D:
    def __init__(self, dataset, batch_size, patch_size, final_patch_size, all_labels, oversample_foreground_percent=0.0,
                 has_ignore=False):
        self.has_ignore = has_ignore
        self.ds, self.batch_size = dataset, batch_size
        self.indices = list(dataset.keys())
        self.patch_size, self.final_patch_size = patch_size, final_patch_size
        self.oversample = oversample_foreground_percent
        self.need_to_pad = (np.array(patch_size) - np.array(final_patch_size)).astype(int)
        self.annotated_classes_key = tuple(all_labels)
        data, seg, _ = dataset.load_case(self.indices[0])
        self.data_shape = (batch_size, data.shape[0], *patch_size)
        self.seg_shape = (batch_size, seg.shape[0], *patch_size)

    def get_do_oversample(self, j):
        return not j < round(self.batch_size * (1 - self.oversample))

    def get_bbox(self, shape, force_fg, class_locations, overwrite_class=None):
        need = self.need_to_pad.copy()
        dim = len(shape)
        for d in range(dim):
            if need[d] + shape[d] < self.patch_size[d]:
                need[d] = self.patch_size[d] - shape[d]
        lbs = [-need[i] // 2 for i in range(dim)]
        ubs = [shape[i] + need[i] // 2 + need[i] % 2 - self.patch_size[i] for i in range(dim)]
        if not force_fg and not self.has_ignore:
            lb = [np.random.randint(lbs[i], ubs[i] + 1) for i in range(dim)]
        else:
            if not force_fg:                                   # ignore label: a patch around an ANNOTATED voxel (:91-97)
                sel = self.annotated_classes_key
                if len(class_locations[sel]) == 0:
                    sel = None
            else:
                eligible = [i for i in class_locations.keys() if len(class_locations[i]) > 0]
                tmp = [i == self.annotated_classes_key if isinstance(i, tuple) else False for i in eligible]
                if any(tmp) and len(eligible) > 1:
                    eligible.pop(np.where(tmp)[0][0])
                if len(eligible) == 0:
                    sel = None
                else:
                    sel = eligible[np.random.choice(len(eligible))] if (overwrite_class is None or overwrite_class not in eligible) \
                        else overwrite_class
            vox = class_locations[sel] if sel is not None else None
            if vox is not None and len(vox) > 0:
                v = vox[np.random.choice(len(vox))]
                lb = [max(lbs[i], v[i + 1] - self.patch_size[i] // 2) for i in range(dim)]
            else:
                lb = [np.random.randint(lbs[i], ubs[i] + 1) for i in range(dim)]
        return lb, [lb[i] + self.patch_size[i] for i in range(dim)]

    def generate_train_batch(self):
        keys = np.random.choice(self.indices, self.batch_size, replace=True, p=None)         # batchgenerators get_indices
        data_all = np.zeros(self.data_shape, dtype=np.float32)
        seg_all = np.zeros(self.seg_shape, dtype=np.int16)
        for j, key in enumerate(keys):
            force_fg = self.get_do_oversample(j)
            data, seg, props = self.ds.load_case(key)
            if not force_fg:
                sel = self.annotated_classes_key if self.has_ignore else None
            else:
                eligible = [i for i in props["class_locations"].keys() if len(props["class_locations"][i]) > 0]
                tmp = [i == self.annotated_classes_key if isinstance(i, tuple) else False for i in eligible]
                if any(tmp) and len(eligible) > 1:
                    eligible.pop(np.where(tmp)[0][0])
                sel = eligible[np.random.choice(len(eligible))] if len(eligible) > 0 else None
            if sel is not None:
                sl = np.random.choice(props["class_locations"][sel][:, 1])
            else:
                sl = np.random.choice(len(data[0]))
            data, seg = data[:, sl], seg[:, sl]
            locs = {sel: props["class_locations"][sel][props["class_locations"][sel][:, 1] == sl][:, (0, 2, 3)]} \
                if sel is not None else None
            shape = data.shape[1:]
            lb, ub = self.get_bbox(shape, force_fg if sel is not None else None, locs, overwrite_class=sel)
            vlb = [max(0, lb[i]) for i in range(2)]
            vub = [min(shape[i], ub[i]) for i in range(2)]
            data = data[:, vlb[0]:vub[0], vlb[1]:vub[1]]
            seg = seg[:, vlb[0]:vub[0], vlb[1]:vub[1]]
            pad = [(-min(0, lb[i]), max(ub[i] - shape[i], 0)) for i in range(2)]
            data_all[j] = np.pad(data, ((0, 0), *pad), "constant", constant_values=0)
            seg_all[j] = np.pad(seg, ((0, 0), *pad), "constant", constant_values=-1)
        return {"data": data_all, "seg": seg_all, "keys": keys}


def downsample_seg_for_ds(seg, ds_scales):
    """DownsampleSegForDSTransform2 (training/data_augmentation/custom_transforms/deep_supervision_donwsampling.py:7-55,
    order 0) after RemoveLabelTransform(-1, 0) (nnUNetTrainer.py:713): the list of targets train_step consumes (B:838-839).
    batchgenerators' resize_segmentation (third-party, absent offline -- unpinned) is, for order 0, skimage's
    resize(order=0, mode="edge", anti_aliasing=False), which is scipy.ndimage.zoom(order=0, mode="nearest",
    grid_mode=True); restated with scipy here."""
    import numpy as np
    from scipy import ndimage
    seg = np.where(seg < 0, 0, seg).astype(np.float32)
    out = []
    for s in ds_scales:
        if all(v == 1 for v in s):
            out.append(seg)
            continue
        new_shape = np.round(np.array(seg.shape[2:], dtype=float) * np.array(s)).astype(int)
        level = np.zeros(seg.shape[:2] + tuple(new_shape), dtype=seg.dtype)
        for b in range(seg.shape[0]):
            for c in range(seg.shape[1]):
                zoom = [n / o for n, o in zip(new_shape, seg.shape[2:])]
                level[b, c] = ndimage.zoom(seg[b, c], zoom, order=0, mode="nearest", grid_mode=True)
        out.append(level)
    return out
