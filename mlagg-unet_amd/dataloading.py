"""Real-data input path of the 2-D train step: the reference's nnUNetDataLoader2D
(training/dataloading/data_loader_2d.py:7-86 on base_data_loader.py:10-139 and nnunet_dataset.py:80-111) re-shaped
for feeding an MI355X, not re-invented: same case folder (<case>.npz or unpacked <case>.npy / <case>_seg.npy, <case>.pkl
with `class_locations`), same sampling decisions in the same order of random draws -- so a seeded run gives the reference's
batches bit for bit -- but

  * only the selected 2-D slice of a case is read (memory-mapped .npy) and the crop is written straight into PINNED host
    batch buffers (the reference loads the case, slices, crops, np.pads into fresh arrays and later collates / converts);
  * `to_device` moves a batch with non-blocking copies on a side HIP stream and builds the five deep-supervision targets
    ON the device (label -1 -> 0 as RemoveLabelTransform, nearest down-sampling as DownsampleSegForDSTransform2), i.e. the
    (data, [target_0..target_4]) pair nnUNetTrainer.train_step consumes (nnUNetTrainer.py:833-845);
  * `PrefetchLoader` overlaps batch assembly (worker threads, one RandomState each) and the H2D copy with the train step.

Out of scope: batchgenerators' augmentation transforms (third-party, absent offline)."""
import os
import pickle
import queue
import threading

import numpy as np
import torch


class Dataset:
    """Case folder index (reference nnUNetDataset).  Properties (.pkl) are read once and cached: they hold the sampled
    foreground locations used for every oversampled patch."""

    def __init__(self, folder, case_identifiers=None):
        ids = case_identifiers if case_identifiers is not None else \
            [i[:-4] for i in os.listdir(folder) if i.endswith("npz") and i.find("segFromPrevStage") == -1]
        self.folder, self.ids = folder, sorted(ids)
        self._props, self._open = {}, {}
        if not self.ids:
            raise RuntimeError(f"no <case>.npz in {folder}")

    def keys(self):
        return list(self.ids)

    def __len__(self):
        return len(self.ids)

    def properties(self, key):
        if key not in self._props:
            with open(os.path.join(self.folder, key + ".pkl"), "rb") as fh:
                self._props[key] = pickle.load(fh)
        return self._props[key]

    def arrays(self, key):
        """(data (C, D, H, W), seg (1, D, H, W)): memory maps of the unpacked .npy files when they exist (kept open),
        else the decompressed .npz members."""
        if key not in self._open:
            base = os.path.join(self.folder, key)
            if os.path.isfile(base + ".npy") and os.path.isfile(base + "_seg.npy"):
                self._open[key] = (np.load(base + ".npy", "r"), np.load(base + "_seg.npy", "r"))
            else:
                z = np.load(base + ".npz")
                return z["data"], z["seg"]                    # not cached: a decompressed case can be large
        return self._open[key]


class DataLoader2D:
    """generate_train_batch() == the reference's, draw for draw.  `rng`: numpy's global state by default (what the
    reference uses); a RandomState for worker threads."""

    def __init__(self, dataset, batch_size, patch_size, final_patch_size, all_labels, oversample_foreground_percent=0.0,
                 rng=None, pin_memory=None, has_ignore=False):
        self.has_ignore = bool(has_ignore)             # label_manager.has_ignore_label (base_data_loader.py:42)
        self.ds, self.batch_size = dataset, int(batch_size)
        self.indices = dataset.keys()
        self.patch_size = tuple(int(v) for v in patch_size)
        self.need_to_pad = (np.array(patch_size) - np.array(final_patch_size)).astype(int)
        self.oversample = float(oversample_foreground_percent)
        self.annotated_classes_key = tuple(all_labels)
        flat = [int(v) for lab in all_labels for v in (lab if isinstance(lab, (tuple, list)) else (lab,))]
        # a larger value in a case file is refused (generate_train_batch); nnU-Net's ignore label is the next integer
        self.max_label = (max(flat) + int(self.has_ignore)) if flat else None
        self.rng = np.random if rng is None else rng
        data, seg = dataset.arrays(self.indices[0])
        self.channels, self.seg_channels = data.shape[0], seg.shape[0]
        self.pin = torch.cuda.is_available() if pin_memory is None else bool(pin_memory)

    def get_do_oversample(self, j):                    # base_data_loader.py:45-49
        return not j < round(self.batch_size * (1 - self.oversample))

    def _bbox(self, shape, force_fg, locs):            # base_data_loader.py:63-139; `locs`: voxels of the chosen class / region
        # on the chosen slice (with an ignore label and no forced foreground: of ALL annotated labels, :91-97)
        need = self.need_to_pad.copy()
        for d in range(2):
            if need[d] + shape[d] < self.patch_size[d]:
                need[d] = self.patch_size[d] - shape[d]
        lbs = [-need[i] // 2 for i in range(2)]
        ubs = [shape[i] + need[i] // 2 + need[i] % 2 - self.patch_size[i] for i in range(2)]
        if (force_fg or self.has_ignore) and locs is not None and len(locs) > 0:
            v = locs[self.rng.choice(len(locs))]
            return [max(lbs[i], int(v[i + 1]) - self.patch_size[i] // 2) for i in range(2)]
        return [self.rng.randint(lbs[i], ubs[i] + 1) for i in range(2)]

    def _buffers(self):
        shp_d = (self.batch_size, self.channels, *self.patch_size)
        shp_s = (self.batch_size, self.seg_channels, *self.patch_size)
        d = torch.zeros(shp_d, dtype=torch.float32, pin_memory=self.pin)
        s = torch.full(shp_s, -1, dtype=torch.int16, pin_memory=self.pin)
        return d, s

    def generate_train_batch(self):
        keys = self.rng.choice(self.indices, self.batch_size, replace=True, p=None)
        data_t, seg_t = self._buffers()
        data_all, seg_all = data_t.numpy(), seg_t.numpy()
        for j, key in enumerate(keys):
            force_fg = self.get_do_oversample(j)
            data, seg = self.ds.arrays(key)
            sel = locs_all = None
            if force_fg:
                cl = self.ds.properties(key)["class_locations"]
                eligible = [i for i in cl.keys() if len(cl[i]) > 0]
                # data_loader_2d.py:29-35: the all-annotated-labels key competes only when nothing else is present
                if len(eligible) > 1 and self.annotated_classes_key in [i for i in eligible if isinstance(i, tuple)]:
                    eligible.remove(self.annotated_classes_key)
                if eligible:
                    sel = eligible[self.rng.choice(len(eligible))]
                    locs_all = cl[sel]
            elif self.has_ignore:                          # data_loader_2d.py:21-23
                sel = self.annotated_classes_key
                locs_all = self.ds.properties(key)["class_locations"][sel]
            sl = self.rng.choice(locs_all[:, 1]) if sel is not None else self.rng.choice(len(data[0]))
            if sel is not None:
                # data_loader_2d.py:55-57: the locations of that class on that slice; get_bbox is told the class
                # (overwrite_class), so it draws a voxel but no class
                locs = locs_all[locs_all[:, 1] == sl][:, (0, 2, 3)]
            else:
                locs = None
            shape = data.shape[2:]
            lb = self._bbox(shape, force_fg if sel is not None else None, locs)
            ub = [lb[i] + self.patch_size[i] for i in range(2)]
            v0 = [max(0, lb[i]) for i in range(2)]
            v1 = [min(shape[i], ub[i]) for i in range(2)]
            o0 = [v0[i] - lb[i] for i in range(2)]
            h, w = v1[0] - v0[0], v1[1] - v0[1]
            data_all[j, :, o0[0]:o0[0] + h, o0[1]:o0[1] + w] = data[:, sl, v0[0]:v1[0], v0[1]:v1[1]]
            seg_all[j, :, o0[0]:o0[0] + h, o0[1]:o0[1] + w] = seg[:, sl, v0[0]:v1[0], v0[1]:v1[1]]
        if self.max_label is not None and int(seg_all.max()) > self.max_label:
            # torch's nll_loss (the reference's CE) fails on a label >= C; the fused loss kernel would quietly count such a
            # pixel as "no class hit", so a corrupt case is stopped here, on the host, where the check costs nothing
            raise RuntimeError(f"segmentation label {int(seg_all.max())} > {self.max_label} (largest label of the dataset) "
                               f"in cases {sorted(set(map(str, keys)))}")
        return {"data": data_t, "seg": seg_t, "keys": keys}


def targets_from_seg(seg, n_levels=5):
    """RemoveLabelTransform(-1, 0) (B:701) + DownsampleSegForDSTransform2, order 0 (B:730): the deep-supervision targets."""
    seg = torch.where(seg < 0, torch.zeros_like(seg), seg)
    targets = [seg]
    for s in range(1, n_levels):
        size = (seg.shape[2] >> s, seg.shape[3] >> s)
        targets.append(torch.nn.functional.interpolate(seg, size=size, mode="nearest-exact"))
    return targets


def to_device(batch, device, n_levels=5, stream=None, augmenter=None):
    """Host batch -> (data (B, C, H, W) fp32, [target_s (B, 1, H/2^s, W/2^s) fp32 labels]) on `device`; with an
    `augmenter` (augmentation.GpuAugmenter) the training transforms run on the device in between."""
    device = torch.device(device)
    ctx = torch.cuda.stream(stream) if stream is not None else _Null()
    with ctx:
        data = batch["data"].to(device, non_blocking=True)
        seg = batch["seg"].to(device, non_blocking=True).float()
        if augmenter is not None:
            data, seg = augmenter(data, seg)
        targets = targets_from_seg(seg, n_levels)
    return data, targets


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class PrefetchLoader:
    """Background batch assembly + H2D copy.  Each worker thread owns a clone of the loader with its own RandomState
    (numpy's slicing / copies release the GIL); batches arrive on the device through a side stream, and `next()` makes the
    caller's current stream wait for that copy only."""

    def __init__(self, loader, device, num_workers=4, depth=6, seed=1234, n_levels=5, augmenter=None):
        self.device = torch.device(device)
        self.q = queue.Queue(maxsize=depth)
        self.stop = threading.Event()
        self.n_levels = n_levels
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.workers = []
        for i in range(num_workers):
            clone = DataLoader2D(loader.ds, loader.batch_size, loader.patch_size,
                                 tuple(np.array(loader.patch_size) - loader.need_to_pad), loader.annotated_classes_key,
                                 loader.oversample, rng=np.random.RandomState(seed + i), pin_memory=loader.pin,
                                 has_ignore=loader.has_ignore)
            aug = augmenter.clone(seed + 7919 * (i + 1)) if augmenter is not None else None
            t = threading.Thread(target=self._work, args=(clone, aug), daemon=True)
            t.start()
            self.workers.append(t)

    def _work(self, loader, augmenter=None):
        while not self.stop.is_set():
            try:
                b = loader.generate_train_batch()
                if self.copy_stream is not None:
                    data, targets = to_device(b, self.device, self.n_levels, self.copy_stream, augmenter)
                    ev = torch.cuda.Event()
                    ev.record(self.copy_stream)
                else:
                    data, targets = to_device(b, self.device, self.n_levels, None, augmenter)
                    ev = None
                item = (data, targets, ev, b)                 # keep the pinned host batch alive until consumed
            except BaseException as e:                        # missing .pkl, bad .npz, out of memory on the copy stream ...
                item = e                                      # handed to the consumer: next() re-raises it
                self.stop.set()
            while True:
                try:
                    self.q.put(item, timeout=0.2)
                    break
                except queue.Full:
                    if self.stop.is_set() and not isinstance(item, BaseException):
                        break

    def next(self):
        while True:
            try:
                item = self.q.get(timeout=1.0)
                break
            except queue.Empty:
                if not any(t.is_alive() for t in self.workers):
                    raise RuntimeError("PrefetchLoader: every worker thread has exited and no batch is queued")
        if isinstance(item, BaseException):
            self.q.put(item)                                  # the next caller sees it too
            raise RuntimeError(f"PrefetchLoader worker failed: {type(item).__name__}: {item}") from item
        data, targets, ev, _ = item
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t in [data] + targets:
                t.record_stream(torch.cuda.current_stream(self.device))
        return data, targets

    def close(self):
        self.stop.set()
        while not self.q.empty():
            try:
                self.q.get_nowait()
            except queue.Empty:
                break
        for t in self.workers:
            t.join(timeout=2)
