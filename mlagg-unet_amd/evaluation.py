"""DSC evaluation: the numbers behind "mean DSC vs reference".

  * validation_step / validation_epoch_end -- the in-loop pseudo-Dice of the reference trainer
    (nnUNetTrainer.py:880-942, 944-978): argmax of the full-resolution head, hard tp/fp/fn per
    foreground class, 2tp / (2tp + fp + fn), nanmean over classes.  Device side: ONE bincount of
    (target * C + prediction) gives the whole confusion matrix (the reference scatters two one-hot
    volumes of B*C*H*W floats and multiplies them three times); DDP sums it with one all_reduce
    (the reference pickles numpy arrays through all_gather_object).
  * compute_dice_coefficient / abdomen_case_dsc / abdomen_mean_dsc -- the offline per-organ DSC of
    evaluation/SurfaceDice.py:481-498 and evaluation/abdomen_DSC_Eval.py:80-113 on label volumes.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from . import trainer

ABDOMEN_ORGANS = ("Liver", "RK", "Spleen", "Pancreas", "Aorta", "IVC", "RAG", "LAG", "Gallbladder", "Esophagus",
                  "Stomach", "Duodenum", "LK")                      # labels 1..13, abdomen_DSC_Eval.py:48-50
SLAB_LABELS = (5, 6, 10)                                            # Aorta, IVC, Esophagus: labelled slices only


def confusion_matrix(pred_labels, target_labels, num_classes):
    """(C, C) int64 counts, row = target label, column = predicted label; stays on the inputs' device."""
    t = target_labels.reshape(-1).long()
    p = pred_labels.reshape(-1).long()
    if t.numel() != p.numel():
        raise RuntimeError("prediction and target differ in size")
    return torch.bincount(t * num_classes + p, minlength=num_classes * num_classes).view(num_classes, num_classes)


def hard_tp_fp_fn(logits, target, ignore_label=None):
    """Foreground tp / fp / fn of argmax(logits) against a label map (reference :899-940, background dropped).  Pixels that
    carry ``ignore_label`` are left out of all three counts (reference B:917-929: ``mask = target != ignore``, the target is
    zeroed there and ``get_tp_fp_fn_tn(..., mask=mask)`` multiplies every count by the mask)."""
    C = logits.shape[1]
    pred, tgt = logits.argmax(1).reshape(-1), target.reshape(-1).long()
    if ignore_label is None:
        cm = confusion_matrix(pred, tgt, C)
    else:
        # no boolean-mask indexing (a nonzero() and a host synchronisation per validation step): ignored pixels are counted into one
        # extra bin that is dropped
        keep = tgt != int(ignore_label)
        idx = torch.where(keep, tgt * C + pred, torch.full_like(tgt, C * C))
        cm = torch.bincount(idx, minlength=C * C + 1)[:C * C].view(C, C)
    tp = cm.diagonal()
    return tp[1:], (cm.sum(0) - tp)[1:], (cm.sum(1) - tp)[1:]


@torch.no_grad()
def validation_step(network, data, target, batch_dice=True, ddp=False, ignore_label=None, loss_fn=None):
    """reference validation_step: {'loss', 'tp_hard', 'fp_hard', 'fn_hard'} (device tensors, no host sync).  ``loss_fn``:
    the trainer's own loss (B:897 ``self.loss(output, target)``); default: the fused Dice + CE deep-supervision loss with
    ``ignore_label`` masked inside K9."""
    output = network(data)
    if not isinstance(output, (list, tuple)):
        output, target = [output], target if isinstance(target, (list, tuple)) else [target]
    if loss_fn is not None:
        loss = loss_fn(output, target)
    else:
        loss = trainer.deep_supervision_loss(output, target, batch_dice=batch_dice, ddp=ddp, ignore_label=ignore_label)
    tp, fp, fn = hard_tp_fp_fn(output[0], target[0], ignore_label)
    return {"loss": loss.detach(), "tp_hard": tp, "fp_hard": fp, "fn_hard": fn}


def validation_epoch_end(val_outputs, group=None):
    """reference on_validation_epoch_end: {'mean_fg_dice', 'dice_per_class_or_region', 'val_losses'}."""
    tp = torch.stack([o["tp_hard"] for o in val_outputs]).sum(0)
    fp = torch.stack([o["fp_hard"] for o in val_outputs]).sum(0)
    fn = torch.stack([o["fn_hard"] for o in val_outputs]).sum(0)
    loss = torch.stack([o["loss"].reshape(()) for o in val_outputs]).double().mean()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        packed = torch.cat([tp, fp, fn]).double()
        packed = torch.cat([packed, loss.reshape(1).to(packed.device)])
        dist.all_reduce(packed, group=group)
        n = tp.numel()
        tp, fp, fn = packed[:n], packed[n:2 * n], packed[2 * n:3 * n]
        loss = packed[-1] / dist.get_world_size(group)       # every rank runs the same number of iterations
    tp, fp, fn = (v.double().cpu().numpy() for v in (tp, fp, fn))
    with np.errstate(invalid="ignore", divide="ignore"):
        per_class = 2 * tp / (2 * tp + fp + fn)
    return {"mean_fg_dice": float(np.nanmean(per_class)), "dice_per_class_or_region": [float(v) for v in per_class],
            "val_losses": float(loss)}


def compute_dice_coefficient(mask_gt, mask_pred):
    """SurfaceDice.py:481-498: 2|A & B| / (|A| + |B|); NaN when both masks are empty."""
    volume_sum = int(mask_gt.sum()) + int(mask_pred.sum())
    if volume_sum == 0:
        return float("nan")
    return 2 * int((mask_gt & mask_pred).sum()) / volume_sum


def abdomen_case_dsc(gt, seg, organs=ABDOMEN_ORGANS, slab_labels=SLAB_LABELS):
    """Per-organ DSC of one case (label volumes indexed [x, y, z]), abdomen_DSC_Eval.py:88-106, rounded to 4
    digits as the script stores them.  Runs where the volumes live: torch tensors on the GPU are reduced
    there (one confusion matrix for the whole-volume organs), numpy arrays on the host."""
    if isinstance(gt, np.ndarray):
        gt, seg = torch.from_numpy(gt.astype(np.int64)), torch.from_numpy(np.asarray(seg).astype(np.int64))
    n = len(organs) + 1
    gt, seg = gt.long(), seg.long()
    gt = torch.where((gt > 0) & (gt < n), gt, 0)          # labels outside the organ list count as "not organ i"
    seg = torch.where((seg > 0) & (seg < n), seg, 0)
    cm = confusion_matrix(seg, gt, n).cpu()
    vol_gt, vol_seg = cm.sum(1), cm.sum(0)
    out = OrderedDict()
    for i, organ in enumerate(organs, 1):
        g, s = int(vol_gt[i]), int(vol_seg[i])
        if g == 0 and s == 0:
            d = 1
        elif g == 0 and s > 0:
            d = 0
        elif i in slab_labels:
            z = torch.nonzero((gt == i).flatten(0, 1).any(0)).flatten()
            lo, hi = int(z.min()), int(z.max())
            a, b = gt[:, :, lo:hi] == i, seg[:, :, lo:hi] == i        # the script's half-open z range
            d = compute_dice_coefficient(a, b)
        else:
            d = 2 * int(cm[i, i]) / (g + s)
        out[organ] = round(d, 4)
    return out


def abdomen_mean_dsc(cases):
    """Column means over cases, then their mean (abdomen_DSC_Eval.py:108-114; pandas means skip NaN)."""
    organs = list(cases[0].keys())
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)       # an organ that is NaN in every case stays NaN
        cols = OrderedDict((o, float(np.nanmean([c[o] for c in cases]))) for o in organs)
    return cols, float(np.nanmean(list(cols.values())))
