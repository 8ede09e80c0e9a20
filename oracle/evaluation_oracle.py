"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the DSC evaluation row of SURVEY.md section 8(f)-3: plain restatements of
  * nnUNetTrainer.validation_step's online evaluation (nnUNetTrainer.py:899-940) on top of
    get_tp_fp_fn_tn (training/loss/dice.py:120-178): one-hot scatter, products, sums over batch + space;
  * on_validation_epoch_end (:944-978);
  * compute_dice_coefficient (evaluation/SurfaceDice.py:481-498) and the per-case loop of
    evaluation/abdomen_DSC_Eval.py:80-106 (nibabel I/O left out: the inputs are the label volumes).
Pinned by tests/golden/evaluation.npz (get_tp_fp_fn_tn and compute_dice_coefficient imported from the reference;
abdomen_DSC_Eval.py is a nibabel script and cannot run here, its loop is restated only)."""
import numpy as np
import torch


def hard_tp_fp_fn(logits, target):
    seg = logits.argmax(1)[:, None]
    onehot = torch.zeros(logits.shape, dtype=torch.float32)
    onehot.scatter_(1, seg, 1)
    y = torch.zeros(logits.shape)
    y.scatter_(1, target.long(), 1)
    axes = [0] + list(range(2, logits.ndim))
    tp = (onehot * y).sum(axes)
    fp = (onehot * (1 - y)).sum(axes)
    fn = ((1 - onehot) * y).sum(axes)
    return tp[1:].numpy(), fp[1:].numpy(), fn[1:].numpy()


def epoch_end(outputs):
    tp = np.sum([o["tp_hard"] for o in outputs], 0)
    fp = np.sum([o["fp_hard"] for o in outputs], 0)
    fn = np.sum([o["fn_hard"] for o in outputs], 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        per_class = [2 * i / (2 * i + j + k) for i, j, k in zip(tp, fp, fn)]
    return float(np.nanmean(per_class)), per_class, float(np.mean([o["loss"] for o in outputs]))


def compute_dice_coefficient(mask_gt, mask_pred):
    volume_sum = mask_gt.sum() + mask_pred.sum()
    if volume_sum == 0:
        return np.nan
    return 2 * (mask_gt & mask_pred).sum() / volume_sum


def abdomen_case_dsc(gt, seg, n_organs=13):
    gt, seg = np.uint8(gt), np.uint8(seg)
    out = []
    for i in range(1, n_organs + 1):
        if np.sum(gt == i) == 0 and np.sum(seg == i) == 0:
            d = 1
        elif np.sum(gt == i) == 0 and np.sum(seg == i) > 0:
            d = 0
        else:
            if i in (5, 6, 10):
                z = np.where(gt == i)[2]
                lo, hi = np.min(z), np.max(z)
                a, b = gt[:, :, lo:hi] == i, seg[:, :, lo:hi] == i
            else:
                a, b = gt == i, seg == i
            d = compute_dice_coefficient(a, b)
        out.append(round(float(d), 4))
    return out


def evaluation_case(seed=31):
    """Synthetic label volumes behind tests/golden/evaluation.npz: blobs of 13 organs, some missing / spurious."""
    rng = np.random.default_rng(seed)
    gt = np.zeros((24, 20, 12), dtype=np.uint8)
    for lab in range(1, 14):
        if lab in (8, 12):                     # organ absent from the ground truth
            continue
        c = rng.integers(3, (21, 17, 9))
        r = rng.integers(2, 5)
        gt[c[0] - r:c[0] + r, c[1] - r:c[1] + r, max(c[2] - r, 0):c[2] + r] = lab
    seg = gt.copy()
    flip = rng.random(gt.shape) < 0.15
    seg[flip] = rng.integers(0, 14, size=int(flip.sum()))
    seg[seg == 12] = 0                         # label 12 absent from both; label 8 only (spuriously) in seg
    return gt, seg
