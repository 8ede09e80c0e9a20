"""Shared check of the GPU tests that run the same optimisation steps twice."""


def assert_same_trajectory(net_a, net_b, steps, lr):
    """Two runs of the same steps: the losses agree to rounding (checked by the callers); the parameters agree up to the
    run-to-run wobble of float-atomic weight gradients (MIOpen's split-K weight-gradient kernels) seen through AdamW's
    m / sqrt(v) normalisation: a gradient that is ~0 relative to that noise can move its weight by up to lr per step in
    either direction.  So: no element further apart than the AdamW bound, and all but a sliver within 5 % of one step."""
    worst, n_far, n_all = 0.0, 0, 0
    for (k, a), b in zip(net_a.state_dict().items(), net_b.state_dict().values()):
        d = (a.float() - b.float()).abs()
        worst = max(worst, float(d.max()))
        n_far += int((d > 0.05 * lr).sum())
        n_all += d.numel()
    assert worst <= 2.0 * lr * steps * 1.01, worst
    assert n_far <= 1e-4 * n_all, (n_far, n_all, worst)
