"""GPU: plugin boundary #1 with the REAL product network behind the reference's trainer loop (fake base class:
tests/fake_nnunet.py restates nnUNetTrainer.py:64-215, 833-863)."""
import copy

import numpy as np
import pytest
import torch

import fake_nnunet as FK
from _trajectory import assert_same_trajectory

pytestmark = pytest.mark.gpu
IMG, NCLS, BATCH = (64, 64), 14, 2


def _trainer(variant="B"):
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import nnunet_plugin
    from oracle import mlagg_oracle as O
    cls = nnunet_plugin.make_trainer_class(FK.nnUNetTrainer, variant=variant)
    tr = cls(FK.make_plans(IMG, BATCH), "2d_bs10", 0, FK.make_dataset_json(NCLS), device=torch.device("cuda"))
    tr.initialize()
    O.deterministic_fill_(tr.network.state_dict())
    return tr


def _batches(n):
    from mlagg_unet_amd import trainer
    out = []
    for it in range(n):
        data, target = trainer.synthetic_batch(BATCH, 1, *IMG, NCLS, seed=700 + it)
        out.append({"data": data, "target": target})
    return out


def test_plugin_train_step_equals_trainer_train_step():
    from mlagg_unet_amd import model, trainer
    tr = _trainer()
    assert isinstance(tr.network, model.MLLA_Uper) and tr.grad_scaler is None
    assert isinstance(tr.optimizer, trainer.ClipAdamW)
    twin = copy.deepcopy(tr.network)
    twin_opt, _ = trainer.configure_optimizers(twin, tr.initial_lr, tr.weight_decay)
    for b in _batches(2):
        torch.manual_seed(5)                                   # DropPath masks
        got = tr.train_step(b)
        torch.manual_seed(5)
        want = trainer.train_step(twin, twin_opt, b["data"].cuda(), [t.cuda() for t in b["target"]], batch_dice=True)
        assert isinstance(got["loss"], np.ndarray) and abs(float(got["loss"]) - float(want)) < 2e-5
    assert tr.base_calls["train_step"] == 0 and tr.base_calls["_build_loss"] == 0
    # the losses above agree to rounding (the first one bit for bit); the parameters: see _assert_same_trajectory
    assert_same_trajectory(tr.network, twin, steps=2, lr=tr.initial_lr)


def test_plugin_and_trainer_steps_are_bit_identical_in_deterministic_mode():
    """The strict form of the test above: with ``trainer.set_deterministic`` two independent runs of the
    same two steps -- the plugin's ``train_step`` behind the reference loop and ``trainer.train_step`` on a twin -- end on
    bit-identical losses and parameters.  (The float-atomic kernels that used to differ were this package's own K9 statistics,
    K3 / K4 parameter sums and clip norm, all replaced by fixed-order reductions, plus MIOpen's atomic solvers, which the
    deterministic attribute excludes: profiles/round3_nondeterminism_*.log.)"""
    from mlagg_unet_amd import trainer
    trainer.set_deterministic(True)
    try:
        tr = _trainer()
        twin = copy.deepcopy(tr.network)
        twin_opt, _ = trainer.configure_optimizers(twin, tr.initial_lr, tr.weight_decay)
        for b in _batches(2):
            torch.manual_seed(5)
            got = tr.train_step(b)
            torch.manual_seed(5)
            want = trainer.train_step(twin, twin_opt, b["data"].cuda(), [t.cuda() for t in b["target"]], batch_dice=True)
            assert float(got["loss"]) == float(want)
        for (k, a), b in zip(tr.network.state_dict().items(), twin.state_dict().values()):
            assert torch.equal(a, b), k
    finally:
        trainer.set_deterministic(False)


def test_plugin_replays_the_step_as_a_hipgraph_after_three_eager_steps():
    """The plugin's train_step behind the reference loop (per-step loss read-back, B:863): three eager steps, then the whole step is
    captured and replayed (trainer.GraphedTrainStep, warm-up 0: no batch is trained on twice).  Deterministic mode, DropPath off
    (eval) so that both sides are comparable bit for bit: seven steps on seven different batches end on the losses and parameters of
    seven eager ``trainer.train_step`` calls on a twin; a batch of another geometry falls back to an eager step."""
    from mlagg_unet_amd import nnunet_plugin, trainer
    assert nnunet_plugin.PLUGIN_GRAPH
    trainer.set_deterministic(True)
    try:
        tr = _trainer()
        tr.network.eval()
        assert tr.optimizer.capturable
        twin = copy.deepcopy(tr.network)
        twin_opt, _ = trainer.configure_optimizers(twin, tr.initial_lr, tr.weight_decay)
        for it, b in enumerate(_batches(7)):
            got = tr.train_step(b)
            want = trainer.train_step(twin, twin_opt, b["data"].cuda(), [t.cuda() for t in b["target"]], batch_dice=True)
            assert float(got["loss"]) == float(want), it
            assert (tr._graphed is not None) == (it >= 3), it
        for (k, a), q in zip(tr.network.state_dict().items(), twin.state_dict().values()):
            assert torch.equal(a, q), k
        assert tr.optimizer.steps_done() == 7 and tr._graph_failed is None
        # another batch geometry: an eager step, the captured graph stays valid for the old one
        data, target = trainer.synthetic_batch(1, 1, *IMG, NCLS, seed=99)
        got = tr.train_step({"data": data, "target": target})
        want = trainer.train_step(twin, twin_opt, data.cuda(), [t.cuda() for t in target], batch_dice=True)
        assert float(got["loss"]) == float(want)
        b = _batches(1)[0]
        got = tr.train_step(b)
        want = trainer.train_step(twin, twin_opt, b["data"].cuda(), [t.cuda() for t in b["target"]], batch_dice=True)
        assert float(got["loss"]) == float(want) and tr.optimizer.steps_done() == 9
    finally:
        trainer.set_deterministic(False)


def test_reference_amp_step_also_runs_on_the_product_network():
    """The inherited body of B:833-863 (autocast('cuda') + GradScaler) is not what the plugin runs, but a maintainer who
    keeps it must not crash: the network leaves autocast for its own precision, gradients come back fp32, and the scaled
    step lands on the same parameters (the GradScaler's power-of-two scale cancels)."""
    from mlagg_unet_amd import trainer
    tr = _trainer()
    twin = copy.deepcopy(tr.network)
    twin_opt = torch.optim.AdamW(twin.parameters(), tr.initial_lr, weight_decay=tr.weight_decay, eps=1e-4)
    tr.optimizer = torch.optim.AdamW(tr.network.parameters(), tr.initial_lr, weight_decay=tr.weight_decay, eps=1e-4)
    tr.grad_scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    for b in _batches(2):
        torch.manual_seed(9)
        got = FK.nnUNetTrainer.train_step(tr, b)               # the reference's AMP step, verbatim logic
        torch.manual_seed(9)
        want = trainer.train_step(twin, twin_opt, b["data"].cuda(), [t.cuda() for t in b["target"]], batch_dice=True)
        assert np.isfinite(got["loss"]) and abs(float(got["loss"]) - float(want)) < 1e-5
    assert tr.base_calls["train_step"] == 2
    assert_same_trajectory(tr.network, twin, steps=2, lr=tr.initial_lr)
