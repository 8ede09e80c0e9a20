"""CPU: plugin boundary #1 behind the reference's trainer loop.  The class made by ``nnunet_plugin.make_trainer_class`` is
built on ``tests/fake_nnunet.nnUNetTrainer`` (a restatement of the reference base class's ``__init__`` / ``initialize`` /
AMP ``train_step``, nnUNetTrainer.py:64-215, 833-863) and driven the way ``run_training`` drives it (B:1202-1223):
construct, ``initialize``, ``train_step(batch)``.  The product network cannot run on the host (its ops refuse CPU
tensors), so ``model.build_network_architecture`` is replaced by a small five-head network here; the GPU twin of this
test (test_plugin_gpu.py) runs the real one."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import fake_nnunet as FK


class StubNet(nn.Module):
    def __init__(self, in_ch, n_cls, ds=True):
        super().__init__()
        self.deep_supervision = ds
        self.body = nn.Conv2d(in_ch, 8, 3, padding=1)
        self.heads = nn.ModuleList([nn.Conv2d(8, n_cls, 1) for _ in range(5)])
        self.dummy_tensor = nn.Parameter(torch.tensor([1.0]), requires_grad=False)

    def forward(self, x):
        f = F.relu(self.body(x))
        outs = [h(F.avg_pool2d(f, 2 ** s) if s else f) for s, h in enumerate(self.heads)]
        return outs if self.deep_supervision else outs[0]


@pytest.fixture
def plugin(monkeypatch):
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import model, nnunet_plugin
    seen = {}

    def build(patch_size, in_ch, n_cls, ds=True, variant="B", precision="fp32"):
        seen["args"] = (tuple(patch_size), in_ch, n_cls, ds, variant)
        seen["precision"] = precision
        torch.manual_seed(0)
        return StubNet(in_ch, n_cls, ds)

    monkeypatch.setattr(model, "build_network_architecture", build)
    monkeypatch.setenv("MLAGG_MIOPEN_TUNED", "0")
    return nnunet_plugin.make_trainer_class(FK.nnUNetTrainer, variant="B"), seen


def _batch(seed, n=3, n_cls=5, size=32):
    g = torch.Generator().manual_seed(seed)
    return {"data": torch.rand(n, 1, size, size, generator=g),
            "target": [torch.round(torch.rand(n, 1, size >> s, size >> s, generator=g) * (n_cls - 1)) for s in range(5)]}


def test_plugin_runs_its_own_step_behind_the_reference_loop(plugin):
    cls, seen = plugin
    from mlagg_unet_amd import trainer
    assert cls.__name__ == "nnUNetTrainer_MLAgg_2D_dt_MS" and issubclass(cls, FK.nnUNetTrainer)
    for name in ("train_step", "initialize", "_build_loss", "configure_optimizers", "build_network_architecture",
                 "set_deep_supervision_enabled", "_get_deep_supervision_scales", "validation_step"):
        assert getattr(cls, name) is not getattr(FK.nnUNetTrainer, name, None), f"{name} is inherited"
    tr = cls(FK.make_plans((32, 32), 3), "2d_bs10", 0, FK.make_dataset_json(5), device=torch.device("cpu"))
    assert (tr.initial_lr, tr.weight_decay, tr.num_epochs, tr.num_iterations_per_epoch) == (5e-4, 3e-5, 500, 250)
    assert tr.grad_scaler is None
    tr.initialize()
    assert seen["args"] == ((32, 32), 1, 5, True, "B")
    assert tr.base_calls["initialize"] == 1 and tr.base_calls["_build_loss"] == 0 and tr.base_calls["plain_ddp_wrap"] == 0
    assert callable(tr.loss) and tr._get_deep_supervision_scales() == [[1 / 2 ** i] * 2 for i in range(5)]
    # the reference step, written out with torch pieces on a copy of the network (B:843-861, fp32 branch)
    ref_net = copy.deepcopy(tr.network)
    ref_opt = torch.optim.AdamW(ref_net.parameters(), 5e-4, weight_decay=3e-5, eps=1e-4)
    for it in range(2):
        b = _batch(10 + it)
        out = tr.train_step(b)
        assert set(out) == {"loss"} and isinstance(out["loss"], np.ndarray)
        ref_opt.zero_grad()
        loss = trainer.deep_supervision_loss_eager(ref_net(b["data"]), b["target"], True, False)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref_net.parameters(), 12)
        ref_opt.step()
        assert abs(float(out["loss"]) - float(loss)) < 1e-6
    assert tr.base_calls["train_step"] == 0                      # the AMP body of B:833-863 never ran
    for (k, a), b in zip(tr.network.state_dict().items(), ref_net.state_dict().values()):
        assert torch.allclose(a, b, atol=1e-7), k
    # optimizer checkpoints keep the reference's parameter indexing: every parameter incl. the frozen dummy_tensor
    assert len(tr.optimizer.param_groups[0]["params"]) == len(list(tr.network.parameters()))
    tr.set_deep_supervision_enabled(False)
    assert tr.network.deep_supervision is False


def test_plugin_drops_the_grad_scaler_of_a_gpu_trainer(plugin):
    """On a GPU device the reference constructor makes a GradScaler (B:152) and its train_step autocasts (B:848); the
    plugin must leave neither in place (constructing needs no GPU)."""
    cls, _ = plugin
    tr = cls(FK.make_plans((32, 32), 3), "2d_bs10", 0, FK.make_dataset_json(5), device=torch.device("cuda"))
    assert tr.device == torch.device("cuda", 0) and tr.grad_scaler is None


def test_ignore_label_dataset_uses_the_product_loss_and_regions_keep_the_reference_classes(plugin):
    """T:106-116: a dataset with an ignore label gets the product loss with that label masked out (value checked against the
    oracle's restatement of DC_and_CE_loss(ignore_label=...)); region targets fall back to the maintainer's DC_and_BCE_loss."""
    from oracle import mlagg_oracle as O
    cls, _ = plugin
    dj = FK.make_dataset_json(5)
    dj["ignore_label"] = 5
    tr = cls(FK.make_plans((32, 32), 3), "2d_bs10", 0, dj, device=torch.device("cpu"))
    tr.initialize()
    assert tr.base_calls["_build_loss"] == 0 and callable(tr.loss)
    g = torch.Generator().manual_seed(4)
    outs = [torch.randn(2, 5, 32 >> s, 32 >> s, generator=g) for s in range(5)]
    tg = [torch.round(torch.rand(2, 1, 32 >> s, 32 >> s, generator=g) * 5) for s in range(5)]        # label 5 = ignore
    want = O.deep_supervision_loss(outs, tg, batch_dice=tr.configuration_manager.batch_dice, ignore_label=5)
    assert abs(float(tr.loss(outs, tg)) - float(want)) < 1e-6
    plain = O.deep_supervision_loss(outs, [t.clamp(max=4) for t in tg], batch_dice=tr.configuration_manager.batch_dice)
    assert abs(float(want) - float(plain)) > 1e-3                   # ... and the mask matters
    dj2 = FK.make_dataset_json(5)
    dj2["regions"] = True
    tr2 = cls(FK.make_plans((32, 32), 3), "2d_bs10", 0, dj2, device=torch.device("cpu"))
    tr2.initialize()
    assert tr2.base_calls["_build_loss"] == 1 and tr2.loss == "reference-loss-classes"


def test_unknown_precision_is_refused():
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import nnunet_plugin
    with pytest.raises(RuntimeError):
        nnunet_plugin.make_trainer_class(FK.nnUNetTrainer, precision="int8")


def _reference_hard_counts(logits, target, ignore_label):
    """B:899-940 written out: one-hot of the argmax, the ignore mask applied to prediction and target, sums over batch + space."""
    C = logits.shape[1]
    seg = logits.argmax(1)[:, None]
    pred = torch.zeros(logits.shape, dtype=torch.float32).scatter_(1, seg, 1)
    tgt = target.clone()
    if ignore_label is not None:
        mask = (tgt != ignore_label).float()
        tgt[tgt == ignore_label] = 0
    else:
        mask = torch.ones_like(tgt)
    onehot = torch.zeros(logits.shape, dtype=torch.float32).scatter_(1, tgt.long(), 1)
    tp = (pred * onehot * mask).sum((0, 2, 3))
    fp = (pred * (1 - onehot) * mask).sum((0, 2, 3))
    fn = ((1 - pred) * onehot * mask).sum((0, 2, 3))
    return tp[1:], fp[1:], fn[1:]


def test_validation_step_masks_the_ignore_label(plugin):
    """B:880-942 on a partially annotated dataset: the ignore label (== number of classes) must neither crash the confusion
    matrix nor count in tp / fp / fn, and the validation loss is the trainer's own (masked) loss."""
    from oracle import mlagg_oracle as O
    cls, _ = plugin
    dj = FK.make_dataset_json(5)
    dj["ignore_label"] = 5
    tr = cls(FK.make_plans((32, 32), 3), "2d_bs10", 0, dj, device=torch.device("cpu"))
    tr.initialize()
    g = torch.Generator().manual_seed(9)
    batch = {"data": torch.rand(3, 1, 32, 32, generator=g),
             "target": [torch.round(torch.rand(3, 1, 32 >> s, 32 >> s, generator=g) * 5) for s in range(5)]}
    assert (batch["target"][0] == 5).any()
    out = tr.validation_step(batch)
    with torch.no_grad():
        logits = tr.network(batch["data"])
    tp, fp, fn = _reference_hard_counts(logits[0], batch["target"][0], 5)
    assert torch.equal(out["tp_hard"].float(), tp) and torch.equal(out["fp_hard"].float(), fp)
    assert torch.equal(out["fn_hard"].float(), fn)
    want = O.deep_supervision_loss(logits, batch["target"], batch_dice=tr.configuration_manager.batch_dice, ignore_label=5)
    assert abs(float(out["loss"]) - float(want)) < 1e-6
    # without an ignore label the counts are the plain ones
    tr2 = cls(FK.make_plans((32, 32), 3), "2d_bs10", 0, FK.make_dataset_json(5), device=torch.device("cpu"))
    tr2.initialize()
    b2 = _batch(3)
    out2 = tr2.validation_step(b2)
    with torch.no_grad():
        tp2, fp2, fn2 = _reference_hard_counts(tr2.network(b2["data"])[0], b2["target"][0], None)
    assert torch.equal(out2["tp_hard"].float(), tp2) and torch.equal(out2["fn_hard"].float(), fn2)
    assert torch.equal(out2["fp_hard"].float(), fp2)


def test_tuned_convolution_database_is_used_only_for_the_shapes_it_holds(monkeypatch):
    """The committed MIOpen find-db covers the fp32 256 x 256 step: any other patch size / precision must not switch to FAST
    find mode with the naive fallback solvers off."""
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import miopen_tuning, model, nnunet_plugin
    calls = []
    monkeypatch.setattr(miopen_tuning, "use_tuned_convolutions", lambda enabled=True: calls.append(enabled))
    monkeypatch.setattr(model, "build_network_architecture", lambda *a, **k: StubNet(1, 5))
    for precision, patch, want in (("fp32", (256, 256), True), ("fp32", (224, 224), False), ("bf16", (256, 256), False),
                                   ("fp16", (512, 640), False)):
        cls = nnunet_plugin.make_trainer_class(FK.nnUNetTrainer, precision=precision)
        cls(FK.make_plans(patch, 3), "2d_bs10", 0, FK.make_dataset_json(5), device=torch.device("cpu"))
        assert calls[-1] is want, (precision, patch)


def test_3d_trainer_plugin_reads_the_plan_and_keeps_the_base_recipe(monkeypatch):
    """nnUNetTrainerUMambaEnc_SS3D (variants/mamba/nnUNetTrainerUMambaEnc_SS3D.py:8-31): the class overrides the network factory and
    inherits SGD + PolyLR + the deep-supervision scales of the base trainer; the factory hands the plan's architecture entries to
    model3d as get_umamba_enc_3d_from_plans (UMambaEnc_SS3D.py:890-942) does."""
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import miopen_tuning, model3d, nnunet_plugin
    seen = {}

    def build(in_ch, n_cls, kernels, strides, n_enc, n_dec, base, cap, ds):
        seen["args"] = (in_ch, n_cls, kernels, strides, n_enc, n_dec, base, cap, ds)
        return StubNet(in_ch, n_cls, ds)

    monkeypatch.setattr(model3d, "build_network_architecture_3d", build)
    monkeypatch.setattr(miopen_tuning, "use_tuned_convolutions", lambda enabled=True: seen.setdefault("miopen", enabled))
    cls = nnunet_plugin.make_umamba_enc_ss3d_trainer_class(FK.nnUNetTrainer)
    assert cls.__name__ == "nnUNetTrainerUMambaEnc_SS3D" and issubclass(cls, FK.nnUNetTrainer)
    strides = [[1, 1, 1], [2, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2], [1, 2, 2]]
    tr = cls(FK.make_plans_3d((96, 160, 160), 2, strides), "3d_fullres", 0, FK.make_dataset_json(14), device=torch.device("cpu"))
    assert tr.grad_scaler is None and seen["miopen"] is False
    tr.initialize()
    assert seen["args"] == (1, 14, [[3, 3, 3]] * 6, strides, [2] * 6, [2] * 5, 32, 320, True)
    assert isinstance(tr.optimizer, torch.optim.SGD) and tr.optimizer.defaults["momentum"] == 0.99 and tr.optimizer.defaults["nesterov"]
    assert tr.initial_lr == 1e-2 and tr.base_calls["_build_loss"] == 0 and callable(tr.loss)
    sc = tr._get_deep_supervision_scales()
    assert len(sc) == 5 and sc[-1] == [1 / 16, 1 / 16, 1 / 16]
    # 2-D plans are refused by the factory
    tr2 = cls(FK.make_plans((32, 32), 2), "2d_bs10", 0, FK.make_dataset_json(5), device=torch.device("cpu"))
    tr2.configuration_manager.conv_kernel_sizes = [[3, 3]] * 4
    with pytest.raises(RuntimeError):
        tr2.initialize()
