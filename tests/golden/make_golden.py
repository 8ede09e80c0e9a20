"""Generate the golden fixtures in this directory from the REFERENCE's own classes.

Run in the build container only (needs /root/reference; the GPU box never runs this):

    python tests/golden/make_golden.py

The reference's model code (T = nnUNetTrainer_MLAgg_2D_dt_MS.py, M = MambaSkip.py) is
imported unmodified with stand-ins registered in ``sys.modules`` for the third-party
packages that are absent offline: timm (DropPath, ...),
MONAI (Unetr blocks), mamba-ssm (``selective_scan_fn``), flash-attn (``flash_attn_func``),
dynamic_network_architectures and three nnunetv2 modules T only needs as names.  The
stand-ins for the *arithmetic* of those packages are the oracle's restatements, so the
fixtures pin everything the reference itself defines and leave those four third-party
boundaries unpinned (stated in every consumer of these files).

Weights are not stored: both sides call ``oracle.mlagg_oracle.deterministic_fill_`` on the
ABI-named state_dict.  Stored: inputs' seeds, outputs, loss values, gradient summaries.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/mlagg"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import mlagg_oracle as O  # noqa: E402

FLASH_SCALE = {"value": None}  # None -> flash default (variant A); 1.0 -> variant B


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def register_standins():
    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    _mod("timm")
    _mod("timm.optim")
    _mod("timm.scheduler")
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=O.DropPath, to_2tuple=to_2tuple, trunc_normal_=nn.init.trunc_normal_)
    sys.modules["timm"].optim = sys.modules["timm.optim"]
    sys.modules["timm"].scheduler = sys.modules["timm.scheduler"]

    class UnetrBasicBlock(O.UnetrBasicBlock):
        def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, res_block):
            assert spatial_dims == 2 and kernel_size == 3 and stride == 1 and norm_name == "instance" and res_block
            super().__init__(in_channels, out_channels)

    class UnetrUpBlock(O.UnetrUpBlock):
        def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, upsample_kernel_size,
                     norm_name, res_block):
            assert spatial_dims == 2 and kernel_size == 3 and upsample_kernel_size == 2 and res_block
            super().__init__(in_channels, out_channels)

    _mod("monai")
    _mod("monai.networks")
    _mod("monai.networks.blocks", UnetOutBlock=object, UnetrBasicBlock=UnetrBasicBlock, UnetrUpBlock=UnetrUpBlock)
    _mod("monai.networks.layers")
    _mod("monai.networks.layers.utils", get_norm_layer=None)
    _mod("monai.networks.blocks.dynunet_block", get_conv_layer=None)

    _mod("mamba_ssm")
    _mod("mamba_ssm.ops")
    _mod("mamba_ssm.ops.selective_scan_interface",
         selective_scan_fn=O.selective_scan_oracle, selective_scan_ref=O.selective_scan_oracle)

    def flash_attn_func(q, k, v, causal=False):
        assert not causal
        return O.softmax_attention_oracle(q, k, v, softmax_scale=FLASH_SCALE["value"])

    _mod("flash_attn", flash_attn_func=flash_attn_func)

    _mod("dynamic_network_architectures")
    _mod("dynamic_network_architectures.initialization")
    _mod("dynamic_network_architectures.initialization.weight_init",
         init_last_bn_before_add_to_0=None, InitWeights_He=None)
    _mod("nnunetv2.training.nnUNetTrainer.nnUNetTrainer", nnUNetTrainer=type("nnUNetTrainer", (), {}))
    _mod("nnunetv2.training.nnUNetTrainer.variants.network_architecture.nnUNetTrainerNoDeepSupervision",
         nnUNetTrainerNoDeepSupervision=type("nnUNetTrainerNoDeepSupervision", (), {}))
    _mod("nnunetv2.utilities.plans_handling.plans_handler", ConfigurationManager=object, PlansManager=object)


def import_reference():
    register_standins()
    M = importlib.import_module("nnunetv2.training.nnUNetTrainer.variants.mamba.MambaSkip")
    T = importlib.import_module("nnunetv2.training.nnUNetTrainer.nnUNetTrainer_MLAgg_2D_dt_MS")
    return T, M


def ref_model(T, img, n_cls=14, in_ch=1):
    m = T.MLLA_Uper(img_size=list(img), patch_size=2, in_channels=in_ch, out_channels=n_cls, embed_dim=96,
                    depths=[2, 2, 2, 2], num_heads=[2, 4, 8, 16], mlp_ratio=2, qkv_bias=True, drop_rate=0.,
                    dropout_path_rate=0.1, sr_ratio=[16, 8, 4, 2], norm_layer=nn.LayerNorm, ape=False,
                    use_checkpoint=False, deep_supervision=True)
    O.deterministic_fill_(m.state_dict())
    return m


def grad_summary(model):
    names, norms = [], []
    for n, p in sorted(model.named_parameters()):
        if p.grad is not None:
            names.append(n)
            norms.append(float(p.grad.double().norm()))
    return names, np.asarray(norms, dtype=np.float64)


def golden_full_model(T, variant):
    FLASH_SCALE["value"] = None if variant == "A" else 1.0
    img = (64, 64)
    m = ref_model(T, img).eval()
    data, target = O.synthetic_batch(1, 1, *img, 14, seed=1234)
    out = m(data)
    from nnunetv2.training.loss.compound_losses import DC_and_CE_loss
    from nnunetv2.training.loss.deep_supervision import DeepSupervisionWrapper
    from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss
    base = DC_and_CE_loss({'batch_dice': True, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                          weight_dice=1, ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss)
    w = np.array([1 / (2 ** i) for i in range(5)])
    loss = DeepSupervisionWrapper(base, w / w.sum())(out, target)
    loss.backward()
    names, norms = grad_summary(m)
    small = {}
    for n, p in m.named_parameters():
        if p.grad is not None and (("lambda_" in n) or n.endswith(("dt_projs_bias", "Ds", "x_proj_weight",
                                                                    "dt_projs_weight", "subln.weight"))):
            small["grad/" + n] = p.grad.numpy().copy()
    small["grad/A_logs"] = m.mambaskip.blocks[0].self_attention.A_logs.grad.numpy().copy()
    np.savez_compressed(
        os.path.join(HERE, f"full_model_64_variant{variant}.npz"),
        img=np.asarray(img), batch=1, n_cls=14, data_seed=1234, weight_seed=0,
        loss=float(loss), grad_names=np.asarray(names), grad_norms=norms,
        **{f"out{i}": o.detach().numpy() for i, o in enumerate(out)}, **small)
    print("full model", variant, float(loss), [tuple(o.shape) for o in out])


def golden_full_model_256(T):
    """The reference network itself at the HEADLINE size of BASELINE config 2 (1 x 1 x 256 x 256, 14 classes, variant B):
    logits of the five heads (full resolution sub-sampled 4x4 to keep the fixture small, plus per-class sums of the full
    maps), the deep-supervision loss and every parameter-gradient norm.  ~2 minutes of CPU time (Python-loop scan over
    L_cat = 21760 steps)."""
    FLASH_SCALE["value"] = 1.0
    img = (256, 256)
    m = ref_model(T, img).eval()
    data, target = O.synthetic_batch(1, 1, *img, 14, seed=4321)
    out = m(data)
    from nnunetv2.training.loss.compound_losses import DC_and_CE_loss
    from nnunetv2.training.loss.deep_supervision import DeepSupervisionWrapper
    from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss
    base = DC_and_CE_loss({'batch_dice': True, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                          weight_dice=1, ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss)
    w = np.array([1 / (2 ** i) for i in range(5)])
    loss = DeepSupervisionWrapper(base, w / w.sum())(out, target)
    loss.backward()
    names, norms = grad_summary(m)
    np.savez_compressed(
        os.path.join(HERE, "full_model_256_variantB.npz"), img=np.asarray(img), data_seed=4321, loss=float(loss),
        grad_names=np.asarray(names), grad_norms=norms,
        out0_sub=out[0].detach().numpy()[:, :, ::4, ::4], out0_class_sums=out[0].detach().double().sum((0, 2, 3)).numpy(),
        out0_abs_max=float(out[0].detach().abs().max()),
        **{f"out{i}": o.detach().numpy() for i, o in enumerate(out) if i > 0})
    print("full model 256", float(loss), [tuple(o.shape) for o in out])


CONFIG_GOLDENS = {
    # tag: (img, in_ch, n_cls, batch, variant, data_seed, autocast dtype or None)
    # BASELINE configs[2] shape (ACDC-like: 224 x 224, 4 classes), fp32 and under the reference's autocast (B:848) in bf16
    "224_variantB": ((224, 224), 1, 4, 1, "B", 2240, None),
    "224_variantB_bf16": ((224, 224), 1, 4, 1, "B", 2240, torch.bfloat16),
    # the headline size with TWO samples: batch-dice across samples and every kernel's batch indexing at 256 x 256
    "256_b2_variantB": ((256, 256), 1, 14, 2, "B", 2562, None),
    # BASELINE configs[4] shape (Endovis17-like: 512 x 640 RGB, 8 classes), shipped flash scaling (variant A)
    "512x640_variantA": ((512, 640), 3, 8, 1, "A", 5126, None),
    "512x640_variantA_fp16": ((512, 640), 3, 8, 1, "A", 5126, torch.float16),
}


def _sub_stride(h, w, budget=8192):
    s = 1
    while ((h + s - 1) // s) * ((w + s - 1) // s) > budget:
        s += 1
    return s


def golden_full_model_config(T, tag):
    """The reference network + DeepSupervisionWrapper(DC_and_CE_loss) at one BASELINE shape: sub-sampled logits of the
    five heads (stride chosen so a map keeps <= 8192 pixels), per-(sample, class) sums and |max| of the FULL maps, the loss
    and every parameter-gradient norm.  With ``autocast`` the forward and the loss run under torch.autocast("cpu", dtype),
    as nnUNetTrainer.train_step does on the device (B:848-851); there is no GradScaler on this side (its scale cancels)."""
    img, in_ch, n_cls, batch, variant, seed, amp = CONFIG_GOLDENS[tag]
    FLASH_SCALE["value"] = None if variant == "A" else 1.0
    m = ref_model(T, img, n_cls, in_ch).eval()
    data, target = O.synthetic_batch(batch, in_ch, *img, n_cls, seed=seed)
    from nnunetv2.training.loss.compound_losses import DC_and_CE_loss
    from nnunetv2.training.loss.deep_supervision import DeepSupervisionWrapper
    from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss
    base = DC_and_CE_loss({'batch_dice': True, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                          weight_dice=1, ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss)
    w = np.array([1 / (2 ** i) for i in range(5)])
    import contextlib
    import time
    t0 = time.time()
    ctx = torch.autocast("cpu", dtype=amp) if amp is not None else contextlib.nullcontext()
    with ctx:
        out = m(data)
        loss = DeepSupervisionWrapper(base, w / w.sum())(out, target)
    loss.backward()
    names, norms = grad_summary(m)
    store = {}
    for i, o in enumerate(out):
        o = o.detach().float()
        s = _sub_stride(o.shape[2], o.shape[3])
        store[f"out{i}_sub"] = o.numpy()[:, :, ::s, ::s]
        store[f"out{i}_stride"] = s
        store[f"out{i}_sums"] = o.double().sum((2, 3)).numpy()
        store[f"out{i}_abs_max"] = float(o.abs().max())
    np.savez_compressed(
        os.path.join(HERE, f"full_model_{tag}.npz"), img=np.asarray(img), in_ch=in_ch, n_cls=n_cls, batch=batch,
        variant=variant, data_seed=seed, autocast="" if amp is None else str(amp).replace("torch.", ""),
        out_dtype=str(out[0].dtype).replace("torch.", ""), loss=float(loss),
        grad_names=np.asarray(names), grad_norms=norms, **store)
    print("full model", tag, float(loss), [tuple(o.shape) for o in out], out[0].dtype, f"{time.time() - t0:.0f}s", flush=True)


GRAD_TENSOR_PARAMS = (
    # >= 20 parameters spread over the encoder, the downs, the MSMM skip module and the decoder: the
    # gradient TENSORS, not only their norms -- a wrong direction with the right norm must not pass
    "mlla.patch_embed.proj1.conv1.weight", "mlla.patch_embed.proj2.conv2.weight", "mlla.patch_embed.proj1.norm1.weight",
    "mlla.layers.0.blocks.0.in_proj.weight", "mlla.layers.0.blocks.0.dwc.weight", "mlla.layers.0.blocks.1.attn.0.kv.weight",
    "mlla.layers.0.blocks.1.attn.1.lepe.weight", "mlla.layers.0.blocks.0.mlp.fc1.weight", "mlla.layers.1.blocks.0.attn.1.q.weight",
    "mlla.layers.1.blocks.1.attn.1.sr.weight", "mlla.layers.1.blocks.1.norm2.weight", "mlla.layers.2.blocks.0.out_proj.weight",
    "mlla.layers.2.blocks.1.attn.0.lambda_q1", "mlla.layers.2.blocks.1.attn.1.subln.weight", "mlla.layers.3.blocks.0.act_proj.weight",
    "mlla.layers.3.blocks.1.mlp.fc2.bias", "mlla.layers.3.blocks.1.attn.1.kv.weight", "mlla.downs.0.conv1.weight",
    "mlla.downs.2.res_conv.weight", "mambaskip.blocks.0.ln_1.weight", "mambaskip.blocks.0.self_attention.in_proj.weight",
    "mambaskip.blocks.0.self_attention.conv2d.0.weight", "mambaskip.blocks.0.self_attention.x_proj_weight",
    "mambaskip.blocks.0.self_attention.dt_projs_weight", "mambaskip.blocks.0.self_attention.dt_projs_bias",
    "mambaskip.blocks.0.self_attention.A_logs", "mambaskip.blocks.0.self_attention.Ds", "mambaskip.blocks.0.self_attention.out_norm.weight",
    "mambaskip.blocks.0.self_attention.out_proj.weight", "mambaskip.blocks.0.mlps.1.fc1.weight", "mambaskip.blocks.0.mlps.3.dwconv.dwconv.weight",
    "mambaskip.blocks.0.conv_branches.0.0.weight", "mambaskip.blocks.0.conv_branches.2.1.weight", "up_2.conv1.weight", "up_0.res_conv.weight",
    "dec_block_2.0.conv2.weight", "dec_block_0.1.conv1.weight", "dec_block_1.0.norm.weight", "encoder0.layer.conv1.conv.weight",
    "encoder0.layer.conv3.conv.weight", "decoder0.transp_conv.conv.weight", "decoder0.conv_block.conv2.conv.weight",
    "out_0.conv_out.weight", "out_3.conv_out.bias",
)
GRAD_TENSOR_CASES = {
    # tag: (img, in_ch, n_cls, batch, variant, data_seed) -- the inputs of the forward goldens of the same shapes
    "256_variantB": ((256, 256), 1, 14, 1, "B", 4321),
    "224_variantB": ((224, 224), 1, 4, 1, "B", 2240),
    "512x640_variantA": ((512, 640), 3, 8, 1, "A", 5126),
}


def _subsample(t, budget=2048):
    flat = t.detach().reshape(-1)
    stride = max(1, -(-flat.numel() // budget))
    return flat[::stride].numpy().copy(), stride


def golden_gradient_tensors(T, tag):
    """Gradient TENSORS of GRAD_TENSOR_PARAMS (sub-sampled to <= 2048 entries each, stride stored) of the reference network +
    DeepSupervisionWrapper(DC_and_CE_loss) at one BASELINE shape, eval mode, same inputs as full_model_<tag>.npz."""
    img, in_ch, n_cls, batch, variant, seed = GRAD_TENSOR_CASES[tag]
    FLASH_SCALE["value"] = None if variant == "A" else 1.0
    m = ref_model(T, img, n_cls, in_ch).eval()
    data, target = O.synthetic_batch(batch, in_ch, *img, n_cls, seed=seed)
    from nnunetv2.training.loss.compound_losses import DC_and_CE_loss
    from nnunetv2.training.loss.deep_supervision import DeepSupervisionWrapper
    from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss
    base = DC_and_CE_loss({'batch_dice': True, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                          weight_dice=1, ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss)
    w = np.array([1 / (2 ** i) for i in range(5)])
    import time
    t0 = time.time()
    loss = DeepSupervisionWrapper(base, w / w.sum())(m(data), target)
    loss.backward()
    params = dict(m.named_parameters())
    store = {}
    for n in GRAD_TENSOR_PARAMS:
        g, stride = _subsample(params[n].grad)
        store["grad/" + n] = g
        store["stride/" + n] = stride
        store["absmax/" + n] = float(params[n].grad.abs().max())
    np.savez_compressed(os.path.join(HERE, f"full_model_{tag}_grads.npz"), img=np.asarray(img), in_ch=in_ch, n_cls=n_cls, batch=batch,
                        variant=variant, data_seed=seed, loss=float(loss), names=np.asarray(GRAD_TENSOR_PARAMS), **store)
    print("gradient tensors", tag, float(loss), len(GRAD_TENSOR_PARAMS), "parameters", f"{time.time() - t0:.0f}s", flush=True)


TRAIN_MODE_DROPS = ((3, 1), (12, 0), (14, 1), (17, 0))      # (DropPath call, sample): encoder stage 1 and 3 branches, MSMM scan, MSMM MLP of scale 2


def golden_train_mode(T):
    """The reference network in TRAIN mode (DropPath active: T:868, 903, 907; M:688, 741, 745; rates linspace(0, 0.1, 8) and 0.1) on
    two 64 x 64 samples.  The per-sample keep masks of the 19 DropPath calls (the first block's rate is 0) are PRESET instead of drawn
    -- with rates <= 0.1 a seeded draw drops one branch in 38; the preset drops four, one of them the MSMM scan branch and one an
    MSMM gated MLP -- applied exactly as timm applies a drawn mask (x * mask / keep), and stored already divided by the keep
    probability so that the other side can inject the same factors; logits, loss, every gradient norm and the GRAD_TENSOR_PARAMS
    gradient tensors."""
    FLASH_SCALE["value"] = 1.0
    img = (64, 64)
    m = ref_model(T, img).train()
    data, target = O.synthetic_batch(2, 1, *img, 14, seed=777)
    drawn = []
    orig = O.DropPath.forward

    def recording_forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_ones((x.shape[0],) + (1,) * (x.ndim - 1))
        for call, sample in TRAIN_MODE_DROPS:
            if call == len(drawn):
                mask[sample] = 0.0
        drawn.append((keep, (mask.reshape(-1) / keep).clone()))
        return x * mask / keep

    O.DropPath.forward = recording_forward
    try:
        torch.manual_seed(2024)
        out = m(data)
    finally:
        O.DropPath.forward = orig
    from nnunetv2.training.loss.compound_losses import DC_and_CE_loss
    from nnunetv2.training.loss.deep_supervision import DeepSupervisionWrapper
    from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss
    base = DC_and_CE_loss({'batch_dice': True, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                          weight_dice=1, ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss)
    w = np.array([1 / (2 ** i) for i in range(5)])
    loss = DeepSupervisionWrapper(base, w / w.sum())(out, target)
    loss.backward()
    names, norms = grad_summary(m)
    params = dict(m.named_parameters())
    store = {}
    for n in GRAD_TENSOR_PARAMS:
        g, stride = _subsample(params[n].grad)
        store["grad/" + n] = g
        store["stride/" + n] = stride
    masks = torch.stack([v for _, v in drawn]).numpy()
    assert masks.shape == (19, 2) and int((masks == 0).sum()) == len(TRAIN_MODE_DROPS), (masks.shape, masks)
    np.savez_compressed(os.path.join(HERE, "full_model_64_train_mode.npz"), img=np.asarray(img), batch=2, n_cls=14, data_seed=777,
                        keep=np.asarray([k for k, _ in drawn]), masks=masks, loss=float(loss), grad_names=np.asarray(names),
                        grad_norms=norms, names=np.asarray(GRAD_TENSOR_PARAMS),
                        **{f"out{i}": o.detach().numpy() for i, o in enumerate(out)}, **store)
    print("train mode", float(loss), masks.tolist())


def golden_mllablock(T, variant):
    FLASH_SCALE["value"] = None if variant == "A" else 1.0
    for tag, dim, res, heads, sr in (("s0", 96, (16, 16), 2, 4), ("s2", 384, (6, 8), 8, 2)):
        blk = T.MLLABlock(dim=dim, input_resolution=res, num_heads=heads, mlp_ratio=2, qkv_bias=True, drop=0.,
                          drop_path=0.0, sr_ratio=sr, norm_layer=nn.LayerNorm).eval()
        O.deterministic_fill_(blk.state_dict(), seed=7)
        g = torch.Generator().manual_seed(99)
        x = torch.randn(2, dim, *res, generator=g).requires_grad_(True)
        y = blk(x)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        names, norms = grad_summary(blk)
        np.savez_compressed(os.path.join(HERE, f"mllablock_{tag}_variant{variant}.npz"), dim=dim, res=np.asarray(res),
                            heads=heads, sr=sr, x=x.detach().numpy(), gy=gy.numpy(), y=y.detach().numpy(),
                            gx=x.grad.numpy(), grad_names=np.asarray(names), grad_norms=norms)
        print("mllablock", tag, variant, float(y.abs().mean()))


def golden_msmm(M):
    dims, hid = [96, 192, 384, 768], 48
    shapes = [(12, 16), (6, 8), (3, 4), (2, 2)]
    layer = M.VSS_Conv_Layer(dims, hid, depth=1, drop_path=0.1, use_checkpoint=False).eval()
    O.deterministic_fill_(layer.state_dict(), seed=3)
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(2, c, h, w, generator=g).requires_grad_(True) for c, (h, w) in zip(dims, shapes)]
    ys = layer(xs)
    gys = [torch.randn(y.shape, generator=g) for y in ys]
    torch.autograd.backward(ys, gys)
    names, norms = grad_summary(layer)
    sa = layer.blocks[0].self_attention
    np.savez_compressed(
        os.path.join(HERE, "msmm_nonsquare.npz"), shapes=np.asarray(shapes), grad_names=np.asarray(names),
        grad_norms=norms,
        **{f"x{i}": x.detach().numpy() for i, x in enumerate(xs)},
        **{f"y{i}": y.detach().numpy() for i, y in enumerate(ys)},
        **{f"gy{i}": t.numpy() for i, t in enumerate(gys)},
        **{f"gx{i}": x.grad.numpy() for i, x in enumerate(xs)},
        g_A_logs=sa.A_logs.grad.numpy(), g_Ds=sa.Ds.grad.numpy(), g_dt_bias=sa.dt_projs_bias.grad.numpy(),
        g_x_proj=sa.x_proj_weight.grad.numpy(), g_dt_w=sa.dt_projs_weight.grad.numpy())
    print("msmm", [float(y.abs().mean()) for y in ys])


def import_ss3d_module():
    """The reference's UMambaEnc_SS3D.py with stand-ins for the names it imports from absent third-party packages:
    dynamic_network_architectures' helper functions (restated: they map a dimension to torch classes), its ``BasicBlockD`` and
    MONAI's ``MLPBlock`` (arithmetic restated in oracle/umamba3d_oracle.py -- unpinned third-party boundaries)."""
    from oracle import umamba3d_oracle as U

    class BasicBlockD(U.BasicBlockD):
        def __init__(self, conv_op, input_channels, output_channels, kernel_size, stride, conv_bias, norm_op, norm_op_kwargs,
                     nonlin, nonlin_kwargs):
            assert conv_op is nn.Conv3d and input_channels == output_channels and stride == 1 and conv_bias
            assert norm_op is nn.InstanceNorm3d and norm_op_kwargs == {"eps": 1e-5, "affine": True} and nonlin is nn.LeakyReLU
            super().__init__(input_channels, list(kernel_size))

    class MLPBlock(U.MLPBlock):
        def __init__(self, hidden_size, mlp_dim, act, dropout_rate, dropout_mode):
            assert act == "GELU" and dropout_rate == 0.0 and dropout_mode == "swin"
            super().__init__(hidden_size, mlp_dim)

    def scalar_to_list(conv_op, scalar):
        return list(scalar) if isinstance(scalar, (tuple, list, np.ndarray)) else [scalar] * 3

    _mod("dynamic_network_architectures.building_blocks")
    _mod("dynamic_network_architectures.building_blocks.helper", get_matching_convtransp=lambda conv_op: nn.ConvTranspose3d,
         convert_conv_op_to_dim=lambda conv_op: 3, get_matching_instancenorm=lambda conv_op: nn.InstanceNorm3d,
         convert_dim_to_conv_op=lambda dim: {2: nn.Conv2d, 3: nn.Conv3d}[dim], maybe_convert_scalar_to_list=scalar_to_list,
         get_matching_pool_op=None)
    _mod("dynamic_network_architectures.building_blocks.residual", BasicBlockD=BasicBlockD)
    sys.modules["monai.networks.blocks"].MLPBlock = MLPBlock
    return importlib.import_module("nnunetv2.training.nnUNetTrainer.variants.mamba.UMambaEnc_SS3D")


UMAMBA3D_SMALL = dict(size=(8, 64, 64), in_ch=1, n_cls=5, batch=2,
                      strides=[[1, 1, 1], [2, 2, 2], [2, 2, 2], [2, 2, 2], [1, 2, 2], [1, 2, 2]])


def golden_umamba3d():
    """The reference's 3-D network (UMambaEnc of UMambaEnc_SS3D.py:815-888, built with the keyword arguments of
    get_umamba_enc_3d_from_plans :890-942: 6 stages, features min(32 * 2^i, 320), 3x3x3 kernels, InstanceNorm3d(affine),
    LeakyReLU, conv bias, deep supervision) on a small volume: the 5 logit maps, the base trainer's loss
    (nnUNetTrainer.py:330-352: DC_and_CE_loss, batch_dice False as in 3d_fullres plans, weights 1/2^i normalised) and every
    gradient norm."""
    from oracle import umamba3d_oracle as U
    S = import_ss3d_module()
    c = UMAMBA3D_SMALL
    n = len(c["strides"])
    net = S.UMambaEnc(input_size=c["size"], input_channels=c["in_ch"], n_stages=n, features_per_stage=U.features_for(n),
                      conv_op=nn.Conv3d, kernel_sizes=[[3, 3, 3]] * n, strides=c["strides"], n_conv_per_stage=[2] * n,
                      num_classes=c["n_cls"], n_conv_per_stage_decoder=[2] * (n - 1), conv_bias=True, norm_op=nn.InstanceNorm3d,
                      norm_op_kwargs={"eps": 1e-5, "affine": True}, dropout_op=None, dropout_op_kwargs=None,
                      nonlin=nn.LeakyReLU, nonlin_kwargs={"inplace": True}, deep_supervision=True).eval()
    O.deterministic_fill_(net.state_dict(), seed=21)
    data, target = U.synthetic_batch_3d(c["batch"], c["in_ch"], c["size"], c["strides"], c["n_cls"], seed=77)
    out = net(data)
    from nnunetv2.training.loss.compound_losses import DC_and_CE_loss
    from nnunetv2.training.loss.deep_supervision import DeepSupervisionWrapper
    from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss
    base = DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                          weight_dice=1, ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss)
    w = np.array([1 / (2 ** i) for i in range(len(out))])
    loss = DeepSupervisionWrapper(base, w / w.sum())(out, target)
    loss.backward()
    names, norms = grad_summary(net)
    keep = ("encoder.mamba_layers.0.blocks.0.self_attention.", "encoder.mamba_layers.3.blocks.0.self_attention.dt_projs",
            "decoder.seg_layers.", "encoder.stem.0.conv1.")
    small = {"grad/" + k: p.grad.numpy() for k, p in net.named_parameters()
             if p.grad is not None and k.startswith(keep) and p.numel() <= 70000}
    np.savez_compressed(os.path.join(HERE, "umamba3d_small.npz"), loss=float(loss), grad_names=np.asarray(names),
                        grad_norms=norms, state_keys=np.asarray(sorted(net.state_dict().keys())),
                        **{f"out{i}": o.detach().numpy() for i, o in enumerate(out)}, **small)
    print("umamba3d", float(loss), [tuple(o.shape) for o in out], len(names), "gradients", len(net.state_dict()), "keys")


def golden_ss3d():
    """The reference's 3-D selective-scan block SS3D (variants/mamba/UMambaEnc_SS3D.py:126-357) on a small volume: block output,
    input gradient and every parameter gradient."""
    S = import_ss3d_module()
    blk = S.SS3D(d_model=16).eval()                            # d_inner 32, dt_rank 1, d_state 16, K = 12
    O.deterministic_fill_(blk.state_dict(), seed=12)
    g = torch.Generator().manual_seed(33)
    x = torch.randn(2, 4, 6, 5, 16, generator=g).requires_grad_(True)      # (B, D, H, W, C): all three extents differ
    y = blk(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    names, norms = grad_summary(blk)
    np.savez_compressed(os.path.join(HERE, "ss3d.npz"), x=x.detach().numpy(), y=y.detach().numpy(), gy=gy.numpy(),
                        gx=x.grad.numpy(), grad_names=np.asarray(names), grad_norms=norms,
                        **{"grad/" + n: p.grad.numpy() for n, p in blk.named_parameters()})
    print("ss3d", float(y.abs().mean()), [str(n) for n in names])


def golden_loss():
    from nnunetv2.training.loss.compound_losses import DC_and_CE_loss
    from nnunetv2.training.loss.deep_supervision import DeepSupervisionWrapper
    from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss
    g = torch.Generator().manual_seed(11)
    vals = {}
    for bd in (True, False):
        base = DC_and_CE_loss({'batch_dice': bd, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                              weight_dice=1, ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss)
        w = np.array([1 / (2 ** i) for i in range(5)])
        wrap = DeepSupervisionWrapper(base, w / w.sum())
        g.manual_seed(11)
        outs = [torch.randn(3, 5, 32 >> s, 32 >> s, generator=g) for s in range(5)]
        tg = [torch.round(torch.rand(3, 1, 32 >> s, 32 >> s, generator=g) * 4) for s in range(5)]
        vals[f"loss_batch_dice_{int(bd)}"] = float(wrap(outs, tg))
    np.savez_compressed(os.path.join(HERE, "loss.npz"), seed=11, **vals)
    print("loss", vals)


def loss_ignore_case():
    """Inputs of the ignore-label loss golden (shared with the tests): 5 deep-supervision levels, 5 classes + ignore label 5;
    level 3 of sample 0 and ALL of the coarsest level are fully ignored (the latter exercises the skipped cross-entropy)."""
    g = torch.Generator().manual_seed(23)
    outs = [torch.randn(3, 5, 32 >> s, 32 >> s, generator=g) for s in range(5)]
    tg = []
    for s in range(5):
        t = torch.round(torch.rand(3, 1, 32 >> s, 32 >> s, generator=g) * 4)
        t[torch.rand(t.shape, generator=g) < 0.3] = 5.0
        tg.append(t)
    tg[3][0] = 5.0
    tg[4][:] = 5.0
    return outs, tg


def golden_loss_ignore():
    """DeepSupervisionWrapper(DC_and_CE_loss(ignore_label=5)) of the reference (T:106-129 with a label manager that has an
    ignore label; L/compound_losses.py:38-51): value and gradient w.r.t. every level's logits."""
    from nnunetv2.training.loss.compound_losses import DC_and_CE_loss
    from nnunetv2.training.loss.deep_supervision import DeepSupervisionWrapper
    from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss
    vals = {}
    for bd in (True, False):
        base = DC_and_CE_loss({'batch_dice': bd, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                              weight_dice=1, ignore_label=5, dice_class=MemoryEfficientSoftDiceLoss)
        w = np.array([1 / (2 ** i) for i in range(5)])
        wrap = DeepSupervisionWrapper(base, w / w.sum())
        outs, tg = loss_ignore_case()
        outs = [o.requires_grad_(True) for o in outs]
        loss = wrap(outs, tg)
        loss.backward()
        vals[f"loss_batch_dice_{int(bd)}"] = float(loss)
        for s, o in enumerate(outs):
            vals[f"grad{s}_batch_dice_{int(bd)}"] = o.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "loss_ignore.npz"), seed=23, **vals)
    print("loss (ignore label)", {k: v for k, v in vals.items() if k.startswith("loss")})


def golden_sliding_window():
    """Reference predict_sliding_window_return_logits (sliding_window_prediction.py:118-210) on its CPU branch, with
    acvl_utils.pad_nd_image (third-party, absent) replaced by the oracle's restatement."""
    from oracle import inference_oracle as IO

    def pad_nd_image(image, new_shape, mode, kwargs, return_slicer, shape_must_be_divisible_by=None):
        assert mode == "constant" and return_slicer
        return IO.pad_nd_image(image, new_shape, kwargs.get("value", 0))

    _mod("acvl_utils")
    _mod("acvl_utils.cropping_and_padding")
    _mod("acvl_utils.cropping_and_padding.padding", pad_nd_image=pad_nd_image)
    S = importlib.import_module("nnunetv2.inference.sliding_window_prediction")
    net, img, small = IO.sliding_window_case()
    out = {}
    for tag, image, mirror in (("mirror", img, (0, 1)), ("plain", img, None), ("padded", small, (1,))):
        r = S.predict_sliding_window_return_logits(net, image, 3, (32, 32), mirror_axes=mirror, tile_step_size=0.5,
                                                   use_gaussian=True, perform_everything_on_gpu=False,
                                                   verbose=False, device=torch.device("cpu"))
        out[tag] = r.float().numpy()
    out["gaussian_32"] = S.compute_gaussian((32, 32)).astype(np.float32)
    out["gaussian_256"] = S.compute_gaussian((256, 256)).astype(np.float32)[::16, ::16]
    steps = [S.compute_steps_for_sliding_window(a, b, c) for a, b, c in
             (((40, 50), (32, 32), 0.5), ((512, 640), (256, 256), 0.5), ((256, 256), (256, 256), 0.5), ((300, 257), (256, 256), 0.25))]
    out["steps"] = np.asarray([v for s_ in steps for ax in s_ for v in ax + [-1]])
    np.savez_compressed(os.path.join(HERE, "sliding_window.npz"), **out)
    print("sliding window", {k: v.shape for k, v in out.items()})


def golden_evaluation():
    """get_tp_fp_fn_tn (training/loss/dice.py:120-178) driven as validation_step drives it (nnUNetTrainer.py:899-940),
    and compute_dice_coefficient (evaluation/SurfaceDice.py:481-498) on the oracle's synthetic label volumes."""
    from oracle import evaluation_oracle as EO
    D = importlib.import_module("nnunetv2.training.loss.dice")
    sys.path.insert(0, os.path.join(os.path.dirname(REF), "evaluation"))
    SD = importlib.import_module("SurfaceDice")
    g = torch.Generator().manual_seed(41)
    logits = torch.randn(3, 6, 24, 20, generator=g)
    target = torch.round(torch.rand(3, 1, 24, 20, generator=g) * 5)
    seg = logits.argmax(1)[:, None]
    onehot = torch.zeros(logits.shape, dtype=torch.float32)
    onehot.scatter_(1, seg, 1)
    tp, fp, fn, _ = D.get_tp_fp_fn_tn(onehot, target, axes=[0, 2, 3], mask=None)
    gt, sg = EO.evaluation_case()
    # both-empty labels: the reference returns np.NaN, an attribute NumPy 2 removed -- recorded as NaN here
    dsc = [float(SD.compute_dice_coefficient(gt == i, sg == i)) if (gt == i).sum() + (sg == i).sum() else float("nan")
           for i in range(1, 14)]
    np.savez_compressed(os.path.join(HERE, "evaluation.npz"), seed=41, tp=tp[1:].numpy(), fp=fp[1:].numpy(),
                        fn=fn[1:].numpy(), whole_volume_dsc=np.asarray(dsc))
    print("evaluation", tp[1:].tolist(), dsc[:4])


def golden_dataloader():
    """nnUNetDataLoader2D.generate_train_batch of the reference (training/dataloading/data_loader_2d.py:7-86 on
    base_data_loader.py and nnunet_dataset.py) on the oracle's synthetic preprocessed folder, numpy seeded.  Stand-ins: the
    batchgenerators DataLoader base class (its get_indices for infinite=True restated) and file helpers, LabelManager."""
    import pickle
    import tempfile
    import typing
    from oracle import dataloading_oracle as DO

    class DataLoader:
        def __init__(self, data, batch_size, num_threads_in_multithreaded=1, seed_for_shuffle=None, return_incomplete=False,
                     shuffle=True, infinite=False, sampling_probabilities=None):
            self._data, self.batch_size, self.infinite = data, batch_size, infinite
            self.sampling_probabilities, self.indices = sampling_probabilities, None

        def get_indices(self):
            assert self.infinite
            return np.random.choice(self.indices, self.batch_size, replace=True, p=self.sampling_probabilities)

    def load_pickle(f):
        with open(f, "rb") as fh:
            return pickle.load(fh)

    _mod("batchgenerators")
    _mod("batchgenerators.dataloading")
    _mod("batchgenerators.dataloading.data_loader", DataLoader=DataLoader)
    _mod("batchgenerators.utilities")
    _mod("batchgenerators.utilities.file_and_folder_operations", join=os.path.join, isfile=os.path.isfile,
         load_pickle=load_pickle, subfiles=None, List=typing.List, os=os)
    _mod("nnunetv2.configuration", default_num_processes=1)
    _mod("nnunetv2.utilities.label_handling")
    _mod("nnunetv2.utilities.label_handling.label_handling", LabelManager=object)
    D2 = importlib.import_module("nnunetv2.training.dataloading.data_loader_2d")
    DS = importlib.import_module("nnunetv2.training.dataloading.nnunet_dataset")

    class LM:
        all_labels = [0, 1, 2, 3]
        has_ignore_label = False

    class LMIgnore(LM):
        has_ignore_label = True                      # label 4 = ignore: partially annotated cases (base_data_loader.py:91-97)

    out = {}
    for tag, unpack, patch, final, bs, fg in (("npz", False, (40, 48), (32, 32), 4, 0.33), ("npy", True, (32, 32), (32, 32), 5, 0.5),
                                              ("ign", True, (36, 36), (32, 32), 6, 0.33)):
        folder = tempfile.mkdtemp()
        DO.write_synthetic_dataset(folder, unpack=unpack, ignore_label=4 if tag == "ign" else None)
        dl = D2.nnUNetDataLoader2D(DS.nnUNetDataset(folder), bs, patch, final, LMIgnore() if tag == "ign" else LM(),
                                   oversample_foreground_percent=fg, sampling_probabilities=None, pad_sides=None)
        np.random.seed(7)
        for it in range(3):
            b = dl.generate_train_batch()
            out[f"{tag}_data_{it}"] = b["data"]
            out[f"{tag}_seg_{it}"] = b["seg"]
            out[f"{tag}_keys_{it}"] = np.asarray([str(k) for k in b["keys"]])
    np.savez_compressed(os.path.join(HERE, "dataloader.npz"), **out)
    print("dataloader", {k: v.shape for k, v in out.items() if k.endswith("_0")})


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    T, M = import_reference()
    if "--config" in sys.argv:
        for tag in sys.argv[sys.argv.index("--config") + 1:]:
            golden_full_model_config(T, tag)
        sys.exit(0)
    if "--grads" in sys.argv:
        for tag in sys.argv[sys.argv.index("--grads") + 1:]:
            golden_gradient_tensors(T, tag)
        sys.exit(0)
    if "--only-train-mode" in sys.argv:
        golden_train_mode(T)
        sys.exit(0)
    if "--only-ss3d" in sys.argv:
        golden_ss3d()
        sys.exit(0)
    if "--only-umamba3d" in sys.argv:
        golden_umamba3d()
        sys.exit(0)
    if "--only-loss" in sys.argv:
        golden_loss()
        golden_loss_ignore()
        sys.exit(0)
    if "--only-256" in sys.argv:
        golden_full_model_256(T)
        sys.exit(0)
    if "--only-sliding-window" in sys.argv:
        golden_sliding_window()
        golden_evaluation()
        golden_dataloader()
        sys.exit(0)
    golden_sliding_window()
    golden_evaluation()
    golden_dataloader()
    golden_loss()
    golden_loss_ignore()
    golden_msmm(M)
    for v in ("B", "A"):
        golden_mllablock(T, v)
        golden_full_model(T, v)
    golden_full_model_256(T)
    for tag in CONFIG_GOLDENS:
        golden_full_model_config(T, tag)
    golden_ss3d()
    golden_umamba3d()
