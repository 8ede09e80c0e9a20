"""Drop-in modules named like the reference's third-party imports, so that the reference's own model
files run unmodified on MI355X:

    import mlagg_unet_amd.shims as shims; shims.install()
    from nnunetv2.training.nnUNetTrainer.variants.mamba.MambaSkip import VSS_Conv_Layer   # reference M

* ``mamba_ssm.ops.selective_scan_interface.selective_scan_fn``  (imported at MambaSkip.py:18, called at
  M:445-451) -> K1, the HIP selective scan.
* ``flash_attn.flash_attn_func`` (imported unconditionally at nnUNetTrainer_MLAgg_2D_dt_MS.py:173, called at
  T:745-750) -> the HIP attention kernels of csrc/flash_attn.hip (ops.flash_attn: fp16 / bf16 tensors, head_dim 24,
  forward and backward), with flash-attn's own contract: 16-bit CUDA tensors only.  The product network does not use
  this shim: its pooled branch is the single fused K4 launch (ops.pooled_diff_attn) instead of four calls.
"""
import sys
import types

from . import ops


def selective_scan_ref(*args, **kwargs):
    raise RuntimeError("selective_scan_ref (the eager CPU path) is not provided by the MI355X build; "
                       "use selective_scan_fn")


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, **unused):
    """q (B, N, nh, e), k/v (B, P, nh, e) -> (B, N, nh, e); default softmax_scale = e^-0.5 (flash-attn)."""
    if dropout_p != 0.0:
        raise RuntimeError("flash_attn_func shim: dropout is not on the MLAgg-UNet path")
    if causal:
        raise RuntimeError("flash_attn_func shim: causal attention is not on the MLAgg-UNet path (T:745-750 pass causal=False)")
    return ops.flash_attn(q, k, v, softmax_scale)          # raises for fp32 / host tensors, as flash-attn itself does


def install():
    """Register the shim modules (idempotent).  Real mamba_ssm / flash_attn wheels, if present, are shadowed."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    root = mod("mamba_ssm")
    opsm = mod("mamba_ssm.ops")
    iface = mod("mamba_ssm.ops.selective_scan_interface", selective_scan_fn=ops.selective_scan_fn,
                selective_scan_ref=selective_scan_ref)
    root.ops = opsm
    opsm.selective_scan_interface = iface
    mod("flash_attn", flash_attn_func=flash_attn_func)
