"""torch.autograd.Function wrappers over the C ABI (include/mlagg_hip.h).

PyTorch is plumbing here: device memory, the current HIP stream and autograd bookkeeping.  Every
op requires CUDA(HIP) fp32 tensors and raises RuntimeError otherwise -- there is no eager path.
"""
import torch

from . import _lib


# ------------------------------------------------------------------------------------------------
# arithmetic type of the dense products.  The reference's default train step runs the network under autocast
# (nnUNetTrainer.py:848: fp16 + GradScaler; BASELINE configs[2]: bf16): Linear / convolution operands are rounded to 16
# bits, sums are fp32.  Here tensors stay fp32 in HBM in every mode; in "bf16" / "fp16" mode the projections (K5) and
# the library convolutions / small GEMMs round their OPERANDS to that type for the matrix cores.
# ------------------------------------------------------------------------------------------------
PRECISIONS = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
_LP_CODE = {torch.bfloat16: 1, torch.float16: 2}            # MLAGG_DTYPE_* of include/mlagg_hip.h
import threading as _threading


class _ComputeStack(_threading.local):
    """Per-thread stack (autograd's backward thread and data-loader threads never see another thread's block)."""

    def __init__(self):
        self.stack = [torch.float32]


_COMPUTE = _ComputeStack()


class compute_precision:
    """with ops.compute_precision("bf16"): ... -- arithmetic type of the dense products inside the block."""

    def __init__(self, precision):
        if precision not in PRECISIONS:
            raise RuntimeError(f"precision {precision!r}: one of {tuple(PRECISIONS)}")
        self.dtype = PRECISIONS[precision]

    def __enter__(self):
        _COMPUTE.stack.append(self.dtype)
        return self

    def __exit__(self, *exc):
        _COMPUTE.stack.pop()
        return False


def compute_dtype():
    return _COMPUTE.stack[-1]


def lp(t, dtype):
    """Operand of a library GEMM / convolution in the arithmetic type of the block (identity in fp32 mode)."""
    return t if (t is None or dtype == torch.float32) else t.to(dtype)


# Library convolutions in 16-bit mode: ON by default -- the literal operand rounding of the reference's autocast step.  The
# maps are fp32 in HBM, so a 16-bit MIOpen convolution costs a cast kernel on its input and another on its output (and again
# in backward), but the 16-bit convolutions themselves are so much faster that the step gains: config 3 (224 x 224, bf16)
# 42.2 -> 34.7 ms, config 5 (512 x 640, fp16) 98.0 -> 82.9 ms.  (Early in round 2, before the fused conv epilogues, the casts
# still outweighed the gain.)  MLAGG_LP_CONV=0 keeps the convolutions in fp32.
import os as _os
LP_CONV = _os.environ.get("MLAGG_LP_CONV", "1") == "1"


# 16-bit modes: maps that only 16-bit library convolutions read or write stay bf16 / fp16 in memory; K10 / K13 convert on the fly
# (MLAGG_LP_IO=0: the round-2 form, fp32 maps everywhere with a cast kernel on both sides of every convolution)
LP_IO = _os.environ.get("MLAGG_LP_IO", "1") == "1"


def conv_dtype():
    return compute_dtype() if LP_CONV else torch.float32


# 16-bit modes, round 4: the 1 x 1 / 3 x 3 stride-1 convolutions run on K18 / K19 in the ONE-product operand form (csrc/opmode.h: operands
# rounded once to bf16 / fp16 in registers, fp32 sums) straight on the fp32 maps -- the kernels of the fp32 step at a sixth of its matrix
# work, no cast kernels, no NHWC transposes.  Maps below LP_K_MIN_PIXELS pixels, strided and transposed convolutions stay 16-bit library
# calls.  MLAGG_LP_K=0: every convolution of the 16-bit modes on the library (the round-3 form).
LP_K = _os.environ.get("MLAGG_LP_K", "1") == "1"
LP_K_MIN_PIXELS = int(_os.environ.get("MLAGG_LP_K_MIN_PIXELS", "1024"))
_DTYPE_BF16X3 = 3
_FORM_TORCH = {1: torch.bfloat16, 2: torch.float16, 3: torch.float32}


def conv_form():
    """Operand form (MLAGG_DTYPE_* code) of the K18 / K19 products in the current mode: three bf16 pieces (fp32), or one rounded operand."""
    cdt = conv_dtype()
    return _DTYPE_BF16X3 if cdt == torch.float32 else _LP_CODE[cdt]


def _lib_conv_fwd(x, w, padding, form):
    """Library forward of a stride-1 convolution inside a K18 / K19 Function, in the arithmetic of `form` (fp32 maps in and out)."""
    if form == _DTYPE_BF16X3:
        return torch.nn.functional.conv2d(x, w, None, 1, padding)
    t = _FORM_TORCH[form]
    return torch.nn.functional.conv2d(x.to(t), w.to(t), None, 1, padding).float()


def _lib_conv_bwd(dy, x, w, padding, mask, form):
    if form != _DTYPE_BF16X3:
        t = _FORM_TORCH[form]
        dy, x, w = dy.to(t), x.to(t), w.to(t)
    out = torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (padding, padding), (1, 1), False, (0, 0), 1, mask)
    return [None if o is None else o.float() for o in out]


def _ptr(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# Leaf gradients on a second stream.  Nothing in backward consumes a weight / bias gradient: only the optimizer (and the data-parallel
# exchange) read them, after the last backward kernel.  Inside ``with leaf_grad_overlap():`` (trainer.train_step wraps backward in
# it) the weight-gradient launches of the projection and convolution Functions go to a side stream that forks off the data-gradient
# chain where their operands are ready; the context's exit joins it back.  The data-gradient chain -- the critical path of backward --
# gets shorter by the weight-gradient kernels and their reductions, which fill the gaps the small-grid kernels of the chain leave
# on the 256 CUs.  Operands handed to the side stream stay referenced until the join (see _LeafStream).  Captured into a hipGraph
# the fork / join become graph edges.
# Outside the context (a bare ``loss.backward()``) everything stays on the current stream.
# ------------------------------------------------------------------------------------------------
# Measured (profiles/round4_e_leaf_stream_ab.log): 37.9 -> 37.7 ms per eager step (0.6 %), but 38.4 -> 39.8 ms under hipGraph replay (the
# fork / join edges cost more than the overlap returns: the backward chain's kernels already fill the chip).  OFF by default.
LEAF_STREAM = _os.environ.get("MLAGG_LEAF_STREAM", "0") == "1"
_LEAF = {"on": False, "side": None, "main": None, "used": False, "epoch": 0, "keep": []}


def leaf_grads_ready():
    """Make the CURRENT stream wait for every leaf gradient enqueued so far (the data-parallel exchange calls it before it gathers a
    bucket in the middle of backward)."""
    if _LEAF["on"] and _LEAF["used"]:
        torch.cuda.current_stream().wait_stream(_LEAF["side"])


class leaf_grad_overlap:
    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        if self.enabled and LEAF_STREAM and torch.cuda.is_available():
            _LEAF["main"] = torch.cuda.current_stream()
            if _LEAF["side"] is None or _LEAF["side"].device != _LEAF["main"].device:
                _LEAF["side"] = torch.cuda.Stream(device=_LEAF["main"].device)
            _LEAF["on"], _LEAF["used"] = True, False
        return self

    def __exit__(self, *exc):
        _LEAF["epoch"] += 1                                      # the uses counted by note_leaf_use belong to the forward just consumed
        if _LEAF["on"]:
            _LEAF["on"] = False
            if _LEAF["used"]:
                _LEAF["main"].wait_stream(_LEAF["side"])         # every leaf gradient is complete before anything reads it
            _LEAF["keep"].clear()                                # (freed behind the join: reuse on the main stream is ordered after it)
        return False


_LEAF_USES = {}


def _leaf_base(t):
    return t._base if t._is_view() and t._base is not None else t


def note_leaf_use(*tensors):
    """Forward-time bookkeeping: how many projections of THIS forward a parameter feeds (see leaf_single_use)."""
    for t in tensors:
        if t is None:
            continue
        p = _leaf_base(t)
        if not (p.is_leaf and p.requires_grad):
            continue
        u = _LEAF_USES.get(id(p))
        if u is None or u[0] != _LEAF["epoch"]:
            _LEAF_USES[id(p)] = [_LEAF["epoch"], 1, p]         # (p itself is kept: an id must not be recycled)
        else:
            u[1] += 1


def leaf_single_use(tensors):
    """True when every parameter behind ``tensors`` received ONE use in the forward being differentiated.  A parameter that feeds two
    projections (kv.weight of the pooled branch: a row slice inside the stacked q | v | sr projection and the whole matrix on the
    pooled tokens) gets two gradient contributions, and AccumulateGrad adds the second in place on the backward stream: both must
    then be produced on that stream, so such projections keep their weight gradients off the leaf-gradient stream."""
    for t in tensors:
        if t is None:
            continue
        u = _LEAF_USES.get(id(_leaf_base(t)))
        if u is not None and u[0] == _LEAF["epoch"] and u[1] > 1:
            return False
    return True


def _leaf_sources(*tensors):
    out = []
    for t in tensors:
        if t is not None:
            out += getattr(t, "_mlagg_sources", [t])
    return out


def _leaf_ok(*tensors):
    """True when the gradients of these forward arguments go NOWHERE but to AccumulateGrad nodes: parameters themselves, or tensors
    whose producer promises it (weight stacks, the padded x_proj: ``_mlagg_leaf_safe``).  A gradient that another backward node reads
    (a sliced or cast weight) must be produced on the backward stream."""
    return all(t is None or t.is_leaf or getattr(t, "_mlagg_leaf_safe", False) for t in tensors)


class _LeafStream:
    """with _LeafStream(dy, x, ...): launches inside go to the side stream (when the overlap is on), behind everything enqueued so far."""

    def __init__(self, *operands, ok=True):
        self.operands, self.prev, self.ok = operands, None, ok

    def __enter__(self):
        if _LEAF["on"] and self.ok:
            cur = torch.cuda.current_stream()
            side = _LEAF["side"]
            if cur.device == side.device:
                side.wait_stream(cur)
                # the operands stay REFERENCED until the join: (i) their memory cannot be handed out again while the side stream
                # reads it, and (ii) autograd cannot add another gradient contribution INTO them in place on the backward stream (it
                # does that to a gradient nobody else holds: the dy of out_proj is also the gradient of the residual skip)
                _LEAF["keep"].extend(t for t in self.operands if t is not None)
                self.prev = cur
                torch.cuda.set_stream(side)
                _LEAF["used"] = True
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_stream(self.prev)
        return False


# flop of the matrix products this package's own kernels run, per family (bench.py's roofline.mfma): None = not counting
FLOP_COUNT = None


def _flop(family, n):
    if FLOP_COUNT is not None:
        FLOP_COUNT[family] = FLOP_COUNT.get(family, 0) + int(n)


def _stream():
    """Raw handle of torch's current HIP stream on the current device.  Every kernel wrapper asks for it: the two C calls below cost
    0.3 us, `torch.cuda.current_stream().cuda_stream` 9 us (tools/host_profile.py: 2.6 ms of host time per direction and step, and
    the 224 x 224 configurations are bound by the host's enqueue rate)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _require(t, name, shape=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f"{name}: expected a float32 tensor on the MI355X device, got "
                           f"{getattr(t, 'dtype', type(t))} on {getattr(t, 'device', '?')}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


class SelectiveScanFn(torch.autograd.Function):
    """K1.  Same contract as mamba-ssm's SelectiveScanFn as used at MambaSkip.py:445-451."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D, delta_bias, delta_softplus):
        b, d, L = u.shape
        n, g = A.shape[1], B.shape[1]
        u = _require(u.contiguous(), "u")
        delta = _require(delta.contiguous(), "delta", (b, d, L))
        A = _require(A.contiguous(), "A", (d, n))
        B = _require(B.contiguous(), "B", (b, g, n, L))
        C = _require(C.contiguous(), "C", (b, g, n, L))
        D = None if D is None else _require(D.contiguous(), "D", (d,))
        delta_bias = None if delta_bias is None else _require(delta_bias.contiguous(), "delta_bias", (d,))
        lib = _lib.lib()
        out = torch.empty_like(u)
        state = torch.empty(lib.mlagg_selscan_state_floats(b, d, L, n), device=u.device, dtype=torch.float32)
        _lib.check(lib.mlagg_selscan_fwd(_ptr(u), _ptr(delta), _ptr(A), _ptr(B), _ptr(C), _ptr(D),
                                         _ptr(delta_bias), _ptr(out), _ptr(state), b, d, L, n, g,
                                         int(bool(delta_softplus)), _stream()), "mlagg_selscan_fwd")
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, state)
        ctx.delta_softplus = bool(delta_softplus)
        return out

    @staticmethod
    def backward(ctx, dout):
        u, delta, A, B, C, D, delta_bias, state = ctx.saved_tensors
        b, d, L = u.shape
        n, g = A.shape[1], B.shape[1]
        dout = _require(dout.contiguous(), "dout", (b, d, L))
        lib = _lib.lib()
        du, ddelta = torch.empty_like(u), torch.empty_like(u)
        dA, dB, dC = torch.empty_like(A), torch.empty_like(B), torch.empty_like(C)
        dD = None if D is None else torch.empty_like(D)
        dbias = None if delta_bias is None else torch.empty_like(delta_bias)
        ws = torch.empty(lib.mlagg_selscan_bwd_workspace_floats(b, d, L, n), device=u.device, dtype=torch.float32)
        _lib.check(lib.mlagg_selscan_bwd(_ptr(u), _ptr(delta), _ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(delta_bias),
                                         _ptr(dout), _ptr(state), _ptr(du), _ptr(ddelta), _ptr(dA), _ptr(dB),
                                         _ptr(dC), _ptr(dD), _ptr(dbias), _ptr(ws), b, d, L, n, g,
                                         int(ctx.delta_softplus), _stream()), "mlagg_selscan_bwd")
        return du, ddelta, dA, dB, dC, dD, dbias, None


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """Drop-in for ``mamba_ssm.ops.selective_scan_interface.selective_scan_fn`` on the arguments the
    reference passes (MambaSkip.py:445-451: z=None, return_last_state=False).  Anything else raises."""
    if z is not None or return_last_state:
        raise RuntimeError("selective_scan_fn: z gating / return_last_state are not on the MLAgg-UNet path")
    return SelectiveScanFn.apply(u, delta, A, B, C, D, delta_bias, delta_softplus)


class SelectiveScanLowRankFn(torch.autograd.Function):
    """K1 with the rank-R delta projection folded in: delta = softplus(Wdt[d] . dtr[b, g(d), :, l] + delta_bias[d]) is formed
    inside the scan kernels, i.e. SS2D_skip's ``einsum("b k r l, k d r -> b k d l", dts, dt_projs_weight)`` +
    ``selective_scan_fn(..., delta_bias, delta_softplus=True)`` (reference MambaSkip.py:430-451) as one op.  The
    (B, 4*96, L) delta tensor and its gradient (334 MB each at config 2) are never materialised."""

    @staticmethod
    def forward(ctx, u, dtr, Wdt, A, B, C, D, delta_bias, delta_softplus):
        b, d, L = u.shape
        n, g, R = A.shape[1], B.shape[1], dtr.shape[2]
        u = _require(u.contiguous(), "u")
        dtr = _require(dtr.contiguous(), "dtr", (b, g, R, L))
        Wdt = _require(Wdt.contiguous(), "Wdt", (d, R))
        A = _require(A.contiguous(), "A", (d, n))
        B = _require(B.contiguous(), "B", (b, g, n, L))
        C = _require(C.contiguous(), "C", (b, g, n, L))
        D = None if D is None else _require(D.contiguous(), "D", (d,))
        delta_bias = None if delta_bias is None else _require(delta_bias.contiguous(), "delta_bias", (d,))
        lib = _lib.lib()
        out = torch.empty_like(u)
        state = torch.empty(lib.mlagg_selscan_state_floats(b, d, L, n), device=u.device, dtype=torch.float32)
        _lib.check(lib.mlagg_selscan_lowrank_fwd(_ptr(u), _ptr(dtr), _ptr(Wdt), R, _ptr(A), _ptr(B), _ptr(C), _ptr(D),
                                                 _ptr(delta_bias), _ptr(out), _ptr(state), b, d, L, n, g,
                                                 int(bool(delta_softplus)), _stream()), "mlagg_selscan_lowrank_fwd")
        ctx.save_for_backward(u, dtr, Wdt, A, B, C, D, delta_bias, state)
        ctx.delta_softplus = bool(delta_softplus)
        return out

    @staticmethod
    def backward(ctx, dout):
        u, dtr, Wdt, A, B, C, D, delta_bias, state = ctx.saved_tensors
        b, d, L = u.shape
        n, g, R = A.shape[1], B.shape[1], dtr.shape[2]
        dout = _require(dout.contiguous(), "dout", (b, d, L))
        lib = _lib.lib()
        du, ddtr, dW = torch.empty_like(u), torch.empty_like(dtr), torch.empty_like(Wdt)
        dA, dB, dC = torch.empty_like(A), torch.empty_like(B), torch.empty_like(C)
        dD = None if D is None else torch.empty_like(D)
        dbias = None if delta_bias is None else torch.empty_like(delta_bias)
        ws = torch.empty(lib.mlagg_selscan_bwd_workspace_floats(b, d, L, n), device=u.device, dtype=torch.float32)
        _lib.check(lib.mlagg_selscan_lowrank_bwd(_ptr(u), _ptr(dtr), _ptr(Wdt), R, _ptr(A), _ptr(B), _ptr(C), _ptr(D),
                                                 _ptr(delta_bias), _ptr(dout), _ptr(state), _ptr(du), _ptr(ddtr), _ptr(dW),
                                                 _ptr(dA), _ptr(dB), _ptr(dC), _ptr(dD), _ptr(dbias), _ptr(ws), b, d, L, n, g,
                                                 int(ctx.delta_softplus), _stream()), "mlagg_selscan_lowrank_bwd")
        return du, ddtr, dW, dA, dB, dC, dD, dbias, None


def selective_scan_lowrank_fn(u, dtr, Wdt, A, B, C, D=None, delta_bias=None, delta_softplus=False):
    return SelectiveScanLowRankFn.apply(u, dtr, Wdt, A, B, C, D, delta_bias, delta_softplus)


# ------------------------------------------------------------------------------------------------
# K1f: the MSMM scan on token-major tensors (csrc/selscan_tok.hip) -- SS2D_skip.forward_corev0 behind x_proj + the four-way sum
# (reference MambaSkip.py:405-473, 534) as ONE op; MLAGG_MSMM_FUSED=0: the round-3 chain (K1' cross_scan / cross_merge around K1)
# ------------------------------------------------------------------------------------------------
MSMM_FUSED = _os.environ.get("MLAGG_MSMM_FUSED", "1") == "1"
MSMM_XB = 36                      # floats per direction of a padded x_proj row: [dt0 dt1 dt2 0 | B(16) | C(16)]
_SCAN_INDEX = {}


def msmm_scan_index(HW, device):
    """(4, L_cat) int32: the token direction k visits at scan position t -- the orders of reference M:419-422 (k = 0 row-major,
    1 column-major, 2 / 3 their reversals inside every scale; scales concatenated in the same order for every direction)."""
    key = (tuple((int(h), int(w)) for h, w in HW), str(device))
    if key not in _SCAN_INDEX:
        rows, off = [[], [], [], []], 0
        for H, W in key[0]:
            n = H * W
            p = torch.arange(n, dtype=torch.int64)
            col = (p % H) * W + p // H                       # position p of the column-major walk -> token y * W + x
            for k, tok in enumerate((p, col, n - 1 - p, col.flip(0))):
                rows[k].append(tok + off)
            off += n
        table = torch.stack([torch.cat(r) for r in rows]).to(torch.int32)
        if not all(bool((table[k].sort().values == torch.arange(off, dtype=torch.int32)).all()) for k in range(4)):
            raise RuntimeError("msmm_scan_index: a direction is not a permutation of the tokens")
        _SCAN_INDEX[key] = table.to(device)
    return _SCAN_INDEX[key]


class PadXProjFn(torch.autograd.Function):
    """x_proj_weight (4, R + 2N, d) -> (4 * 36, d) with a zero row behind the three dt rows of every direction, so that the B / C
    blocks of a projection row start on 16-byte boundaries; backward drops the pad rows' (exactly zero) gradient."""

    @staticmethod
    def forward(ctx, w):
        K, per, dI = w.shape
        ctx.per = per
        ctx.leaf, ctx.leaf_params = w.is_leaf, [w]
        note_leaf_use(w)
        out = w.new_zeros(K, MSMM_XB, dI)
        out[:, :3] = w[:, :3]
        out[:, 4:] = w[:, 3:]
        return out.view(K * MSMM_XB, dI)

    @staticmethod
    def backward(ctx, g):
        with _LeafStream(g, ok=ctx.leaf and leaf_single_use(ctx.leaf_params)):   # the projection's weight gradient may live on the leaf-gradient stream
            g3 = g.view(-1, MSMM_XB, g.shape[-1])
            return torch.cat([g3[:, :3], g3[:, 4:]], dim=1)


def pad_x_proj(w):
    out = PadXProjFn.apply(w)
    if w.is_leaf:
        out._mlagg_leaf_safe = True                    # its gradient reaches x_proj_weight's AccumulateGrad through PadXProjFn only
        out._mlagg_sources = [w]
    return out


class MsmmScanFn(torch.autograd.Function):
    """K1f.  xc (B, L, 96), xdbl (B, L, 144), idx (4, L) int32, Wdt (384, 3), A (384, 16), D (384), bias (384) -> y (B, L, 96)."""

    @staticmethod
    def forward(ctx, xc, xdbl, idx, Wdt, A, D, bias):
        xc = _require(xc.contiguous(), "xc")
        B, L, dI = xc.shape
        xdbl = _require(xdbl.contiguous(), "x_dbl", (B, L, 4 * MSMM_XB))
        if idx.dtype != torch.int32 or tuple(idx.shape) != (4, L) or not idx.is_cuda or not idx.is_contiguous():
            raise RuntimeError("msmm_scan: idx must be a contiguous int32 (4, L) device table")
        lib = _lib.lib()
        if not lib.mlagg_msmm_scan_supported(dI, int(A.shape[1]), int(Wdt.shape[1]), 4, L):
            raise RuntimeError(f"msmm_scan: shape (d_inner {dI}, d_state {A.shape[1]}, rank {Wdt.shape[1]}, L {L}) is outside K1f")
        Wdt = _require(Wdt.contiguous(), "Wdt", (4 * dI, 3))
        A = _require(A.contiguous(), "A", (4 * dI, 16))
        D = None if D is None else _require(D.contiguous(), "D", (4 * dI,))
        bias = None if bias is None else _require(bias.contiguous(), "delta_bias", (4 * dI,))
        y = torch.empty(B, L, dI, device=xc.device, dtype=torch.float32)
        state = torch.empty(lib.mlagg_msmm_scan_state_floats(B, L), device=xc.device, dtype=torch.float32)
        ws = torch.empty(lib.mlagg_msmm_scan_fwd_workspace_floats(B, L), device=xc.device, dtype=torch.float32)
        _lib.check(lib.mlagg_msmm_scan_fwd(_ptr(xc), _ptr(xdbl), _ptr(idx), _ptr(Wdt), _ptr(A), _ptr(D), _ptr(bias), _ptr(y), _ptr(state),
                                           _ptr(ws), B, L, _stream()), "mlagg_msmm_scan_fwd")
        ctx.save_for_backward(xc, xdbl, idx, Wdt, A, D, bias, state)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, xdbl, idx, Wdt, A, D, bias, state = ctx.saved_tensors
        B, L, dI = xc.shape
        dy = _require(dy.contiguous(), "dy", (B, L, dI))
        lib = _lib.lib()
        dev = xc.device
        dxc, dxdbl = torch.empty_like(xc), torch.empty_like(xdbl)
        dW, dA = torch.empty_like(Wdt), torch.empty_like(A)
        dD = None if D is None else torch.empty_like(D)
        dbias = None if bias is None else torch.empty_like(bias)
        ws = torch.empty(lib.mlagg_msmm_scan_bwd_workspace_floats(B, L), device=dev, dtype=torch.float32)
        _lib.check(lib.mlagg_msmm_scan_bwd(_ptr(xc), _ptr(xdbl), _ptr(idx), _ptr(Wdt), _ptr(A), _ptr(D), _ptr(bias), _ptr(dy), _ptr(state),
                                           _ptr(dxc), _ptr(dxdbl), _ptr(dW), _ptr(dA), _ptr(dD), _ptr(dbias), _ptr(ws), B, L, _stream()),
                   "mlagg_msmm_scan_bwd")
        return dxc, dxdbl, None, dW, dA, dD, dbias


def msmm_scan(xc, xdbl, idx, Wdt, A, D=None, delta_bias=None):
    return MsmmScanFn.apply(xc, xdbl, idx, Wdt, A, D, delta_bias)


def msmm_scan_supported(xc, d_state, dt_rank):
    return bool(MSMM_FUSED and xc.is_cuda and xc.dim() == 3 and
                _lib.lib().mlagg_msmm_scan_supported(int(xc.shape[2]), int(d_state), int(dt_rank), 4, int(xc.shape[1])))


def _rows(t, name):
    """(B, N, C) view whose last dim is contiguous and whose batch/token dims collapse to one row
    stride (true for fresh Linear outputs and their channel slices); returns (tensor, row_stride)."""
    _require(t, name)
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1) or t.stride(1) % 4 or \
            t.data_ptr() % 16:
        t = t.contiguous()
    return t, t.stride(1)


class _GradSlot:
    """Column block [col0, col0 + width) of the gradient buffer of one ``split_cols`` call.  The FIRST kernel wrapper that consumes the
    piece claims the slot; its backward kernel then writes d(piece) straight into the shared (B, N, total) buffer (row stride =
    total), so the split's backward hands that buffer on as it is instead of concatenating the pieces
    (``CatArrayBatchedCopy``: 40 launches, 0.79 ms of the 256 x 256 step in profiles/round2_d_kernel_trace_timed_region.md)."""

    __slots__ = ("arena", "col0", "width", "claimed", "dim")

    def __init__(self, arena, col0, width, dim=-1):
        self.arena, self.col0, self.width, self.claimed, self.dim = arena, col0, width, False, dim

    def view(self):
        return self.arena.buffer().narrow(self.dim, self.col0, self.width)


class _GradArena:
    def __init__(self, shape, device):
        self.shape, self.device, self.buf = tuple(shape), device, None

    def buffer(self):
        if self.buf is None:
            self.buf = torch.empty(self.shape, device=self.device, dtype=torch.float32)
        return self.buf


def _claim(t):
    """The gradient slot of a ``split_cols`` piece, if it has one nobody claimed yet (a piece feeding two kernels: the second takes
    the ordinary path and autograd's sum of the two gradients is copied into the slot)."""
    slot = getattr(t, "_mlagg_slot", None)
    if slot is None or slot.claimed or not torch.is_grad_enabled():
        return None
    slot.claimed = True
    return slot


def _grad_out(slot, shape, device):
    """(tensor, row stride) a backward kernel writes d(input) into: the claimed arena slot or a fresh contiguous tensor."""
    if slot is not None:
        v = slot.view()
        return v, v.stride(-2)
    t = torch.empty(shape, device=device, dtype=torch.float32)
    return t, shape[-1]


class SplitColsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, arena, *sizes):
        ctx.arena, ctx.sizes = arena, sizes
        return t.split(list(sizes), dim=-1)

    @staticmethod
    def backward(ctx, *grads):
        arena, sizes = ctx.arena, ctx.sizes
        buf = arena.buffer()
        col0 = 0
        for g, w in zip(grads, sizes):
            dst = buf[..., col0:col0 + w]
            if g is None:
                dst.zero_()
            elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                dst.copy_(g)
            col0 += w
        arena.buf = None
        return (buf, None) + (None,) * len(sizes)


class SplitPlanesFn(torch.autograd.Function):
    """``t.split(sizes, dim=1)`` of an NCHW map with ONE gradient buffer for the pieces (see split_planes)."""

    @staticmethod
    def forward(ctx, t, arena, *sizes):
        ctx.arena, ctx.sizes = arena, sizes
        return t.split(list(sizes), dim=1)

    @staticmethod
    def backward(ctx, *grads):
        arena, sizes = ctx.arena, ctx.sizes
        buf = arena.buffer()
        c0 = 0
        for g, w in zip(grads, sizes):
            dst = buf.narrow(1, c0, w)
            if g is None:
                dst.zero_()
            elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                dst.copy_(g)
            c0 += w
        arena.buf = None
        return (buf, None) + (None,) * len(sizes)


GRAD_ARENA = _os.environ.get("MLAGG_GRAD_ARENA", "1") == "1"
PLANE_ARENA = _os.environ.get("MLAGG_PLANE_ARENA", "1") == "1"


def split_planes(t, sizes):
    """``t.split(sizes, dim=1)`` of an NCHW map whose pieces feed this package's kernels (the (mamba | conv) halves of the MSMM inputs,
    MambaSkip.py:727-733): the same views, and the kernels' backward passes -- the token transpose, K19's data gradient -- write their
    results into ONE (B, C, H, W) gradient buffer, which the split's backward hands on instead of concatenating the pieces
    (``CatArrayBatchedCopy`` behind SplitWithSizesBackward: 163 us of the step).  A piece whose consumer cannot is copied into place."""
    if not (PLANE_ARENA and GRAD_ARENA and t.is_cuda and t.requires_grad and torch.is_grad_enabled() and t.dim() == 4 and t.is_contiguous()
            and t.dtype == torch.float32 and (t.shape[2] * t.shape[3]) % 4 == 0):
        return t.split(list(sizes), dim=1)
    arena = _GradArena(t.shape, t.device)
    pieces = SplitPlanesFn.apply(t, arena, *sizes)
    c0 = 0
    for p_, w in zip(pieces, sizes):
        p_._mlagg_slot = _GradSlot(arena, c0, w, dim=1)
        c0 += w
    return pieces


def transpose_2d_into(src, dst):
    """(B, R, C) -> dst (B, C, R...) whose samples are dense (C, R) blocks at any sample stride (a channel slice of an NCHW map)."""
    _require(src, "src")
    B, R, C = src.shape
    if not (src.stride(2) == 1 and src.stride(1) == C and src.stride(0) >= R * C and src.data_ptr() % 16 == 0
            and (C % 4 or src.stride(0) % 4 == 0)):
        src = src.contiguous()
    _lib.check(_lib.lib().mlagg_transpose_2d_into(_ptr(src), src.stride(0), _ptr(dst), dst.stride(0), B, R, C, _stream()),
               "mlagg_transpose_2d_into")
    return dst

FUSED_RESIDUAL_NORM = _os.environ.get("MLAGG_FUSED_RESIDUAL_NORM", "1") == "1"


def split_cols(t, sizes):
    """``t.split(sizes, dim=-1)`` of a fresh (B, N, total) projection output whose pieces feed this package's kernels: the pieces are
    the same strided views, and the kernels' backward passes write into ONE gradient buffer (see _GradSlot)."""
    if not (GRAD_ARENA and t.is_cuda and t.requires_grad and torch.is_grad_enabled() and t.dim() == 3 and t.is_contiguous()
            and all(w % 4 == 0 for w in sizes)):
        return t.split(list(sizes), dim=-1)
    arena = _GradArena(t.shape, t.device)
    pieces = SplitColsFn.apply(t, arena, *sizes)
    col0 = 0
    for p_, w in zip(pieces, sizes):
        p_._mlagg_slot = _GradSlot(arena, col0, w)
        col0 += w
    return pieces


class DWConv3x3Fn(torch.autograd.Function):
    """K2: depthwise 3x3 (+bias, optional SiLU) on token-major maps."""

    @staticmethod
    def forward(ctx, x, weight, bias, H, W, silu, slot=None, res=None):
        ctx.slot = slot
        x, xs = _rows(x, "x")
        B, N, C = x.shape
        if N != H * W:
            raise RuntimeError(f"dwconv3x3: {N} tokens != {H}x{W}")
        if res is not None:
            if silu or tuple(res.shape) != (B, N, C):
                raise RuntimeError("dwconv3x3: the residual is added to the plain convolution, same shape as the output")
            res = _require(res.contiguous(), "res")
        w = _require(weight.reshape(C, 9).contiguous(), "weight")
        y = torch.empty(B, N, C, device=x.device, dtype=torch.float32)
        pre = torch.empty_like(y) if silu else None      # pre-activation, needed by SiLU's backward
        _lib.check(_lib.lib().mlagg_dwconv3x3_fwd(_ptr(x), xs, _ptr(w), _ptr(bias), _ptr(res), _ptr(y), C, _ptr(pre), B, H, W, C,
                                                  int(silu), _stream()), "mlagg_dwconv3x3_fwd")
        ctx.save_for_backward(x, w, pre)
        ctx.geom = (H, W, bool(silu), bias is not None, weight.shape)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, pre = ctx.saved_tensors
        H, W, silu, has_bias, wshape = ctx.geom
        B, N, C = x.shape
        dy, dys = _rows(dy, "dy")
        dx, dxs = _grad_out(ctx.slot, (B, N, C), x.device)
        dw = torch.empty(C, 9, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32) if has_bias else None
        lib = _lib.lib()
        ws = torch.empty(lib.mlagg_dwconv3x3_bwd_workspace_floats(B, H, W, C), device=x.device, dtype=torch.float32)
        _lib.check(lib.mlagg_dwconv3x3_bwd(_ptr(x), x.stride(1), _ptr(w), _ptr(dy), dys, _ptr(pre), _ptr(dx), dxs,
                                           _ptr(dw), _ptr(db), _ptr(ws), B, H, W, C, int(silu), _stream()),
                   "mlagg_dwconv3x3_bwd")
        return dx, dw.reshape(wshape), db, None, None, None, None, (dy if ctx.has_res else None)


class DWConvGatedFn(torch.autograd.Function):
    """K2, gated form: y = SiLU(dwconv3x3(x) + bias) * gate -- ConvolutionalGLU's ``self.act(self.dwconv(x, H, W)) * v`` (MambaSkip.py:
    559-577) with the product in the convolution's epilogue; backward writes d(gate) from the weight-gradient pass (three ATen
    multiplications per scale and step gone)."""

    @staticmethod
    def forward(ctx, x, gate, weight, bias, H, W, slot=None, gate_slot=None):
        ctx.slots = (slot, gate_slot)
        x, xs = _rows(x, "x")
        gate, gs = _rows(gate, "gate")
        B, N, C = x.shape
        if N != H * W or tuple(gate.shape) != (B, N, C):
            raise RuntimeError(f"dwconv3x3_gated: bad shapes x {tuple(x.shape)} gate {tuple(gate.shape)} map {H}x{W}")
        w = _require(weight.reshape(C, 9).contiguous(), "weight")
        y = torch.empty(B, N, C, device=x.device, dtype=torch.float32)
        pre = torch.empty_like(y)
        _lib.check(_lib.lib().mlagg_dwconv3x3_gated_fwd(_ptr(x), xs, _ptr(w), _ptr(bias), _ptr(gate), gs, _ptr(y), C, _ptr(pre), B, H, W, C,
                                                        _stream()), "mlagg_dwconv3x3_gated_fwd")
        ctx.save_for_backward(x, gate, w, pre)
        ctx.geom = (H, W, bias is not None, weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gate, w, pre = ctx.saved_tensors
        H, W, has_bias, wshape = ctx.geom
        B, N, C = x.shape
        dy, dys = _rows(dy, "dy")
        dx, dxs = _grad_out(ctx.slots[0], (B, N, C), x.device)
        dgate, dgs = _grad_out(ctx.slots[1], (B, N, C), x.device)
        dw = torch.empty(C, 9, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32) if has_bias else None
        lib = _lib.lib()
        ws = torch.empty(lib.mlagg_dwconv3x3_bwd_workspace_floats(B, H, W, C), device=x.device, dtype=torch.float32)
        _lib.check(lib.mlagg_dwconv3x3_gated_bwd(_ptr(x), x.stride(1), _ptr(w), _ptr(dy), dys, _ptr(pre), _ptr(gate), gate.stride(1),
                                                 _ptr(dx), dxs, _ptr(dgate), dgs, _ptr(dw), _ptr(db), _ptr(ws), B, H, W, C, _stream()),
                   "mlagg_dwconv3x3_gated_bwd")
        return dx, dgate, dw.reshape(wshape), db, None, None, None, None


def dwconv3x3_gated(x, gate, weight, bias, H, W):
    """SiLU(depthwise 3x3(x) + bias) * gate on token-major maps (x, gate: column blocks of one projection output are fine)."""
    return DWConvGatedFn.apply(x, gate, weight, bias, H, W, _claim(x), _claim(gate))


def dwconv3x3_nlc(x, weight, bias, H, W, silu=False, res=None):
    """Depthwise 3x3 on a token-major map; `res` (same shape as the output) is added in the same pass."""
    return DWConv3x3Fn.apply(x, weight, bias, H, W, silu, _claim(x), res)


class DWConv3dFn(torch.autograd.Function):
    """K2v: depthwise 3x3x3 convolution (+ SiLU) on token-major volumes (B, D*H*W, C); weight (C, 1, 3, 3, 3)."""

    @staticmethod
    def forward(ctx, x, weight, bias, dims, silu):
        D, H, W = dims
        x, xs = _rows(x, "x")
        B, L, C = x.shape
        if L != D * H * W or weight.numel() != C * 27:
            raise RuntimeError(f"dwconv3d: bad shapes x {tuple(x.shape)} weight {tuple(weight.shape)} dims {dims}")
        w = _require(weight.reshape(C, 27).contiguous(), "weight")
        bias = None if bias is None else _require(bias.contiguous(), "bias", (C,))
        y = torch.empty(B, L, C, device=x.device, dtype=torch.float32)
        pre = torch.empty_like(y) if silu else None
        _lib.check(_lib.lib().mlagg_dwconv3d_fwd(_ptr(x), xs, _ptr(w), _ptr(bias), _ptr(y), C, _ptr(pre), B, D, H, W, C,
                                                 int(silu), _stream()), "mlagg_dwconv3d_fwd")
        ctx.save_for_backward(x, w, pre)
        ctx.geom = (D, H, W, bool(silu), bias is not None, weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, pre = ctx.saved_tensors
        D, H, W, silu, has_bias, wshape = ctx.geom
        B, L, C = x.shape
        dy, dys = _rows(dy, "dy")
        dx = torch.empty(B, L, C, device=x.device, dtype=torch.float32)
        dw = torch.empty(C, 27, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32) if has_bias else None
        lib = _lib.lib()
        ws = torch.empty(lib.mlagg_dwconv3d_bwd_workspace_floats(B, D, H, W, C), device=x.device, dtype=torch.float32)
        _lib.check(lib.mlagg_dwconv3d_bwd(_ptr(x), x.stride(1), _ptr(w), _ptr(dy), dys, _ptr(pre), _ptr(dx), C, _ptr(dw), _ptr(db),
                                          _ptr(ws), B, D, H, W, C, int(silu), _stream()), "mlagg_dwconv3d_bwd")
        return dx, dw.reshape(wshape), db, None, None


def dwconv3d_nlc(x, weight, bias, dims, silu=False):
    return DWConv3dFn.apply(x, weight, bias, dims, silu)


class SelectiveScan1Fn(torch.autograd.Function):
    """K1s: the d_state = 1 selective scan of the 3-D network (UMambaEnc_SS3D.py:244-296 + the 12-way sum at :338) on the
    token-major volume: u is gathered and y scattered through the permutation table inside the kernels.
    tok (B, L, C), idx (K, L) int32, dtr (B, K, R, L), Bs / Cs (B, K, L), Wdt (K*C, R), A / D / bias (K*C) -> (B, L, C)."""

    @staticmethod
    def forward(ctx, tok, idx, dtr, Bs, Cs, Wdt, A, D, bias):
        tok, ts = _rows(tok, "tok")
        B, L, C = tok.shape
        K, R = idx.shape[0], dtr.shape[2]
        if idx.dtype != torch.int32 or tuple(idx.shape) != (K, L) or not idx.is_cuda:
            raise RuntimeError("selective_scan1: idx must be an int32 (K, L) device table")
        dtr = _require(dtr.contiguous(), "dtr", (B, K, R, L))
        Bs = _require(Bs.contiguous(), "Bs", (B, K, L))
        Cs = _require(Cs.contiguous(), "Cs", (B, K, L))
        Wdt = _require(Wdt.contiguous(), "Wdt", (K * C, R))
        A, D, bias = (_require(v.reshape(-1).contiguous(), n, (K * C,)) for v, n in ((A, "A"), (D, "D"), (bias, "delta_bias")))
        lib = _lib.lib()
        yk = torch.empty(B, L, K * C, device=tok.device, dtype=torch.float32)
        state = torch.empty(lib.mlagg_selscan1_state_floats(B, L, C, K), device=tok.device, dtype=torch.float32)
        _lib.check(lib.mlagg_selscan1_fwd(_ptr(tok), ts, _ptr(idx), _ptr(dtr), _ptr(Bs), _ptr(Cs), _ptr(Wdt), R, _ptr(A), _ptr(D),
                                          _ptr(bias), _ptr(yk), _ptr(state), B, L, C, K, _stream()), "mlagg_selscan1_fwd")
        y = torch.empty(B, L, C, device=tok.device, dtype=torch.float32)
        _lib.check(lib.mlagg_block_sum(_ptr(yk), _ptr(y), B * L, K, C, _stream()), "mlagg_block_sum")
        ctx.save_for_backward(tok, idx, dtr, Bs, Cs, Wdt, A, D, bias, state)
        return y

    @staticmethod
    def backward(ctx, dy):
        tok, idx, dtr, Bs, Cs, Wdt, A, D, bias, state = ctx.saved_tensors
        B, L, C = tok.shape
        K, R = idx.shape[0], dtr.shape[2]
        dy, ds = _rows(dy, "dy")
        lib = _lib.lib()
        dev = tok.device
        duk = torch.empty(B, L, K * C, device=dev, dtype=torch.float32)
        ddtr, dBs, dCs = torch.empty_like(dtr), torch.empty_like(Bs), torch.empty_like(Cs)
        dpar = torch.empty(K * C, 3 + R, device=dev, dtype=torch.float32)
        ws = torch.empty(lib.mlagg_selscan1_bwd_workspace_floats(B, L, C, K, R), device=dev, dtype=torch.float32)
        _lib.check(lib.mlagg_selscan1_bwd(_ptr(tok), tok.stride(1), _ptr(idx), _ptr(dtr), _ptr(Bs), _ptr(Cs), _ptr(Wdt), R, _ptr(A),
                                          _ptr(D), _ptr(bias), _ptr(dy), ds, _ptr(state), _ptr(duk), _ptr(ddtr), _ptr(dBs), _ptr(dCs),
                                          _ptr(dpar), _ptr(ws), B, L, C, K, _stream()), "mlagg_selscan1_bwd")
        dtok = torch.empty(B, L, C, device=dev, dtype=torch.float32)
        _lib.check(lib.mlagg_block_sum(_ptr(duk), _ptr(dtok), B * L, K, C, _stream()), "mlagg_block_sum")
        return dtok, None, ddtr, dBs, dCs, dpar[:, 3:], dpar[:, 0], dpar[:, 1], dpar[:, 2]


def selective_scan1(tok, idx, dtr, Bs, Cs, Wdt, A, D, bias):
    return SelectiveScan1Fn.apply(tok, idx, dtr, Bs, Cs, Wdt, A, D, bias)


class LocalDiffAttnFn(torch.autograd.Function):
    """K3: fused 3x3-window differential attention + RMSNorm + LePE (AggregatedAttention local branch)."""

    @staticmethod
    def forward(ctx, q, kv, lam, subln_w, lepe_w, lepe_b, H, W, nh, scale, q_slot=None, kv_slot=None):
        ctx.slots = (q_slot, kv_slot)
        q, qs = _rows(q, "q")
        kv, kvs = _rows(kv, "kv")
        B, N, d = q.shape
        if N != H * W or d != nh * 48 or kv.shape[2] != 2 * d:
            raise RuntimeError(f"local_diff_attn: bad shapes q {tuple(q.shape)} kv {tuple(kv.shape)} H {H} W {W} nh {nh}")
        lam = _require(lam.reshape(1).contiguous(), "lambda")
        subln_w = _require(subln_w.contiguous(), "subln.weight", (48,))
        lw = _require(lepe_w.reshape(d, 9).contiguous(), "lepe.weight")
        lb = _require(lepe_b.contiguous(), "lepe.bias", (d,))
        out = torch.empty(B, N, d, device=q.device, dtype=torch.float32)
        _lib.check(_lib.lib().mlagg_local_attn_fwd(_ptr(q), qs, _ptr(kv), kvs, _ptr(lam), _ptr(subln_w), _ptr(lw),
                                                   _ptr(lb), _ptr(out), d, B, H, W, nh, float(scale), _stream()),
                   "mlagg_local_attn_fwd")
        ctx.save_for_backward(q, kv, lam, subln_w, lw)
        ctx.geom = (H, W, nh, float(scale), lepe_w.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, lam, subln_w, lw = ctx.saved_tensors
        H, W, nh, scale, lwshape = ctx.geom
        B, N, d = q.shape
        dout, dos = _rows(dout, "dout")
        lib = _lib.lib()
        dq, dqs = _grad_out(ctx.slots[0], (B, N, d), q.device)
        dkv, dkvs = _grad_out(ctx.slots[1], (B, N, 2 * d), q.device)
        small = torch.zeros(1 + 48 + d * 9 + d, device=q.device, dtype=torch.float32)
        dlam, dsub, dlw, dlb = small[:1], small[1:49], small[49:49 + d * 9], small[49 + d * 9:]
        ws = torch.empty(lib.mlagg_local_attn_bwd_workspace_floats(B, H, W, nh), device=q.device, dtype=torch.float32)
        _lib.check(lib.mlagg_local_attn_bwd(_ptr(q), q.stride(1), _ptr(kv), kv.stride(1), _ptr(lam), _ptr(subln_w),
                                            _ptr(lw), _ptr(dout), dos, _ptr(dq), dqs, _ptr(dkv), dkvs, _ptr(dlam),
                                            _ptr(dsub), _ptr(dlw), _ptr(dlb), _ptr(ws), B, H, W, nh, scale, _stream()),
                   "mlagg_local_attn_bwd")
        return dq, dkv, dlam.reshape(()), dsub, dlw.reshape(lwshape), dlb, None, None, None, None, None, None


class PooledDiffAttnFn(torch.autograd.Function):
    """K4: fused pooled differential attention + RMSNorm (AggregatedAttention global branch)."""

    @staticmethod
    def forward(ctx, q, kp, vp, lam, subln_w, nh, scale, q_slot=None):
        ctx.slot = q_slot
        q, qs = _rows(q, "q")
        kp, kps = _rows(kp, "k_pool")
        vp, vps = _rows(vp, "v_pool")
        B, N, d = q.shape
        P = kp.shape[1]
        if d != nh * 48 or tuple(kp.shape) != (B, P, d) or tuple(vp.shape) != (B, P, d):
            raise RuntimeError(f"pooled_diff_attn: bad shapes q {tuple(q.shape)} k {tuple(kp.shape)} v {tuple(vp.shape)}")
        lam = _require(lam.reshape(1).contiguous(), "lambda")
        subln_w = _require(subln_w.contiguous(), "subln.weight", (48,))
        out = torch.empty(B, N, d, device=q.device, dtype=torch.float32)
        need = any(ctx.needs_input_grad)      # grad mode is off inside Function.forward
        lse = torch.empty(B, N, nh, 2, device=q.device, dtype=torch.float32) if need else None
        o_pre = torch.empty(B, N, d, device=q.device, dtype=torch.float32) if need else None
        _lib.check(_lib.lib().mlagg_pooled_attn_fwd(_ptr(q), qs, _ptr(kp), kps, _ptr(vp), vps, _ptr(lam), _ptr(subln_w),
                                                    _ptr(out), d, _ptr(lse), _ptr(o_pre), B, N, P, nh, float(scale),
                                                    _stream()), "mlagg_pooled_attn_fwd")
        ctx.save_for_backward(q, kp, vp, lam, subln_w, lse, o_pre)
        ctx.geom = (nh, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kp, vp, lam, subln_w, lse, o_pre = ctx.saved_tensors
        nh, scale = ctx.geom
        B, N, d = q.shape
        P = kp.shape[1]
        dout, dos = _rows(dout, "dout")
        lib = _lib.lib()
        dq, dqs = _grad_out(ctx.slot, (B, N, d), q.device)
        dkp = torch.empty(B, P, d, device=q.device, dtype=torch.float32)
        dvp = torch.empty(B, P, d, device=q.device, dtype=torch.float32)
        small = torch.zeros(1 + 48, device=q.device, dtype=torch.float32)
        ws = torch.empty(lib.mlagg_pooled_attn_bwd_workspace_floats(B, N, P, nh), device=q.device, dtype=torch.float32)
        _lib.check(lib.mlagg_pooled_attn_bwd(_ptr(q), q.stride(1), _ptr(kp), kp.stride(1), _ptr(vp), vp.stride(1),
                                             _ptr(lam), _ptr(subln_w), _ptr(dout), dos, _ptr(lse), _ptr(o_pre),
                                             _ptr(dq), dqs, _ptr(dkp), d, _ptr(dvp), d, _ptr(small[:1]), _ptr(small[1:]),
                                             _ptr(ws), B, N, P, nh, scale, _stream()), "mlagg_pooled_attn_bwd")
        return dq, dkp, dvp, small[0].reshape(()), small[1:], None, None, None, None


class PooledDiffAttnLpFn(torch.autograd.Function):
    """K4lp: the pooled differential attention on the 16-bit matrix cores (csrc/pooled_attn_lp.hip) -- the 16-bit modes' form of K4:
    q * scale, k, v, the softmax weights and d(o) are bf16 / fp16 MFMA operands (what the reference's four flash_attn_func calls see,
    nnUNetTrainer_MLAgg_2D_dt_MS.py:733-751), sums / softmax / RMSNorm and every tensor in memory fp32."""

    @staticmethod
    def forward(ctx, q, kp, vp, lam, subln_w, nh, scale, cdt, q_slot=None):
        ctx.slot = q_slot
        q, qs = _rows(q, "q")
        kp, kps = _rows(kp, "k_pool")
        vp, vps = _rows(vp, "v_pool")
        B, N, d = q.shape
        P = kp.shape[1]
        if d != nh * 48 or tuple(kp.shape) != (B, P, d) or tuple(vp.shape) != (B, P, d):
            raise RuntimeError(f"pooled_diff_attn: bad shapes q {tuple(q.shape)} k {tuple(kp.shape)} v {tuple(vp.shape)}")
        lam = _require(lam.reshape(1).contiguous(), "lambda")
        subln_w = _require(subln_w.contiguous(), "subln.weight", (48,))
        out = torch.empty(B, N, d, device=q.device, dtype=torch.float32)
        need = any(ctx.needs_input_grad)
        lse = torch.empty(B, N, nh, 2, device=q.device, dtype=torch.float32) if need else None
        o12 = torch.empty(2, B, N, d, device=q.device, dtype=torch.float32) if need else None
        _lib.check(_lib.lib().mlagg_pooled_attn_lp_fwd(_ptr(q), qs, _ptr(kp), kps, _ptr(vp), vps, _ptr(lam), _ptr(subln_w), _ptr(out), d,
                                                       _ptr(lse), _ptr(o12[0]) if need else None, _ptr(o12[1]) if need else None, B, N, P,
                                                       nh, float(scale), _LP_CODE[cdt], _stream()), "mlagg_pooled_attn_lp_fwd")
        ctx.save_for_backward(q, kp, vp, lam, subln_w, lse, o12)
        ctx.geom = (nh, float(scale), cdt)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kp, vp, lam, subln_w, lse, o12 = ctx.saved_tensors
        nh, scale, cdt = ctx.geom
        B, N, d = q.shape
        P = kp.shape[1]
        dout, dos = _rows(dout, "dout")
        lib = _lib.lib()
        dq, dqs = _grad_out(ctx.slot, (B, N, d), q.device)
        dkp = torch.empty(B, P, d, device=q.device, dtype=torch.float32)
        dvp = torch.empty(B, P, d, device=q.device, dtype=torch.float32)
        small = torch.empty(1 + 48, device=q.device, dtype=torch.float32)
        ws = torch.empty(lib.mlagg_pooled_attn_lp_bwd_workspace_floats(B, N, P, nh), device=q.device, dtype=torch.float32)
        _lib.check(lib.mlagg_pooled_attn_lp_bwd(_ptr(q), q.stride(1), _ptr(kp), kp.stride(1), _ptr(vp), vp.stride(1), _ptr(lam),
                                                _ptr(subln_w), _ptr(dout), dos, _ptr(lse), _ptr(o12[0]), _ptr(o12[1]), _ptr(dq), dqs,
                                                _ptr(dkp), d, _ptr(dvp), d, _ptr(small[:1]), _ptr(small[1:]), _ptr(ws), B, N, P, nh, scale,
                                                _LP_CODE[cdt], _stream()), "mlagg_pooled_attn_lp_bwd")
        return dq, dkp, dvp, small[0].reshape(()), small[1:], None, None, None, None


class FlashAttnFn(torch.autograd.Function):
    """Boundary #3: softmax(q k^T scale) v on 16-bit (B, N, nh, 24) / (B, P, nh, 24) tensors, fp32 arithmetic."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        if not (q.is_cuda and q.dtype in _LP_CODE and k.dtype == q.dtype and v.dtype == q.dtype):
            raise RuntimeError("flash_attn_func: fp16 / bf16 tensors on the MI355X device expected "
                               f"(got {q.dtype}, {k.dtype}, {v.dtype} on {q.device})")
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, N, nh, e = q.shape
        P = k.shape[1]
        if tuple(k.shape) != (B, P, nh, e) or tuple(v.shape) != (B, P, nh, e):
            raise RuntimeError(f"flash_attn_func: bad shapes q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)}")
        out = torch.empty_like(q)
        need = any(ctx.needs_input_grad[:3])
        lse = torch.empty(B, nh, N, device=q.device, dtype=torch.float32) if need else None
        _lib.check(_lib.lib().mlagg_flash_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), B, N, P, nh, e, float(scale),
                                                   _LP_CODE[q.dtype], _stream()), "mlagg_flash_attn_fwd")
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        B, N, nh, e = q.shape
        P = k.shape[1]
        dout = dout.contiguous().to(q.dtype)
        lib = _lib.lib()
        dq = torch.empty_like(q)
        ws = torch.empty(lib.mlagg_flash_attn_bwd_workspace_floats(B, N, P, nh, e), device=q.device, dtype=torch.float32)
        _lib.check(lib.mlagg_flash_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dq), _ptr(ws),
                                            B, N, P, nh, e, ctx.scale, _LP_CODE[q.dtype], _stream()), "mlagg_flash_attn_bwd")
        dkv = ws[B * nh * N:].view(B, P, nh, 2, e)
        return dq, dkv[:, :, :, 0].to(q.dtype), dkv[:, :, :, 1].to(q.dtype), None


def flash_attn(q, k, v, softmax_scale=None):
    return FlashAttnFn.apply(q, k, v, q.shape[-1] ** -0.5 if softmax_scale is None else softmax_scale)


def local_diff_attn(q, kv, lam, subln_w, lepe_w, lepe_b, H, W, nh, scale):
    return LocalDiffAttnFn.apply(q, kv, lam, subln_w, lepe_w, lepe_b, H, W, nh, scale, _claim(q), _claim(kv))


K4_LP = _os.environ.get("MLAGG_K4_LP", "1") == "1"


def pooled_diff_attn(q, k_pool, v_pool, lam, subln_w, nh, scale):
    cdt = compute_dtype()
    if cdt != torch.float32 and K4_LP and k_pool.shape[1] <= 320:
        return PooledDiffAttnLpFn.apply(q, k_pool, v_pool, lam, subln_w, nh, scale, cdt, _claim(q))
    return PooledDiffAttnFn.apply(q, k_pool, v_pool, lam, subln_w, nh, scale, _claim(q))


# fp32 projections on the 16-bit matrix instructions (csrc/linear_lp.hip MODE 2: each fp32 operand as three bf16 pieces, six partial
# products, fp32 accumulation -- as accurate against float64 as the fp32 instruction, tools/bench_linear.py).  MLAGG_K5_X3=0: K5 on
# v_mfma_f32_32x32x2_f32.
K5_X3 = _os.environ.get("MLAGG_K5_X3", "1") == "1"
WGRAD_MIN_ROWS = int(_os.environ.get("MLAGG_WGRAD_MIN_ROWS", "0"))      # 0: by the state of the library (below); a number forces it
# K5w against the library's weight-gradient GEMM at short token counts (2 560): with the TUNED table loaded (gemm_tuning: the headline
# configuration) the library ties (35.88 vs 35.89 ms), so K5w starts at 8192 tokens; with the default heuristics (every other configuration)
# K5w wins 0.4 ms of the 224 x 224 bf16 step (profiles/round4_n_lp_k5_min_rows_ab.log) and starts at 2048
GEMM_TABLE_LOADED = [False]


def wgrad_min_rows():
    return WGRAD_MIN_ROWS if WGRAD_MIN_ROWS > 0 else (8192 if GEMM_TABLE_LOADED[0] else 2048)
# 16-bit modes: the one-product K5 from this many tokens on (2048 measured slower than the library's 16-bit GEMM: 28.81 vs 28.62 ms on config 3,
# profiles/round4_n_lp_k5_min_rows_ab.log)
LP_K5_MIN_ROWS = int(_os.environ.get("MLAGG_LP_K5_MIN_ROWS", "8192"))
K5_MIN_ROWS = int(_os.environ.get("MLAGG_K5_MIN_ROWS", "16384"))     # fp32 forward / dx: K5 from this many tokens on (at 10240 tokens the
#                            library's split-K kernels win: 80-320 K5 workgroups do not fill 256 CUs evenly; A/B on the step: +0.9 %)


def _rows2d(t, name):
    """(..., C) tensor -> (M, C) view with unit inner stride and one row stride; copies only if it must."""
    _require(t, name)
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(1) != 1:
        t2 = t2.contiguous()
    return t2, t2.stride(0)


def _mfma_rows(t, name):
    """(..., C) -> (M, C) view usable by the MFMA projection kernels (unit inner stride, 16-byte aligned rows)."""
    t2, ts = _rows2d(t, name)
    if ts % 4 or t2.data_ptr() % 16:
        t2 = t2.contiguous()
        ts = t2.shape[1]
    return t2, ts


# ------------------------------------------------------------------------------------------------
# K5, round-4 form (csrc/linear_x3.hip): the weight operand is an IMAGE (its three bf16 pieces, laid out for the kernel; the same for
# W^T for the data gradient), built once per step for every projection of a network in ONE launch (WeightImageSet) or, for a weight
# nobody registered, on the fly.  MLAGG_K5_V2=0: the round-3 kernels + the library GEMM for short token counts.
# ------------------------------------------------------------------------------------------------
K5_V2 = _os.environ.get("MLAGG_K5_V2", "1") == "1"
# end of round 4: stages 2 / 3 (10 240 / 2 560 tokens) too -- against the TUNED library GEMMs the kernel wins 11 of 16 products there
# (tools/bench_linear_x3.py with BENCH_TUNED_GEMM=1) and the step 0.1-0.3 ms on two boxes (profiles/round4_m_x3_min_rows_ab.log); mid-round,
# before the fused Mlp epilogues and the per-network image set, the same switch had lost 0.2 ms
X3_MIN_ROWS = int(_os.environ.get("MLAGG_X3_MIN_ROWS", "2048"))
_IMAGE_EPOCH = [0]              # bumped by whatever rewrites parameters behind autograd's back (ClipAdamW's raw-pointer update)


def invalidate_weight_images():
    _IMAGE_EPOCH[0] += 1


def image_epoch():
    return _IMAGE_EPOCH[0]


def _image_pair(w):
    """(img, imgT) of a contiguous fp32 (N, K) matrix, built now (one launch)."""
    N, K = w.shape
    lib = _lib.lib()
    img = torch.empty(lib.mlagg_weight_image_bytes(N, K), dtype=torch.uint8, device=w.device)
    imgT = torch.empty(lib.mlagg_weight_image_bytes(K, N), dtype=torch.uint8, device=w.device)
    _lib.check(lib.mlagg_weight_image(_ptr(w), K, _ptr(img), _ptr(imgT), N, K, _stream()), "mlagg_weight_image")
    return img, imgT


class WeightImageSet:
    """The weight images of one network.  The first forward pass under ``with images:`` records which persistent matrices (parameters
    and stacked-weight buffers) the projections ask for; from then on ``begin`` rebuilds all their images with one launch over a
    device table, and lookups are a dictionary hit checked against the matrix's version counter (an in-place change since the build
    -- a stack refreshed again, a loaded checkpoint -- falls back to building that one image on the fly)."""

    active = None

    def __init__(self):
        self.tensors, self.entries, self.table, self.store, self.ptrs, self.max_tiles, self.built = [], {}, None, None, None, 0, None

    def _rebuild_table(self):
        import numpy as np
        lib = _lib.lib()
        dev = self.tensors[0].device
        sizes = [(lib.mlagg_weight_image_bytes(*t.shape), lib.mlagg_weight_image_bytes(t.shape[1], t.shape[0])) for t in self.tensors]
        offs, total = [], 0
        for a, b in sizes:
            offs.append((total, total + ((a + 255) & ~255)))
            total += ((a + 255) & ~255) + ((b + 255) & ~255)
        self.store = torch.empty(total, dtype=torch.uint8, device=dev)
        base = self.store.data_ptr()
        jobs = np.zeros(len(self.tensors), dtype=np.dtype([("w", "<u8"), ("img", "<u8"), ("imgT", "<u8"), ("N", "<i4"), ("K", "<i4"),
                                                           ("ws", "<i4"), ("pad", "<i4")]))
        self.views, self.max_tiles = [], 0
        for i, (t, (oa, ob), (sa, sb)) in enumerate(zip(self.tensors, offs, sizes)):
            N, K = t.shape
            jobs[i] = (t.data_ptr(), base + oa, base + ob, N, K, K, 0)
            self.views.append((self.store[oa:oa + sa], self.store[ob:ob + sb]))
            self.max_tiles = max(self.max_tiles, ((N + 31) // 32) * ((K + 31) // 32))
        self.table = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
        self.ptrs = tuple(t.data_ptr() for t in self.tensors)

    def begin(self):
        WeightImageSet.active = self
        if not self.tensors:
            return
        if self.table is None or self.ptrs != tuple(t.data_ptr() for t in self.tensors):
            self._rebuild_table()
            self.built = None
        ep = _IMAGE_EPOCH[0]
        sig = (ep,) + tuple(t._version for t in self.tensors)
        if sig == self.built and not torch.cuda.is_current_stream_capturing():
            return                                    # nothing changed since the last build (inference loops, a second forward of a step)
        _lib.check(_lib.lib().mlagg_weight_images(_ptr(self.table), len(self.tensors), self.max_tiles, _stream()), "mlagg_weight_images")
        self.built = sig
        self.entries = {t.data_ptr(): (t._version, ep, v) for t, v in zip(self.tensors, self.views)}

    def end(self):
        WeightImageSet.active = None

    def __enter__(self):
        self.begin()
        return self

    def __exit__(self, *exc):
        self.end()
        return False

    def lookup(self, w):
        e = self.entries.get(w.data_ptr())
        if e is not None and e[0] == w._version and e[1] == _IMAGE_EPOCH[0]:
            return e[2]
        pair = _image_pair(w)
        # what is worth keeping: a parameter, or the buffer behind a stacked-weight view.  Never the view itself: a tensor with a
        # grad_fn keeps the AccumulateGrad nodes of an earlier iteration alive, and autograd then runs them on the stream they were
        # created on -- inside a hipGraph capture that cross-stream hand-off is a segmentation fault
        keep = getattr(w, "_mlagg_buffer", w if isinstance(w, torch.nn.Parameter) else None)
        if keep is not None and keep.grad_fn is None and keep.dim() == 2 and keep.is_contiguous() and \
                keep.data_ptr() == w.data_ptr() and all(keep.data_ptr() != t.data_ptr() for t in self.tensors):
            self.tensors.append(keep)               # from the next forward on: part of the one-launch build
            self.table = None
        return pair


def weight_images(w):
    """(img, imgT) of the contiguous (N, K) matrix ``w`` that are current NOW."""
    s = WeightImageSet.active
    return _image_pair(w) if s is None else s.lookup(w)


def _x3_ok(M, N, K):
    return K5_V2 and M >= X3_MIN_ROWS and bool(_lib.lib().mlagg_linear_x3_supported(M, N, K))


def _x3(x2, xs, img, bias, M, N, K, epilogue=0, pre=None, pre_stride=0, out_shape=None, out=None, out_stride=None):
    """One launch of mlagg_linear_x3; returns y, or (pre-activation, activation) for the GELU epilogue.  ``out`` / ``out_stride``: a
    destination the caller owns (rows of out_stride floats: a column block of a wider buffer)."""
    y = out if out is not None else torch.empty(out_shape if out_shape is not None else (M, N), device=x2.device, dtype=torch.float32)
    ys = N if out_stride is None else out_stride
    act = torch.empty_like(y) if epilogue == 1 else None
    _flop("K5", 2 * M * N * K)
    _lib.check(_lib.lib().mlagg_linear_x3(_ptr(x2), xs, _ptr(img), _ptr(bias), _ptr(y), ys, _ptr(act), _ptr(pre), pre_stride, M, N, K,
                                          epilogue, _stream()), "mlagg_linear_x3")
    return y if epilogue != 1 else (y, act)


def _linear_wgrad(dy2, dys, x, O, I, has_bias):
    """dW (O, I) and db (O) of a token-major Linear: K5w for long token counts, the library GEMM + K8 column sums below."""
    M = dy2.shape[0]
    x2, xs = _rows2d(x, "x")
    lib = _lib.lib()
    if M >= wgrad_min_rows():
        # dW | db in one allocation (every entry is written by the reduction)
        buf = torch.empty(O * I + (O if has_bias else 0), device=dy2.device, dtype=torch.float32)
        dW = buf[:O * I].view(O, I)
        db = buf[O * I:] if has_bias else None
        ws = torch.empty(lib.mlagg_linear_wgrad_workspace_floats(M, O, I), device=dy2.device, dtype=torch.float32)
        wgrad = lib.mlagg_linear_wgrad_x3 if K5_X3 else lib.mlagg_linear_wgrad
        _flop("K5w", 2 * M * O * I)
        _lib.check(wgrad(_ptr(dy2), dys, _ptr(x2), xs, _ptr(dW), _ptr(db), _ptr(ws), M, O, I, _stream()), "mlagg_linear_wgrad")
        return dW, db
    dW = dy2.t().matmul(x2)
    db = (column_sum(dy2) if dy2.is_cuda else dy2.sum(0)) if has_bias else None
    return dW, db


class LinearFn(torch.autograd.Function):
    """y = x W^T + b for token-major activations: forward and dx on K5 (round-4 form on weight images at every token count from
    X3_MIN_ROWS up; MLAGG_K5_V2=0 / 16-bit modes: the round-3 kernels for long token counts, the library GEMM below), dW / db on K5w."""

    @staticmethod
    def forward(ctx, x, weight, bias, slot=None):
        ctx.save_for_backward(x, weight)
        ctx.slot = slot
        ctx.has_bias = bias is not None
        ctx.cdt = cdt = compute_dtype()
        ctx.imgT = None
        ctx.leaf = _leaf_ok(weight, bias)
        ctx.leaf_params = _leaf_sources(weight, bias)
        note_leaf_use(weight, bias)
        O, I = weight.shape
        M = x.numel() // I
        if cdt == torch.float32 and x.is_cuda and _x3_ok(M, O, I):
            x2, xs = _mfma_rows(x, "x")
            w = _require(weight.contiguous(), "weight")
            img, ctx.imgT = weight_images(w)
            return _x3(x2, xs, img, bias, M, O, I, out_shape=x.shape[:-1] + (O,))
        if M >= (K5_MIN_ROWS if cdt == torch.float32 else LP_K5_MIN_ROWS) and I % 4 == 0 and x.is_cuda:
            x2, xs = _mfma_rows(x, "x")
            w = _require(weight.contiguous(), "weight")
            y = torch.empty(x.shape[:-1] + (O,), device=x.device, dtype=torch.float32)
            _flop("K5", 2 * M * O * I)
            if cdt == torch.float32 and K5_X3:
                _lib.check(_lib.lib().mlagg_linear_lp_fwd(_ptr(x2), xs, _ptr(w), _ptr(bias), _ptr(y), O, M, O, I, _DTYPE_BF16X3,
                                                          _stream()), "mlagg_linear_lp_fwd")
            elif cdt == torch.float32:
                _lib.check(_lib.lib().mlagg_linear_fwd(_ptr(x2), xs, _ptr(w), _ptr(bias), _ptr(y), O, M, O, I, _stream()),
                           "mlagg_linear_fwd")
            else:
                _lib.check(_lib.lib().mlagg_linear_lp_fwd(_ptr(x2), xs, _ptr(w), _ptr(bias), _ptr(y), O, M, O, I,
                                                          _LP_CODE[cdt], _stream()), "mlagg_linear_lp_fwd")
            return y
        if cdt != torch.float32:
            return torch.nn.functional.linear(x.to(cdt), weight.to(cdt), lp(bias, cdt)).float()
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dW = db = None
        O, I = weight.shape
        cdt = ctx.cdt
        dy2, dys = _mfma_rows(dy, "dy")
        M = dy2.shape[0]
        lib = _lib.lib()
        if ctx.needs_input_grad[0]:
            if ctx.imgT is not None and _x3_ok(M, I, O):
                # dx = dy . W on the image of W^T built with the forward's image (no per-step transpose of the weight); a claimed
                # split_cols slot: written straight into the shared gradient buffer of the pieces
                if ctx.slot is not None:
                    dx, dxs = _grad_out(ctx.slot, x.shape, dy.device)
                    _x3(dy2, dys, ctx.imgT, None, M, I, O, out=dx, out_stride=dxs)
                else:
                    dx = _x3(dy2, dys, ctx.imgT, None, M, I, O, out_shape=x.shape)
            elif (M >= K5_MIN_ROWS if cdt == torch.float32 else M >= LP_K5_MIN_ROWS) and O % 4 == 0 and I % 4 == 0:
                w = _require(weight.contiguous(), "weight")
                dx = torch.empty(x.shape, device=dy.device, dtype=torch.float32)
                _flop("K5", 2 * M * O * I)
                if cdt == torch.float32 and K5_X3:
                    # dx = dy . W as the forward form of the kernel on W^T (I, O): its weight tile is then read along the
                    # contraction, the fast staging path (the transpose is a (O, I) copy of a few hundred KB)
                    wt = transpose_2d(w.unsqueeze(0))[0]
                    _lib.check(lib.mlagg_linear_lp_fwd(_ptr(dy2), dys, _ptr(wt), None, _ptr(dx), I, M, I, O, _DTYPE_BF16X3, _stream()),
                               "mlagg_linear_lp_fwd")
                elif cdt == torch.float32:
                    _lib.check(lib.mlagg_linear_dgrad(_ptr(dy2), dys, _ptr(w), _ptr(dx), I, M, O, I, _stream()),
                               "mlagg_linear_dgrad")
                else:
                    _lib.check(lib.mlagg_linear_lp_dgrad(_ptr(dy2), dys, _ptr(w), _ptr(dx), I, M, O, I, _LP_CODE[cdt],
                                                         _stream()), "mlagg_linear_lp_dgrad")
            elif cdt != torch.float32:
                dx = dy.to(cdt).matmul(weight.to(cdt)).float()
            else:
                dx = dy.matmul(weight)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            # weight / bias gradients stay fp32 in every mode (K5w: the token sum is the long one)
            with _LeafStream(dy2, x, ok=ctx.leaf and leaf_single_use(ctx.leaf_params)):
                dW, db = _linear_wgrad(dy2, dys, x, O, I, ctx.has_bias)
        return dx, dW, db, None


def linear(x, weight, bias=None):
    # a split_cols piece as input: its gradient is written in place when both products of this layer run on K5
    slot = None
    if getattr(x, "_mlagg_slot", None) is not None and x.is_cuda and compute_dtype() == torch.float32:
        O, I = weight.shape
        M = x.numel() // I
        if _x3_ok(M, O, I) and _x3_ok(M, I, O):
            slot = _claim(x)
    return LinearFn.apply(x, weight, bias, slot)


class MlpFn(torch.autograd.Function):
    """fc2(GELU(fc1(x))) of reference Mlp (T:176-192) as four K5 launches: fc1 writes the pre-activation AND its GELU, and in backward
    the data gradient of fc2 comes out already multiplied by GELU'(pre) -- the two elementwise GELU passes of the ATen form are gone."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        H, I = w1.shape
        O = w2.shape[0]
        M = x.numel() // I
        x2, xs = _mfma_rows(x, "x")
        w1c, w2c = _require(w1.contiguous(), "fc1.weight"), _require(w2.contiguous(), "fc2.weight")
        img1, img1T = weight_images(w1c)
        img2, img2T = weight_images(w2c)
        pre, act = _x3(x2, xs, img1, b1, M, H, I, epilogue=1)
        y = _x3(act, H, img2, b2, M, O, H, out_shape=x.shape[:-1] + (O,))
        ctx.save_for_backward(x, w1, w2, pre, act)
        ctx.leaf = _leaf_ok(w1, b1, w2, b2)
        ctx.leaf_params = _leaf_sources(w1, b1, w2, b2)
        note_leaf_use(w1, b1, w2, b2)
        ctx.images = (img1T, img2T)
        ctx.bias = (b1 is not None, b2 is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, pre, act = ctx.saved_tensors
        img1T, img2T = ctx.images
        H, I = w1.shape
        O = w2.shape[0]
        dy2, dys = _mfma_rows(dy, "dy")
        M = dy2.shape[0]
        ok = ctx.leaf and leaf_single_use(ctx.leaf_params)
        with _LeafStream(dy2, act, ok=ok):
            dW2, db2 = _linear_wgrad(dy2, dys, act, O, H, ctx.bias[1])
        dpre = _x3(dy2, dys, img2T, None, M, H, O, epilogue=2, pre=pre, pre_stride=H)          # (dy . W2) * GELU'(pre)
        with _LeafStream(dpre, x, ok=ok):
            dW1, db1 = _linear_wgrad(dpre, H, x, H, I, ctx.bias[0])
        dx = _x3(dpre, H, img1T, None, M, I, H, out_shape=x.shape) if ctx.needs_input_grad[0] else None
        return dx, dW1, db1, dW2, db2


def mlp_supported(x, w1, w2):
    H, I = w1.shape
    O = w2.shape[0]
    M = x.numel() // I
    return bool(compute_dtype() == torch.float32 and x.is_cuda and _x3_ok(M, H, I) and _x3_ok(M, O, H) and _x3_ok(M, I, H)
                and _x3_ok(M, H, O))


def mlp(x, w1, b1, w2, b2):
    return MlpFn.apply(x, w1, b1, w2, b2)


class LayerNormFn(torch.autograd.Function):
    """K6: LayerNorm over the last dimension (C in {48, 96, 192, 384, 768})."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        C = x.shape[-1]
        x2, xs = _rows2d(x, "x")
        if xs % 4 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
            xs = C
        rows = x2.shape[0]
        y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        stats = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().mlagg_layernorm_fwd(_ptr(x2), xs, _ptr(weight), _ptr(bias), _ptr(y), _ptr(stats), rows, C,
                                                  float(eps), _stream()), "mlagg_layernorm_fwd")
        ctx.save_for_backward(x2, weight, stats)
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, weight, stats = ctx.saved_tensors
        rows, C = x2.shape
        dy2, dys = _rows2d(dy, "dy")
        if dys % 4 or dy2.data_ptr() % 16:
            dy2 = dy2.contiguous()
            dys = C
        lib = _lib.lib()
        dx = torch.empty(ctx.xshape, device=dy.device, dtype=torch.float32)
        dg = torch.empty(C, device=dy.device, dtype=torch.float32)
        db = torch.empty(C, device=dy.device, dtype=torch.float32) if ctx.has_bias else None
        ws = torch.empty(lib.mlagg_layernorm_bwd_workspace_floats(rows, C), device=dy.device, dtype=torch.float32)
        _lib.check(lib.mlagg_layernorm_bwd(_ptr(x2), x2.stride(0), _ptr(dy2), dys, _ptr(weight), _ptr(stats), _ptr(dx),
                                           _ptr(dg), _ptr(db), _ptr(ws), rows, C, _stream()), "mlagg_layernorm_bwd")
        return dx, dg, db, None


def layer_norm(x, weight, bias, eps=1e-5):
    """LayerNorm over the last dimension: K6 for the channel counts of the MLAgg-UNet path, ATen otherwise."""
    if _lib.lib().mlagg_layernorm_supported(int(x.shape[-1])):
        return LayerNormFn.apply(x, weight, bias, eps)
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), weight, bias, eps)


class ResidualLayerNormFn(torch.autograd.Function):
    """K6 with the residual junction in front: (xsum, y) = (skip + branch * scale[sample], LayerNorm(xsum)); backward adds the
    gradient reaching xsum from its other consumers inside the LayerNorm-backward kernel (no separate add / scale kernels)."""

    @staticmethod
    def forward(ctx, skip, branch, scale, weight, bias, eps):
        skip = _require(skip.contiguous(), "skip")
        branch = _require(branch.contiguous(), "branch")
        if skip.shape != branch.shape:
            raise RuntimeError("residual_layer_norm: skip and branch differ in shape")
        C = skip.shape[-1]
        rows = skip.numel() // C
        B = skip.shape[0]
        if scale is not None:
            scale = _require(scale.reshape(-1).contiguous(), "scale", (B,))
        xsum = torch.empty_like(skip)
        y = torch.empty_like(skip)
        stats = torch.empty(rows, 2, device=skip.device, dtype=torch.float32)
        _lib.check(_lib.lib().mlagg_residual_layernorm_fwd(_ptr(skip), _ptr(branch), _ptr(scale), _ptr(weight), _ptr(bias), _ptr(xsum),
                                                           _ptr(y), _ptr(stats), rows, rows // B, C, float(eps), _stream()),
                   "mlagg_residual_layernorm_fwd")
        ctx.save_for_backward(xsum, weight, stats, scale)
        ctx.has_bias = bias is not None
        return xsum, y

    @staticmethod
    def backward(ctx, dxsum, dy):
        xsum, weight, stats, scale = ctx.saved_tensors
        C = xsum.shape[-1]
        rows = xsum.numel() // C
        B = xsum.shape[0]
        lib = _lib.lib()
        if dy is None:                  # the norm's output went nowhere: a plain residual junction
            dres = _require(dxsum.contiguous(), "dxsum")
            dbranch = dres if scale is None else dres * scale.view((-1,) + (1,) * (dres.dim() - 1))
            return dres, dbranch, None, torch.zeros_like(weight), (torch.zeros_like(weight) if ctx.has_bias else None), None
        dy2, dys = _rows2d(dy, "dy")
        if dys % 4 or dy2.data_ptr() % 16:
            dy2 = dy2.contiguous()
            dys = C
        dres = None if dxsum is None else _require(dxsum.contiguous(), "dxsum")
        dskip = torch.empty_like(xsum)
        dbranch = torch.empty_like(xsum) if scale is not None else None
        dg = torch.empty(C, device=dy.device, dtype=torch.float32)
        db = torch.empty(C, device=dy.device, dtype=torch.float32) if ctx.has_bias else None
        ws = torch.empty(lib.mlagg_layernorm_bwd_workspace_floats(rows, C), device=dy.device, dtype=torch.float32)
        _lib.check(lib.mlagg_residual_layernorm_bwd(_ptr(xsum), _ptr(dy2), dys, _ptr(dres), _ptr(scale), _ptr(weight), _ptr(stats),
                                                    _ptr(dskip), _ptr(dbranch), _ptr(dg), _ptr(db), _ptr(ws), rows, rows // B, C,
                                                    _stream()), "mlagg_residual_layernorm_bwd")
        return dskip, (dskip if dbranch is None else dbranch), None, dg, db, None


def residual_layer_norm(skip, branch, scale, weight, bias, eps=1e-5):
    """(skip + branch * scale[sample], LayerNorm of that sum) in one pass each way; scale None: plain sum."""
    if not _lib.lib().mlagg_layernorm_supported(int(skip.shape[-1])):
        raise RuntimeError(f"residual_layer_norm: {skip.shape[-1]} channels are outside K6's row shapes")
    return ResidualLayerNormFn.apply(skip, branch, scale, weight, bias, eps)


class DWConv3x3NCHWFn(torch.autograd.Function):
    """K2n: depthwise 3x3 (stride 1 or 2, padding 1) + bias on NCHW maps."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        x = _require(x.contiguous(), "x")
        B, C, H, W = x.shape
        w = _require(weight.reshape(C, 9).contiguous(), "weight")
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty(B, C, Ho, Wo, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().mlagg_dwconv3x3_nchw_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, C, H, W, int(stride),
                                                       _stream()), "mlagg_dwconv3x3_nchw_fwd")
        ctx.save_for_backward(x, w)
        ctx.meta = (int(stride), bias is not None, weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, has_bias, wshape = ctx.meta
        B, C, H, W = x.shape
        dy = _require(dy.contiguous(), "dy")
        lib = _lib.lib()
        dx = torch.empty_like(x)
        dw = torch.empty(C, 9, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32) if has_bias else None
        ws = torch.empty(lib.mlagg_dwconv3x3_nchw_bwd_workspace_floats(B, C, H, W, stride), device=x.device,
                         dtype=torch.float32)
        _lib.check(lib.mlagg_dwconv3x3_nchw_bwd(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws), B, C, H,
                                                W, stride, _stream()), "mlagg_dwconv3x3_nchw_bwd")
        return dx, dw.reshape(wshape), db, None


def dwconv3x3_nchw(x, weight, bias, stride=1):
    return DWConv3x3NCHWFn.apply(x, weight, bias, stride)


DWC_RES = _os.environ.get("MLAGG_DWC_RES", "1") == "1"


class DWConvResNCHWFn(torch.autograd.Function):
    """(conv1(x), x) of a residual MedNeXtBlock (T:256-300: ``x1 = conv1(x) ... x1 = x + x1``): the block input goes through K2n and, as
    the second output, on to the residual sum.  Backward receives both gradients of x and K2n's data-gradient kernel sums them
    (mlagg_dwconv3x3_nchw_bwd_res) -- autograd's add_ kernel over the map (three per stage, 161 us of the step) is gone."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _require(x.contiguous(), "x")
        B, C, H, W = x.shape
        w = _require(weight.reshape(C, 9).contiguous(), "weight")
        y = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().mlagg_dwconv3x3_nchw_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, C, H, W, 1, _stream()),
                   "mlagg_dwconv3x3_nchw_fwd")
        ctx.save_for_backward(x, w)
        ctx.meta = (bias is not None, weight.shape)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, w = ctx.saved_tensors
        has_bias, wshape = ctx.meta
        B, C, H, W = x.shape
        if dy is None:                                            # only the residual path carried a gradient
            return dres, None, None
        dy = _require(dy.contiguous(), "dy")
        if dres is not None:
            dres = _require(dres.contiguous(), "dres")
            if dres.dtype != torch.float32:
                dres = dres.float()
        lib = _lib.lib()
        dx = torch.empty_like(x)
        dw = torch.empty(C, 9, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32) if has_bias else None
        ws = torch.empty(lib.mlagg_dwconv3x3_nchw_bwd_workspace_floats(B, C, H, W, 1), device=x.device, dtype=torch.float32)
        _lib.check(lib.mlagg_dwconv3x3_nchw_bwd_res(_ptr(x), _ptr(w), _ptr(dy), _ptr(dres), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws), B, C, H,
                                                    W, 1, _stream()), "mlagg_dwconv3x3_nchw_bwd_res")
        return dx, dw.reshape(wshape), db


def dwconv3x3_nchw_res(x, weight, bias):
    """(conv(x), x): the depthwise convolution and the map itself for the block's residual sum; None when the fused backward does not
    apply (the caller keeps the plain form)."""
    if not (DWC_RES and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[3] % 4 == 0 and x.requires_grad
            and torch.is_grad_enabled()):
        return None
    return DWConvResNCHWFn.apply(x, weight, bias)


def _int_array(vals):
    import ctypes
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _xscan(tok, tok_stride, blk_stride, seq, B, HW, CB, nblk, merge):
    Hs, Ws = _int_array([h for h, _ in HW]), _int_array([w for _, w in HW])
    lib = _lib.lib()
    if merge:
        _lib.check(lib.mlagg_cross_merge(_ptr(seq), tok, tok_stride, blk_stride, B, len(HW), Hs, Ws, CB, nblk, _stream()),
                   "mlagg_cross_merge")
    else:
        _lib.check(lib.mlagg_cross_scan(tok, tok_stride, blk_stride, _ptr(seq), B, len(HW), Hs, Ws, CB, nblk, _stream()),
                   "mlagg_cross_scan")


class CrossScanFn(torch.autograd.Function):
    """K1' scatter: token-major (B, L_cat, nblk*CB) -> (B, 4*CB, L_cat) scan sequences (nblk = 1: the four
    directions share the source; nblk = 4: direction k reads channel block k)."""

    @staticmethod
    def forward(ctx, tok, HW, CB, nblk):
        tok = _require(tok.contiguous(), "tok")
        B, Lc, width = tok.shape
        if Lc != sum(h * w for h, w in HW) or width != nblk * CB:
            raise RuntimeError(f"cross_scan: bad shape {tuple(tok.shape)} for maps {HW}, CB {CB}, nblk {nblk}")
        seq = torch.empty(B, 4 * CB, Lc, device=tok.device, dtype=torch.float32)
        _xscan(tok.data_ptr(), width, CB, seq, B, HW, CB, nblk, merge=False)
        ctx.meta = (tuple(HW), CB, nblk, width)
        return seq

    @staticmethod
    def backward(ctx, dseq):
        HW, CB, nblk, width = ctx.meta
        dseq = _require(dseq.contiguous(), "dseq")
        B, _, Lc = dseq.shape
        dtok = torch.empty(B, Lc, width, device=dseq.device, dtype=torch.float32)
        _xscan(dtok.data_ptr(), width, CB, dseq, B, HW, CB, nblk, merge=True)
        return dtok, None, None, None


class CrossMergeFn(torch.autograd.Function):
    """K1' gather: (B, 4*CB, L_cat) scan-order outputs -> (B, L_cat, CB) token-major sum of the four directions."""

    @staticmethod
    def forward(ctx, seq, HW, CB):
        seq = _require(seq.contiguous(), "seq")
        B, rows, Lc = seq.shape
        if rows != 4 * CB or Lc != sum(h * w for h, w in HW):
            raise RuntimeError(f"cross_merge: bad shape {tuple(seq.shape)}")
        tok = torch.empty(B, Lc, CB, device=seq.device, dtype=torch.float32)
        _xscan(tok.data_ptr(), CB, CB, seq, B, HW, CB, 1, merge=True)
        ctx.meta = (tuple(HW), CB)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        HW, CB = ctx.meta
        dtok = _require(dtok.contiguous(), "dtok")
        B, Lc, _ = dtok.shape
        dseq = torch.empty(B, 4 * CB, Lc, device=dtok.device, dtype=torch.float32)
        _xscan(dtok.data_ptr(), CB, CB, dseq, B, HW, CB, 1, merge=False)
        return dseq, None, None


class CrossScanBCFn(torch.autograd.Function):
    """The single consumer of the token-major x_proj output (B, L_cat, 4*35), split as at MambaSkip.py:433:
    direction k's dt columns [35k, 35k+3), B columns [35k+3, 35k+19) and C columns [35k+19, 35k+35) all come back as
    scan-order rows -- (B, 4, 3, L_cat), (B, 4, 16, L_cat), (B, 4, 16, L_cat) -- for the low-rank selective scan.
    One backward assembles the whole x_proj gradient (no per-slice zero-fills and accumulations)."""

    @staticmethod
    def forward(ctx, xdbl, HW, dt_rank, d_state):
        xdbl = _require(xdbl.contiguous(), "x_dbl")
        B, Lc, width = xdbl.shape
        per = dt_rank + 2 * d_state
        if width != 4 * per:
            raise RuntimeError("cross_scan_bc: x_dbl must hold 4 directions")
        Bs = torch.empty(B, 4 * d_state, Lc, device=xdbl.device, dtype=torch.float32)
        Cs = torch.empty(B, 4 * d_state, Lc, device=xdbl.device, dtype=torch.float32)
        base = xdbl.data_ptr()
        _xscan(base + 4 * dt_rank, width, per, Bs, B, HW, d_state, 4, merge=False)
        _xscan(base + 4 * (dt_rank + d_state), width, per, Cs, B, HW, d_state, 4, merge=False)
        dtr = torch.empty(B, 4 * dt_rank, Lc, device=xdbl.device, dtype=torch.float32)
        _xscan(base, width, per, dtr, B, HW, dt_rank, 4, merge=False)
        ctx.meta = (tuple(HW), dt_rank, d_state, width)
        return dtr.view(B, 4, dt_rank, Lc), Bs.view(B, 4, d_state, Lc), Cs.view(B, 4, d_state, Lc)

    @staticmethod
    def backward(ctx, ddtr, dBs, dCs):
        HW, dt_rank, d_state, width = ctx.meta
        per = dt_rank + 2 * d_state
        dBs = _require(dBs.contiguous(), "dBs")
        dCs = _require(dCs.contiguous(), "dCs")
        B, Lc = dBs.shape[0], dBs.shape[-1]
        ddtr = _require(ddtr.contiguous(), "ddtr")
        dx = torch.empty(B, Lc, width, device=dBs.device, dtype=torch.float32)
        base = dx.data_ptr()
        _xscan(base, width, per, ddtr, B, HW, dt_rank, 4, merge=True)
        _xscan(base + 4 * dt_rank, width, per, dBs, B, HW, d_state, 4, merge=True)
        _xscan(base + 4 * (dt_rank + d_state), width, per, dCs, B, HW, d_state, 4, merge=True)
        return dx, None, None, None


class IndexScanFn(torch.autograd.Function):
    """K1' for volumes: token-major (B, L, width) -> scan rows (B, K*CB, L) by permutation table idx (K, L) int32; direction k
    reads columns [k*blk, k*blk + CB) of the source (blk = 0: every direction reads the same CB columns)."""

    @staticmethod
    def forward(ctx, tok, idx, CB, blk, col0):
        tok = _require(tok.contiguous(), "tok")
        B, L, width = tok.shape
        K = idx.shape[0]
        if idx.dtype != torch.int32 or tuple(idx.shape) != (K, L) or not idx.is_cuda or col0 + (K - 1) * blk + CB > width:
            raise RuntimeError("index_scan: idx must be an int32 (K, L) device table and the column blocks must fit the rows")
        seq = torch.empty(B, K * CB, L, device=tok.device, dtype=torch.float32)
        _lib.check(_lib.lib().mlagg_index_scan(tok.data_ptr() + 4 * col0, width, blk, _ptr(idx), _ptr(seq), B, L, K, CB, _stream()),
                   "mlagg_index_scan")
        ctx.save_for_backward(idx)
        ctx.meta = (CB, blk, col0, width)
        return seq

    @staticmethod
    def backward(ctx, dseq):
        (idx,) = ctx.saved_tensors
        CB, blk, col0, width = ctx.meta
        dseq = _require(dseq.contiguous(), "dseq")
        B, _, L = dseq.shape
        K = idx.shape[0]
        if blk == 0 and width == CB:
            dtok = torch.empty(B, L, width, device=dseq.device, dtype=torch.float32)      # the summed form zero-fills itself
            _lib.check(_lib.lib().mlagg_index_merge(_ptr(dseq), _ptr(idx), _ptr(dtok), width, 0, B, L, K, CB, _stream()),
                       "mlagg_index_merge")
            return dtok, None, None, None, None
        if blk == 0:
            raise RuntimeError("index_scan: a shared source must be exactly CB columns wide")
        dtok = torch.zeros(B, L, width, device=dseq.device, dtype=torch.float32)
        _lib.check(_lib.lib().mlagg_index_merge(_ptr(dseq), _ptr(idx), dtok.data_ptr() + 4 * col0, width, blk, B, L, K, CB, _stream()),
                   "mlagg_index_merge")
        return dtok, None, None, None, None


class IndexMergeFn(torch.autograd.Function):
    """(B, K*CB, L) scan-order outputs -> (B, L, CB) token-major SUM of the K directions (SS3D.forward's torch.sum(y, dim=1))."""

    @staticmethod
    def forward(ctx, seq, idx, CB):
        seq = _require(seq.contiguous(), "seq")
        B, rows, L = seq.shape
        K = idx.shape[0]
        if rows != K * CB or tuple(idx.shape) != (K, L) or idx.dtype != torch.int32:
            raise RuntimeError(f"index_merge: bad shapes seq {tuple(seq.shape)} idx {tuple(idx.shape)}")
        tok = torch.empty(B, L, CB, device=seq.device, dtype=torch.float32)
        if K > 1 and CB % 4 == 0:
            # every direction into its own column block (plain stores), then the K blocks summed in a fixed order: deterministic,
            # and faster than K float-atomic read-modify-writes per output
            wide = torch.empty(B, L, K * CB, device=seq.device, dtype=torch.float32)
            _lib.check(_lib.lib().mlagg_index_merge(_ptr(seq), _ptr(idx), _ptr(wide), K * CB, CB, B, L, K, CB, _stream()),
                       "mlagg_index_merge")
            _lib.check(_lib.lib().mlagg_block_sum(_ptr(wide), _ptr(tok), B * L, K, CB, _stream()), "mlagg_block_sum")
        else:
            _lib.check(_lib.lib().mlagg_index_merge(_ptr(seq), _ptr(idx), _ptr(tok), CB, 0, B, L, K, CB, _stream()), "mlagg_index_merge")
        ctx.save_for_backward(idx)
        ctx.CB = CB
        return tok

    @staticmethod
    def backward(ctx, dtok):
        (idx,) = ctx.saved_tensors
        dtok = _require(dtok.contiguous(), "dtok")
        B, L, CB = dtok.shape
        K = idx.shape[0]
        dseq = torch.empty(B, K * CB, L, device=dtok.device, dtype=torch.float32)
        _lib.check(_lib.lib().mlagg_index_scan(_ptr(dtok), CB, 0, _ptr(idx), _ptr(dseq), B, L, K, CB, _stream()), "mlagg_index_scan")
        return dseq, None, None


class IndexScanBCFn(torch.autograd.Function):
    """The single consumer of the token-major x_proj output of the d_state = 1 blocks, (B, L, K * (R + 2)) split per direction as
    [dt (R) | B | C] (UMambaEnc_SS3D.py:262-263): scan-order rows dtr (B, K, R, L), Bs (B, K, L), Cs (B, K, L).  One backward
    assembles the whole x_proj gradient (every column is written: no zero-fill, no per-slice accumulation)."""

    @staticmethod
    def forward(ctx, xdbl, idx, R):
        xdbl = _require(xdbl.contiguous(), "x_dbl")
        B, L, width = xdbl.shape
        K, per = idx.shape[0], R + 2
        if width != K * per or idx.dtype != torch.int32 or tuple(idx.shape) != (K, L) or not idx.is_cuda:
            raise RuntimeError("index_scan_bc: x_dbl must be (B, L, K * (R + 2)) and idx an int32 (K, L) device table")
        lib = _lib.lib()
        dtr = torch.empty(B, K * R, L, device=xdbl.device, dtype=torch.float32)
        bc = torch.empty(2, B, K, L, device=xdbl.device, dtype=torch.float32)
        base = xdbl.data_ptr()
        _lib.check(lib.mlagg_index_scan(base, width, per, _ptr(idx), _ptr(dtr), B, L, K, R, _stream()), "mlagg_index_scan")
        _lib.check(lib.mlagg_index_scan(base + 4 * R, width, per, _ptr(idx), _ptr(bc[0]), B, L, K, 1, _stream()), "mlagg_index_scan")
        _lib.check(lib.mlagg_index_scan(base + 4 * (R + 1), width, per, _ptr(idx), _ptr(bc[1]), B, L, K, 1, _stream()),
                   "mlagg_index_scan")
        ctx.save_for_backward(idx)
        ctx.meta = (R, width)
        return dtr.view(B, K, R, L), bc[0], bc[1]

    @staticmethod
    def backward(ctx, ddtr, dBs, dCs):
        (idx,) = ctx.saved_tensors
        R, width = ctx.meta
        per = R + 2
        ddtr = _require(ddtr.contiguous(), "ddtr")
        dBs = _require(dBs.contiguous(), "dBs")
        dCs = _require(dCs.contiguous(), "dCs")
        B, K, _, L = ddtr.shape
        lib = _lib.lib()
        dx = torch.empty(B, L, width, device=ddtr.device, dtype=torch.float32)
        base = dx.data_ptr()
        _lib.check(lib.mlagg_index_merge(_ptr(ddtr), _ptr(idx), base, width, per, B, L, K, R, _stream()), "mlagg_index_merge")
        _lib.check(lib.mlagg_index_merge(_ptr(dBs), _ptr(idx), base + 4 * R, width, per, B, L, K, 1, _stream()), "mlagg_index_merge")
        _lib.check(lib.mlagg_index_merge(_ptr(dCs), _ptr(idx), base + 4 * (R + 1), width, per, B, L, K, 1, _stream()),
                   "mlagg_index_merge")
        return dx, None, None


def index_scan_bc(xdbl, idx, R):
    return IndexScanBCFn.apply(xdbl, idx, R)


def index_scan(tok, idx, CB, blk=0, col0=0):
    return IndexScanFn.apply(tok, idx, CB, blk, col0)


def index_merge(seq, idx, CB):
    return IndexMergeFn.apply(seq, idx, CB)


def cross_scan(tok, HW, CB, nblk):
    return CrossScanFn.apply(tok, HW, CB, nblk)


def cross_merge(seq, HW, CB):
    return CrossMergeFn.apply(seq, HW, CB)


def cross_scan_bc(xdbl, HW, dt_rank, d_state):
    return CrossScanBCFn.apply(xdbl, HW, dt_rank, d_state)


class GateFn(torch.autograd.Function):
    """K7: concat(a0, a1) * SiLU(act) (the MLLA block's gate) in one pass each way."""

    @staticmethod
    def forward(ctx, a0, a1, act, slot=None):
        ctx.slot = slot
        a0 = _require(a0.contiguous(), "a0")
        a1 = _require(a1.contiguous(), "a1")
        act2, acts = _rows2d(act, "act")
        h = a0.shape[-1]
        rows = a0.numel() // h
        if a1.shape != a0.shape or act.shape[-1] != 2 * h or act2.shape[0] != rows:
            raise RuntimeError("gate: shape mismatch")
        if acts % 4 or act2.data_ptr() % 16:
            act2 = act2.contiguous()
            acts = 2 * h
        out = torch.empty(a0.shape[:-1] + (2 * h,), device=a0.device, dtype=torch.float32)
        _lib.check(_lib.lib().mlagg_gate_fwd(_ptr(a0), _ptr(a1), _ptr(act2), acts, _ptr(out), rows, h, _stream()),
                   "mlagg_gate_fwd")
        ctx.save_for_backward(a0, a1, act2)
        return out

    @staticmethod
    def backward(ctx, dout):
        a0, a1, act2 = ctx.saved_tensors
        h = a0.shape[-1]
        rows = a0.numel() // h
        d2, ds = _rows2d(dout, "dout")
        if ds % 4 or d2.data_ptr() % 16:
            d2 = d2.contiguous()
            ds = 2 * h
        da0, da1 = torch.empty_like(a0), torch.empty_like(a1)
        dact, dacts = _grad_out(ctx.slot, a0.shape[:-1] + (2 * h,), a0.device)
        _lib.check(_lib.lib().mlagg_gate_bwd(_ptr(d2), ds, _ptr(a0), _ptr(a1), _ptr(act2), act2.stride(0), _ptr(da0),
                                             _ptr(da1), _ptr(dact), dacts, rows, h, _stream()), "mlagg_gate_bwd")
        return da0, da1, dact, None


def gate(a0, a1, act):
    return GateFn.apply(a0, a1, act, _claim(act))


class GeluPoolFn(torch.autograd.Function):
    """K17: r x r window mean of GELU(s) on a token-major map (the pooled branch's key / value reduction, T:722)."""

    @staticmethod
    def forward(ctx, s, H, W, r, slot=None):
        ctx.slot = slot
        s, ss = _rows(s, "s")
        B, N, d = s.shape
        if N != H * W or H % r or W % r:
            raise RuntimeError(f"gelu_pool: {N} tokens, map {H}x{W}, window {r}")
        pooled = torch.empty(B, (H // r) * (W // r), d, device=s.device, dtype=torch.float32)
        _lib.check(_lib.lib().mlagg_gelu_pool_fwd(_ptr(s), ss, _ptr(pooled), B, H, W, d, r, _stream()), "mlagg_gelu_pool_fwd")
        ctx.save_for_backward(s)
        ctx.geom = (H, W, r)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        (s,) = ctx.saved_tensors
        H, W, r = ctx.geom
        B, N, d = s.shape
        dp = _require(dpooled.contiguous(), "dpooled")
        ds, dss = _grad_out(ctx.slot, (B, N, d), s.device)
        _lib.check(_lib.lib().mlagg_gelu_pool_bwd(_ptr(s), s.stride(1), _ptr(dp), _ptr(ds), dss, B, H, W, d, r, _stream()),
                   "mlagg_gelu_pool_bwd")
        return ds, None, None, None, None


def gelu_pool(s, H, W, r):
    return GeluPoolFn.apply(s, H, W, r, _claim(s))


class DiffLambdaFn(torch.autograd.Function):
    """K8: lambda = exp(<q1, k1>) - exp(<q2, k2>) + lambda_init of the differential attention, one launch each way."""

    @staticmethod
    def forward(ctx, q1, k1, q2, k2, lambda_init):
        vs = [_require(v.contiguous(), "lambda vector") for v in (q1, k1, q2, k2)]
        n = vs[0].numel()
        if any(v.numel() != n for v in vs):
            raise RuntimeError("diff_lambda: the four vectors differ in length")
        out = torch.empty(3, device=q1.device, dtype=torch.float32)            # [lambda, exp1, exp2]
        _lib.check(_lib.lib().mlagg_diff_lambda_fwd(*(_ptr(v) for v in vs), float(lambda_init), n, _ptr(out),
                                                    out.data_ptr() + 4, _stream()), "mlagg_diff_lambda_fwd")
        ctx.save_for_backward(*vs, out)
        return out[0]

    @staticmethod
    def backward(ctx, dlam):
        q1, k1, q2, k2, out = ctx.saved_tensors
        n = q1.numel()
        dlam = _require(dlam.reshape(1).contiguous(), "dlambda")
        g = torch.empty(4, n, device=q1.device, dtype=torch.float32)
        _lib.check(_lib.lib().mlagg_diff_lambda_bwd(_ptr(dlam), _ptr(q1), _ptr(k1), _ptr(q2), _ptr(k2), out.data_ptr() + 4, n,
                                                    _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(g[3]), _stream()),
                   "mlagg_diff_lambda_bwd")
        return g[0].view_as(q1), g[1].view_as(k1), g[2].view_as(q2), g[3].view_as(k2), None


def diff_lambda(q1, k1, q2, k2, lambda_init):
    return DiffLambdaFn.apply(q1, k1, q2, k2, lambda_init)


class ScaledResidualFn(torch.autograd.Function):
    """K8: skip + branch * scale[sample] (residual under stochastic depth), float4 streams both ways."""

    @staticmethod
    def forward(ctx, skip, branch, scale):
        skip = _require(skip.contiguous(), "skip")
        branch = _require(branch.contiguous(), "branch")
        scale = _require(scale.reshape(-1).contiguous(), "scale")
        B = scale.numel()
        if skip.shape != branch.shape or skip.shape[0] != B:
            raise RuntimeError("scaled_residual: shape mismatch")
        per = skip.numel() // B
        out = torch.empty_like(skip)
        _lib.check(_lib.lib().mlagg_scaled_residual(_ptr(skip), _ptr(branch), _ptr(scale), _ptr(out), B, per, _stream()),
                   "mlagg_scaled_residual")
        ctx.save_for_backward(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        g = _require(g.contiguous(), "grad")
        B = scale.numel()
        db = torch.empty_like(g)
        _lib.check(_lib.lib().mlagg_scaled_residual(None, _ptr(g), _ptr(scale), _ptr(db), B, g.numel() // B, _stream()),
                   "mlagg_scaled_residual")
        return g, db, None


def scaled_residual(skip, branch, scale):
    return ScaledResidualFn.apply(skip, branch, scale)


class DiceCEStatsFn(torch.autograd.Function):
    """K9: per-level Dice / cross-entropy statistics of ALL deep-supervision levels (one kernel per level each way).

    ``DiceCEStatsFn.apply(n_levels, ignore_label, *logits, *targets)`` -> (ip (L, B, 2, C): intersect and sum_pred per sample and
    class, gt (L, B, C): label counts, ce (L,): summed -log softmax of the label); gradients flow to the logits from ip and ce.
    ``ignore_label`` (int, -1: none): pixels with that label are left out of every sum and get no gradient."""

    @staticmethod
    def forward(ctx, n, ignore_label, *tensors):
        logits, targets = tensors[:n], tensors[n:]
        ctx.ignore = int(ignore_label)
        B, C = logits[0].shape[:2]
        dev = logits[0].device
        lib = _lib.lib()
        if C > lib.mlagg_dice_ce_max_classes():
            raise RuntimeError(f"dice_ce_stats: at most {lib.mlagg_dice_ce_max_classes()} classes")
        buf = torch.empty(n * B * 3 * C + n, device=dev, dtype=torch.float32)       # every entry is written by the level's reduce
        ip = buf[:n * B * 2 * C].view(n, B, 2, C)
        gt = buf[n * B * 2 * C:n * B * 3 * C].view(n, B, C)
        ce = buf[n * B * 3 * C:]
        saved = []
        for i, (z, t) in enumerate(zip(logits, targets)):
            z = _require(z.contiguous(), "logits")
            t = _require(t.contiguous(), "target")
            if z.shape[:2] != (B, C) or t.numel() * C != z.numel():
                raise RuntimeError("dice_ce_stats: logits / target shapes of a level do not match")
            hw = z.numel() // (B * C)
            ws = torch.empty(lib.mlagg_dice_ce_stats_workspace_floats(B, C, hw), device=dev, dtype=torch.float32)
            _lib.check(lib.mlagg_dice_ce_stats(_ptr(z), _ptr(t), _ptr(ip[i]), _ptr(gt[i]), ce.data_ptr() + 4 * i, _ptr(ws), B, C, hw,
                                               ctx.ignore, _stream()), "mlagg_dice_ce_stats")
            saved += [z, t]
        ctx.save_for_backward(*saved)
        ctx.n = n
        ctx.mark_non_differentiable(gt)
        return ip, gt, ce

    @staticmethod
    def backward(ctx, g_ip, g_gt, g_ce):
        n = ctx.n
        saved = ctx.saved_tensors
        B, C = saved[0].shape[:2]
        lib = _lib.lib()
        g_ip = torch.zeros(n, B, 2, C, device=saved[0].device) if g_ip is None else _require(g_ip.contiguous(), "g_ip")
        g_ce = torch.zeros(n, device=saved[0].device) if g_ce is None else _require(g_ce.contiguous(), "g_ce")
        grads = []
        for i in range(n):
            z, t = saved[2 * i], saved[2 * i + 1]
            dz = torch.empty_like(z)
            _lib.check(lib.mlagg_dice_ce_grad(_ptr(z), _ptr(t), _ptr(g_ip[i]), g_ce.data_ptr() + 4 * i, _ptr(dz), B, C,
                                              z.numel() // (B * C), ctx.ignore, _stream()), "mlagg_dice_ce_grad")
            grads.append(dz)
        return (None, None, *grads, *([None] * n))


def dice_ce_stats(logits, targets, ignore_label=None):
    return DiceCEStatsFn.apply(len(logits), -1 if ignore_label is None else int(ignore_label), *logits, *targets)


def transpose_2d(src):
    """(B, R, C) -> (B, C, R) contiguous on the tiled transpose kernel (K8).  The source matrices must be contiguous;
    their batch stride may be larger than R * C (a channel slice of an NCHW map), anything else is copied first."""
    _require(src, "src")
    B, R, C = src.shape
    if not (src.stride(2) == 1 and src.stride(1) == C and src.stride(0) >= R * C and src.data_ptr() % 16 == 0
            and (C % 4 or src.stride(0) % 4 == 0)):
        src = src.contiguous()
    dst = torch.empty(B, C, R, device=src.device, dtype=torch.float32)
    _lib.check(_lib.lib().mlagg_transpose_2d(_ptr(src), src.stride(0), _ptr(dst), B, R, C, _stream()), "mlagg_transpose_2d")
    return dst


class ChannelBiasFn(torch.autograd.Function):
    """y = x + bias[c] on an NCHW map, in place on the convolution output; backward reduces the bias gradient with the
    library's plane-sum kernel (torch's convolution_backward does it with a generic reduction at 1.3-2 TB/s)."""

    @staticmethod
    def forward(ctx, x, bias):
        ctx.mark_dirty(x)
        x.add_(bias.view(1, -1, *([1] * (x.dim() - 2))))
        return x

    @staticmethod
    def backward(ctx, g):
        g = _require(g.contiguous(), "grad")
        B, C = g.shape[:2]
        lib = _lib.lib()
        db = torch.empty(C, device=g.device, dtype=torch.float32)
        ws = torch.empty(lib.mlagg_channel_sum_workspace_floats(B, C), device=g.device, dtype=torch.float32)
        _lib.check(lib.mlagg_channel_sum(_ptr(g), _ptr(db), _ptr(ws), B, C, g.numel() // (B * C), _stream()), "mlagg_channel_sum")
        return g, db


def channel_bias(x, bias):
    return ChannelBiasFn.apply(x, bias)


EPI_NONE, EPI_GELU = 0, 1


class ChannelEpilogueFn(torch.autograd.Function):
    """K13: y = act(x + bias[c] + res) on the NCHW output `x` of a library convolution, one pass.  `x` is consumed: without an
    activation it is updated in place and returned; with GELU it is overwritten with the pre-activation (kept for backward)
    and the result is a new map."""

    @staticmethod
    def forward(ctx, x, bias, res, act):
        x = _require(x, "x")
        if not x.is_contiguous():
            raise RuntimeError("channel_epilogue: contiguous NCHW map expected")
        res = None if res is None else _require(res.contiguous(), "res", x.shape)
        B, C = x.shape[:2]
        hw = x.numel() // (B * C)
        if act == EPI_GELU and x._version != 0:
            # the GELU form overwrites x with the pre-activation WITHOUT telling autograd: only a map nothing else has
            # written or saved in a modified state (a fresh convolution output) may be handed in
            raise RuntimeError("channel_epilogue(GELU): x must be the fresh output of the producing call")
        y = torch.empty_like(x) if act == EPI_GELU else None
        _lib.check(_lib.lib().mlagg_channel_epilogue_fwd(_ptr(x), _ptr(bias), _ptr(res), _ptr(y), B, C, hw, int(act), _stream()),
                   "mlagg_channel_epilogue_fwd")
        ctx.meta = (int(act), bias is not None, res is not None)
        if act == EPI_GELU:
            # x now holds the pre-activation.  It is the fresh output of the convolution call in front of this function
            # (nothing else reads it, convolution backward does not need its own output), so it is simply kept.
            ctx.save_for_backward(x)
            return y
        ctx.mark_dirty(x)
        return x

    @staticmethod
    def backward(ctx, dy):
        act, has_bias, has_res = ctx.meta
        dy = _require(dy.contiguous(), "dy")
        B, C = dy.shape[:2]
        hw = dy.numel() // (B * C)
        lib = _lib.lib()
        db = torch.empty(C, device=dy.device, dtype=torch.float32) if has_bias else None
        ws = torch.empty(lib.mlagg_channel_sum_workspace_floats(B, C), device=dy.device, dtype=torch.float32) if has_bias else None
        if act == EPI_GELU:
            (pre,) = ctx.saved_tensors
            dx = torch.empty_like(dy)
            _lib.check(lib.mlagg_channel_gelu_bwd(_ptr(pre), _ptr(dy), _ptr(dx), _ptr(db), _ptr(ws), B, C, hw, _stream()),
                       "mlagg_channel_gelu_bwd")
        else:
            dx = dy
            if has_bias:
                _lib.check(lib.mlagg_channel_sum(_ptr(dy), _ptr(db), _ptr(ws), B, C, hw, _stream()), "mlagg_channel_sum")
        return dx, db, (dx if has_res else None), None


def channel_epilogue(x, bias=None, res=None, act=EPI_NONE):
    return ChannelEpilogueFn.apply(x, bias, res, act)


def column_sum(x2):
    """Sum over the rows of a (rows, cols) matrix with unit inner stride."""
    _require(x2, "x")
    rows, cols = x2.shape
    out = torch.empty(cols, device=x2.device, dtype=torch.float32)
    lib = _lib.lib()
    n = lib.mlagg_column_sum_workspace_floats(rows, cols)
    ws = torch.empty(n, device=x2.device, dtype=torch.float32) if n else None
    _lib.check(lib.mlagg_column_sum(_ptr(x2), x2.stride(0), _ptr(out), _ptr(ws), rows, cols, _stream()), "mlagg_column_sum")
    return out


ACT_NONE, ACT_LEAKY, ACT_SILU = 0, 1, 2


_DT_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}            # MLAGG_DTYPE_* of include/mlagg_hip.h


def _require_map(t, name, shape=None):
    """A device map in fp32, bf16 or fp16 (the kernels of the convolutional chains take the element type as an argument)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype in _DT_CODE):
        raise RuntimeError(f"{name}: expected an fp32 / bf16 / fp16 tensor on the MI355X device, got "
                           f"{getattr(t, 'dtype', type(t))} on {getattr(t, 'device', '?')}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


SLICE_GRADS = _os.environ.get("MLAGG_SLICE_GRADS", "1") == "1"     # 0: copy channel-slice gradients before the kernels (the round-3 form)


def _map_slice(t, name):
    """A gradient map as the K10 / shuffle kernels can read it without a copy: (tensor, elements between samples) -- dense, or a channel
    slice of a wider dense map (what ``torch.cat([a, b], 1)``'s backward hands to the producers of a and b); anything else is copied."""
    if SLICE_GRADS and t.is_cuda and t.dim() >= 3 and not t.is_contiguous():
        inner, want = 1, []
        for v in reversed(t.shape[1:]):
            want.append(inner)
            inner *= int(v)
        want.reverse()
        plane = inner // int(t.shape[1])
        if tuple(t.stride()[1:]) == tuple(want) and t.stride(0) >= inner and t.stride(0) % plane == 0 and t.stride(0) % 4 == 0 and \
                t.data_ptr() % 16 == 0 and t.dtype in _DT_CODE:
            return t, t.stride(0)
    t = _require_map(t.contiguous(), name)
    return t, 0


class PlaneNormFn(torch.autograd.Function):
    """K10: per-(batch, channel)-plane normalisation of an NCHW map fused with what follows it: y = act(norm(x) + res)
    (GroupNorm(C, C); InstanceNorm2d + LeakyReLU; InstanceNorm2d(affine) + SiLU; the residual sum of the UnetResBlock).
    x, res and y may each be fp32, bf16 or fp16 IN MEMORY (16-bit modes: maps between the 16-bit library convolutions);
    statistics and arithmetic are fp32."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, eps, act, slope, out_dtype):
        x = _require_map(x.contiguous(), "x")
        res = None if res is None else _require_map(res.contiguous(), "res", x.shape)
        B, C = x.shape[:2]
        hw = x.numel() // (B * C)
        y = torch.empty(x.shape, device=x.device, dtype=out_dtype or x.dtype)
        stats = torch.empty(B * C, 2, device=x.device, dtype=torch.float32)
        nws = _lib.lib().mlagg_plane_norm_fwd_workspace_floats(B, C, hw)          # > 0: planes cut into segments (3-D volumes)
        ws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws else None
        _lib.check(_lib.lib().mlagg_plane_norm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(res), _ptr(y), _ptr(stats), _ptr(ws), B, C, hw,
                                                   float(eps), int(act), float(slope), _DT_CODE[x.dtype],
                                                   0 if res is None else _DT_CODE[res.dtype], _DT_CODE[y.dtype], _stream()),
                   "mlagg_plane_norm_fwd")
        ctx.save_for_backward(x, gamma, beta, res, stats)
        ctx.meta = (int(act), float(slope))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, res, stats = ctx.saved_tensors
        act, slope = ctx.meta
        dy, dyb = _map_slice(dy, "dy")
        B, C = x.shape[:2]
        hw = x.numel() // (B * C)
        lib = _lib.lib()
        dx = torch.empty_like(x)
        dres = torch.empty_like(res) if (res is not None and ctx.needs_input_grad[3]) else None
        dg = torch.empty_like(gamma) if gamma is not None else None
        db = torch.empty_like(beta) if beta is not None else None
        segmented = lib.mlagg_plane_norm_fwd_workspace_floats(B, C, hw) > 0
        ws = torch.empty(lib.mlagg_plane_norm_bwd_workspace_floats(B, C, hw), device=x.device, dtype=torch.float32) \
            if (dg is not None or db is not None or segmented) else None
        _lib.check(lib.mlagg_plane_norm_bwd_strided(_ptr(x), _ptr(dy), dyb, _ptr(gamma), _ptr(beta), _ptr(res), _ptr(stats), _ptr(dx),
                                                    _ptr(dres), _ptr(dg), _ptr(db), _ptr(ws), B, C, hw, act, slope, _DT_CODE[x.dtype],
                                                    _DT_CODE[dy.dtype], 0 if res is None else _DT_CODE[res.dtype], _stream()),
                   "mlagg_plane_norm_bwd_strided")
        return dx, dg, db, dres, None, None, None, None


def plane_norm(x, gamma, beta, eps=1e-5, act=ACT_NONE, slope=0.0, res=None, out_dtype=None):
    return PlaneNormFn.apply(x, gamma, beta, res, eps, act, slope, out_dtype)


class ChannelEpilogueLpFn(torch.autograd.Function):
    """K13 in the 16-bit modes: y = act(x + bias[c] + res) where x is the bf16 / fp16 output of a library convolution (left
    untouched: backward recomputes the pre-activation from it), y in ``out_dtype``; dx comes back in x's type -- the convolution's
    gradient operand -- with d(bias) from the same pass."""

    @staticmethod
    def forward(ctx, x, bias, res, act, out_dtype):
        x = _require_map(x.contiguous(), "x")
        res = None if res is None else _require_map(res.contiguous(), "res", x.shape)
        B, C = x.shape[:2]
        hw = x.numel() // (B * C)
        y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
        _lib.check(_lib.lib().mlagg_channel_epilogue_lp_fwd(_ptr(x), _DT_CODE[x.dtype], _ptr(bias), _ptr(res),
                                                            0 if res is None else _DT_CODE[res.dtype], _ptr(y), _DT_CODE[y.dtype], B, C, hw,
                                                            int(act), _stream()), "mlagg_channel_epilogue_lp_fwd")
        ctx.save_for_backward(x if act == EPI_GELU else None, bias, res if act == EPI_GELU else None)
        ctx.meta = (int(act), x.dtype, None if res is None else res.dtype, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, bias, res = ctx.saved_tensors
        act, xdt, rdt, shape = ctx.meta
        dy = _require_map(dy.contiguous(), "dy", shape)
        B, C = shape[:2]
        hw = dy.numel() // (B * C)
        lib = _lib.lib()
        dx = torch.empty(shape, device=dy.device, dtype=xdt)
        db = torch.empty(C, device=dy.device, dtype=torch.float32) if bias is not None else None
        ws = torch.empty(lib.mlagg_channel_sum_workspace_floats(B, C), device=dy.device, dtype=torch.float32) if bias is not None else None
        _lib.check(lib.mlagg_channel_epilogue_lp_bwd(_ptr(x), _DT_CODE[xdt], _ptr(bias), _ptr(res), 0 if res is None else _DT_CODE[res.dtype],
                                                     _ptr(dy), _DT_CODE[dy.dtype], _ptr(dx), _DT_CODE[xdt], _ptr(db), _ptr(ws), B, C, hw,
                                                     act, _stream()), "mlagg_channel_epilogue_lp_bwd")
        dres = None
        if rdt is not None and ctx.needs_input_grad[2]:
            # d(res) = d(pre): dy itself without an activation, else the values of dx -- in res's own element type
            src = dy if act == EPI_NONE else dx
            dres = src if src.dtype == rdt else (dx if dx.dtype == rdt else src.to(rdt))
        return dx, db, dres, None, None


def channel_epilogue_lp(x, bias=None, res=None, act=EPI_NONE, out_dtype=torch.float32):
    return ChannelEpilogueLpFn.apply(x, bias, res, act, out_dtype)


# ------------------------------------------------------------------------------------------------
# K15: full convolutions with the weight gradient on this package's tap-GEMM kernel
# ------------------------------------------------------------------------------------------------
def _planes(t, name):
    """(B, C, *spatial) map whose samples are contiguous (C, P) blocks -- a channel slice of a wider NCHW map qualifies; anything else
    is copied.  Returns (tensor, floats between samples, P)."""
    _require(t, name)
    inner, want = 1, []
    for v in reversed(t.shape[1:]):
        want.append(inner)
        inner *= int(v)
    want.reverse()
    if tuple(t.stride()[1:]) != tuple(want) or t.stride(0) < inner or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    return t, t.stride(0), inner // int(t.shape[1])


# K18 against MIOpen's tuned fp32 GEMM kernels on the 1 x 1 shapes of the 256 x 256 step (tools/bench_conv1x1.py,
# profiles/round3_g_conv1x1_*.log): the forward / data-gradient kernel wins where the pixel count is large and the contraction
# short-to-medium (128 x 128 and 256 x 256 maps: 70 vs 110, 64 vs 93, 98 vs 215 us), ties at 64 x 64 and loses on the small maps
# (few, long-K tiles); the weight gradient wins from 64 x 64 up (80 vs 142, 70 vs 88, 150 vs 252 us).  Each of the three products
# of a layer goes to the faster side.
K18 = _os.environ.get("MLAGG_K18", "1") == "1"
# round 4: 64 x 64 maps too (a tie with the library in the step -- 34.61 vs 34.60 ms, profiles/round4_k_k18_thresholds_ab.log -- and no NHWC transposes);
# 32 x 32 maps lose 0.3 ms
K18_FWD_MIN_PIXELS = int(_os.environ.get("MLAGG_K18_FWD_MIN_PIXELS", "4096"))
K18_FWD_MIN_K = int(_os.environ.get("MLAGG_K18_FWD_MIN_K", "96"))
K18_WGRAD_MIN_PIXELS = int(_os.environ.get("MLAGG_K18_WGRAD_MIN_PIXELS", "4096"))


K18_THIN = _os.environ.get("MLAGG_K18_THIN", "1") == "1"
K18_THIN_CH = 32


def _k18_product(O, I, P, form=_DTYPE_BF16X3):
    """forward-form product y (O) = w (O, I) . x (I) on K18?  (the data gradient asks with O and I exchanged.)  A contraction that is
    not a multiple of 16 runs on zero-padded weight columns (mlagg_conv1x1_fwd_ragged)."""
    I16 = -(-I // 16) * 16
    if not bool(_lib.lib().mlagg_conv1x1_supported(O, I16, P)):
        return False
    if form != _DTYPE_BF16X3:                # one product per block: a stream of the maps, ahead of cast + library + cast wherever it runs
        return P >= LP_K_MIN_PIXELS and (I == I16 or K18_THIN)
    if K18_THIN and P >= K18_FWD_MIN_PIXELS and min(O, I) <= K18_THIN_CH:
        # a thin side (the 14-class heads and their data gradients): one pass over the wide map, where the library's GEMM kernels
        # took 101 us forward / 261 us backward for 48 -> 14 channels at 256 x 256 (profiles/round4_i_library_convolutions_by_shape.md)
        return True
    return I == I16 and P >= K18_FWD_MIN_PIXELS and I >= K18_FWD_MIN_K


def _conv1x1_k18(x, xb, w, y, B, O, I, P, form, accumulate=False):
    """y (B, O, P) (+)= w (O, I) . x (B, I, P) on K18; a contraction that is not a multiple of 16 on zero-padded weight columns."""
    I16 = -(-I // 16) * 16
    if I16 != I:
        wp = torch.zeros(O, I16, device=w.device, dtype=torch.float32)
        wp[:, :I] = w
        w = wp
    _lib.check(_lib.lib().mlagg_conv1x1_fwd_acc(_ptr(x), xb, _ptr(w), None, _ptr(y), O * P, B, O, I16, I, P, form, int(accumulate), _stream()),
               "mlagg_conv1x1_fwd_acc")


class Conv1x1Fn(torch.autograd.Function):
    """y = conv2d(x, W) for a 1 x 1 kernel (stride 1, no bias): each of the three products on K18 (this package's split-bf16 GEMM
    kernels: forward, data gradient = the forward kernel on W^T, weight gradient with the pixels as the contraction) or on the
    library, whichever is faster at the shape (see K18_* above)."""

    @staticmethod
    def forward(ctx, x, weight, form=_DTYPE_BF16X3):
        x, xb, P = _planes(x, "x")
        B, I = x.shape[:2]
        O = weight.shape[0]
        w = _require(weight.reshape(O, I).contiguous(), "weight")
        if _k18_product(O, I, P, form):
            y = torch.empty((B, O) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
            _flop("K18", 2 * B * O * I * P)
            _conv1x1_k18(x, xb, w, y, B, O, I, P, form)
        else:
            y = _lib_conv_fwd(x, weight, 0, form)
        ctx.save_for_backward(x, w)
        ctx.wshape, ctx.form = weight.shape, form
        ctx.leaf, ctx.leaf_params = _leaf_ok(weight), [weight]
        note_leaf_use(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, I = x.shape[:2]
        O = w.shape[0]
        dy, dyb, P = _planes(dy, "dy")
        lib = _lib.lib()
        form = ctx.form
        dx = dW = None
        if ctx.needs_input_grad[0]:
            if _k18_product(I, O, P, form):
                wt = transpose_2d(w.unsqueeze(0))[0]                                   # (I, O): the contraction runs along its rows
                dx = torch.empty((B, I) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
                _flop("K18", 2 * B * O * I * P)
                _conv1x1_k18(dy, dyb, wt, dx, B, I, O, P, form)
            else:
                dx = _lib_conv_bwd(dy, x, w.view(ctx.wshape), 0, (True, False, False), form)[0]
        if ctx.needs_input_grad[1]:
            with _LeafStream(dy, x, ok=ctx.leaf and leaf_single_use(ctx.leaf_params)):
                dW = torch.empty(O, I, device=x.device, dtype=torch.float32)
                ws = torch.empty(lib.mlagg_conv1x1_wgrad_workspace_floats(B, O, I, P), device=x.device, dtype=torch.float32)
                _flop("K18", 2 * B * O * I * P)
                _lib.check(lib.mlagg_conv1x1_wgrad_lp(_ptr(dy), dyb, _ptr(x), x.stride(0), _ptr(dW), _ptr(ws), B, O, I, P, form,
                                                      _stream()), "mlagg_conv1x1_wgrad_lp")
            dW = dW.view(ctx.wshape)
        return dx, dW, None


CONVT2 = _os.environ.get("MLAGG_CONVT2", "1") == "1"


def _pixel_shuffle2(src, B, O, H, W, inverse):
    """(B, 4 O, H, W) -> (B, O, 2 H, 2 W), or back (inverse)."""
    src = _require(src.contiguous(), "src")
    dst = torch.empty((B, 4 * O, H, W) if inverse else (B, O, 2 * H, 2 * W), device=src.device, dtype=torch.float32)
    _lib.check(_lib.lib().mlagg_pixel_shuffle2(_ptr(src), _ptr(dst), B, O, H, W, int(inverse), _stream()), "mlagg_pixel_shuffle2")
    return dst


class ConvT2x2Fn(torch.autograd.Function):
    """y = conv_transpose2d(x, W (I, O, 2, 2), stride 2): the taps do not overlap, so it is the pointwise product with the (4 O, I) matrix
    [tap (a, c)][o][i] = W[i][o][a][c] on K18 followed by a pixel shuffle; backward: the inverse shuffle of dy, then K18's data
    gradient (contraction 4 O) and weight gradient.  (The library ran this layer of the 256 x 256 step in 215 us forward + 350 us
    backward: profiles/round4_i_library_convolutions_by_shape.md.)"""

    @staticmethod
    def forward(ctx, x, weight, form=_DTYPE_BF16X3):
        x, xb, P = _planes(x, "x")
        B, I, H, W = x.shape
        O = weight.shape[1]
        w4 = _require(weight.permute(2, 3, 1, 0).reshape(4 * O, I).contiguous(), "weight")
        z = torch.empty(B, 4 * O, H, W, device=x.device, dtype=torch.float32)
        _flop("K18", 2 * B * 4 * O * I * P)
        _conv1x1_k18(x, xb, w4, z, B, 4 * O, I, P, form)
        ctx.save_for_backward(x, w4)
        ctx.form, ctx.O = form, O
        ctx.leaf, ctx.leaf_params = _leaf_ok(weight), [weight]
        note_leaf_use(weight)
        return _pixel_shuffle2(z, B, O, H, W, False)

    @staticmethod
    def backward(ctx, dy):
        x, w4 = ctx.saved_tensors
        B, I, H, W = x.shape
        O, form, P = ctx.O, ctx.form, H * W
        lib = _lib.lib()
        dy, dyb = _map_slice(dy, "dy")                                                # a half of cat([up, skip])'s gradient: read in place
        if dy.dtype != torch.float32:
            dy, dyb = dy.float(), 0
        dyu = torch.empty(B, 4 * O, H, W, device=x.device, dtype=torch.float32)        # (B, 4 O, H, W)
        _lib.check(lib.mlagg_pixel_unshuffle2_strided(_ptr(dy), dyb, _ptr(dyu), B, O, H, W, _stream()), "mlagg_pixel_unshuffle2_strided")
        dx = dW = None
        if ctx.needs_input_grad[0]:
            wt = transpose_2d(w4.unsqueeze(0))[0]                                      # (I, 4 O)
            dx = torch.empty(B, I, H, W, device=x.device, dtype=torch.float32)
            _flop("K18", 2 * B * 4 * O * I * P)
            _conv1x1_k18(dyu, 4 * O * P, wt, dx, B, I, 4 * O, P, form)
        if ctx.needs_input_grad[1]:
            with _LeafStream(dyu, x, ok=ctx.leaf and leaf_single_use(ctx.leaf_params)):
                dW4 = torch.empty(4 * O, I, device=x.device, dtype=torch.float32)
                ws = torch.empty(lib.mlagg_conv1x1_wgrad_workspace_floats(B, 4 * O, I, P), device=x.device, dtype=torch.float32)
                _flop("K18", 2 * B * 4 * O * I * P)
                _lib.check(lib.mlagg_conv1x1_wgrad_lp(_ptr(dyu), 4 * O * P, _ptr(x), x.stride(0), _ptr(dW4), _ptr(ws), B, 4 * O, I, P, form,
                                                      _stream()), "mlagg_conv1x1_wgrad_lp")
                dW = dW4.view(2, 2, O, I).permute(3, 2, 0, 1).contiguous()
        return dx, dW, None


def conv_t2x2_supported(x, weight, stride, padding, output_padding, dilation, groups, form=_DTYPE_BF16X3):
    """A kernel-2 / stride-2 transposed convolution (no padding) on an fp32 device map whose three products K18 takes."""
    if not (CONVT2 and K18 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and groups == 1):
        return False
    if tuple(weight.shape[2:]) != (2, 2) or any(int(v) != 2 for v in stride) or any(int(v) != 0 for v in padding) or \
            any(int(v) != 0 for v in output_padding) or any(int(v) != 1 for v in dilation):
        return False
    I, O = int(weight.shape[0]), int(weight.shape[1])
    H, W = int(x.shape[2]), int(x.shape[3])
    P = H * W
    return W % 2 == 0 and P % 16 == 0 and _k18_product(4 * O, I, P, form) and _k18_product(I, 4 * O, P, form)


def conv_t2x2(x, weight, form=_DTYPE_BF16X3):
    return ConvT2x2Fn.apply(x, weight, form)


def conv1x1_supported(x, weight, stride, padding, dilation, groups, form=_DTYPE_BF16X3):
    """A 1 x 1, stride-1, dense convolution on an fp32 device map whose weight gradient (at least) runs on K18 in operand form `form`."""
    if not (K18 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and groups == 1):
        return False
    if tuple(weight.shape[2:]) != (1, 1) or any(int(v) != 1 for v in stride) or any(int(v) != 0 for v in padding) or \
            any(int(v) != 1 for v in dilation):
        return False
    P = int(x.shape[2] * x.shape[3])
    return P >= (K18_WGRAD_MIN_PIXELS if form == _DTYPE_BF16X3 else LP_K_MIN_PIXELS) and P % 16 == 0


def conv1x1(x, weight, form=_DTYPE_BF16X3):
    return Conv1x1Fn.apply(x, weight, form)


K19 = _os.environ.get("MLAGG_K19", "1") == "1"
# K19 against MIOpen's tuned Winograd kernels (tools/bench_conv3x3.py, batch 10): 17-30 % faster from 32 x 32 maps up (48 -> 48 at
# 256 x 256: 302 vs 416 us forward, 272 vs 356 us data gradient; 96 -> 96 at 128 x 128: 202 vs 262), slower on the 16 x 16 maps
# (720 channels: few pixel tiles, long contractions: 315 vs 262 us)
K19_MIN_PIXELS = int(_os.environ.get("MLAGG_K19_MIN_PIXELS", "1024"))


K19_WGRAD = _os.environ.get("MLAGG_K19_WGRAD", "1") == "1"
K19_WGRAD_MIN_PIXELS = int(_os.environ.get("MLAGG_K19_WGRAD_MIN_PIXELS", "1024"))
K19_WGRAD_MIN_CH = int(_os.environ.get("MLAGG_K19_WGRAD_MIN_CH", "48"))      # round 4: the 16-wide tiles fill 48-channel layers (was 96)


def _k19_wgrad(O, I, H, W, form=_DTYPE_BF16X3):
    """3 x 3 weight gradient on K19?  (16-bit operand forms: wherever the kernel runs -- one product per tap and block.)  Measured against MIOpen's implicit-GEMM kernels + their NHWC transposes (tools/bench_conv3x3.py):
    706 vs 816, 280 vs 323, 175 vs 205, 190 vs 212 us where a channel extent reaches 96 (32-channel tiles are then well filled);
    48 x 48 channels fill 56 % of a tile pair and lose or tie (492 vs 493, 179 vs 119 us), 16 x 16 maps tie."""
    if form != _DTYPE_BF16X3:
        return K19_WGRAD and H * W >= LP_K_MIN_PIXELS and bool(_lib.lib().mlagg_conv3x3_wgrad_supported(O, I, H, W))
    return (K19_WGRAD and H * W >= K19_WGRAD_MIN_PIXELS and max(O, I) >= K19_WGRAD_MIN_CH and
            bool(_lib.lib().mlagg_conv3x3_wgrad_supported(O, I, H, W)))


def _k19_product(O, I, H, W, form=_DTYPE_BF16X3):
    """forward-form 3 x 3 product (O output channels, contraction I) on K19?  (the data gradient asks with O and I exchanged)"""
    floor = K19_MIN_PIXELS if form == _DTYPE_BF16X3 else LP_K_MIN_PIXELS
    return K19 and H * W >= floor and bool(_lib.lib().mlagg_conv3x3_supported(O, I, H, W))


def _conv3x3_k19(x, xb, w, transposed, O, I, H, W, form=_DTYPE_BF16X3, out=None):
    lib = _lib.lib()
    B = x.shape[0]
    y = torch.empty(B, O, H, W, device=x.device, dtype=torch.float32) if out is None else out
    ws = torch.empty(lib.mlagg_conv3x3_workspace_bytes(O, I), device=x.device, dtype=torch.uint8)
    _flop("K19", 2 * 9 * B * O * I * H * W)
    _lib.check(lib.mlagg_conv3x3_fwd_lp(_ptr(x), xb, _ptr(w), int(transposed), None, _ptr(y), y.stride(0), _ptr(ws), B, O, I, H, W, form,
                                        _stream()), "mlagg_conv3x3_fwd_lp")
    return y


class Conv3x3Fn(torch.autograd.Function):
    """y = conv2d(x, W, padding=1) for a dense 3 x 3 kernel (stride 1, no bias): forward and data gradient on K19 (nine shifted
    split-bf16 GEMMs straight on the NCHW maps) where it beats the library's Winograd kernels, the weight gradient on the library."""

    @staticmethod
    def forward(ctx, x, weight, form=_DTYPE_BF16X3, slot=None):
        ctx.slot = slot
        x, xb, P = _planes(x, "x")
        B, I, H, W = x.shape
        O = weight.shape[0]
        w = _require(weight.contiguous(), "weight")
        if _k19_product(O, I, H, W, form):
            y = _conv3x3_k19(x, xb, w, False, O, I, H, W, form)
        else:
            y = _lib_conv_fwd(x, w, 1, form)
        ctx.save_for_backward(x, w)
        ctx.form = form
        ctx.leaf, ctx.leaf_params = _leaf_ok(weight), [weight]
        note_leaf_use(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, I, H, W = x.shape
        O = w.shape[0]
        form = ctx.form
        dx = dW = None
        if ctx.needs_input_grad[0]:
            if _k19_product(I, O, H, W, form):
                dy, dyb, _ = _planes(dy, "dy")
                out = ctx.slot.view() if ctx.slot is not None else None          # a piece of split_planes: written where the map's gradient lives
                dx = _conv3x3_k19(dy, dyb, w, True, I, O, H, W, form, out)
            else:
                dx = _lib_conv_bwd(dy, x, w, 1, (True, False, False), form)[0]
        if ctx.needs_input_grad[1]:
            lib = _lib.lib()
            if _k19_wgrad(O, I, H, W, form):
                dy, dyb, _ = _planes(dy, "dy")
                with _LeafStream(dy, x, ok=ctx.leaf and leaf_single_use(ctx.leaf_params)):
                    dW = torch.empty(O, I, 3, 3, device=x.device, dtype=torch.float32)
                    ws = torch.empty(lib.mlagg_conv3x3_wgrad_workspace_floats(B, O, I, H, W), device=x.device, dtype=torch.float32)
                    _flop("K19", 2 * 9 * B * O * I * H * W)
                    _lib.check(lib.mlagg_conv3x3_wgrad_lp(_ptr(dy), dyb, _ptr(x), x.stride(0), _ptr(dW), _ptr(ws), B, O, I, H, W, form,
                                                          _stream()), "mlagg_conv3x3_wgrad_lp")
            else:
                dyc = dy.contiguous()
                with _LeafStream(dyc, x, w, ok=ctx.leaf and leaf_single_use(ctx.leaf_params)):
                    dW = _lib_conv_bwd(dyc, x, w, 1, (False, True, False), form)[1]
        return dx, dW, None, None


def conv3x3_supported(x, weight, stride, padding, dilation, groups, form=_DTYPE_BF16X3):
    """A dense 3 x 3, stride-1, padding-1 convolution on an fp32 device map whose forward or data gradient runs on K19 (16-bit operand
    forms: or its weight gradient -- the one-channel stem)."""
    if not (K19 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and groups == 1):
        return False
    if tuple(weight.shape[2:]) != (3, 3) or any(int(v) != 1 for v in stride) or any(int(v) != 1 for v in padding) or \
            any(int(v) != 1 for v in dilation):
        return False
    O, I = int(weight.shape[0]), int(weight.shape[1])
    H, W = int(x.shape[2]), int(x.shape[3])
    return _k19_product(O, I, H, W, form) or _k19_product(I, O, H, W, form) or (form != _DTYPE_BF16X3 and _k19_wgrad(O, I, H, W, form))


def conv3x3(x, weight, form=_DTYPE_BF16X3):
    return Conv3x3Fn.apply(x, weight, form, _claim(x))


K19_3D = _os.environ.get("MLAGG_K19_3D", "1") == "1"
K19_3D_WGRAD = _os.environ.get("MLAGG_K19_3D_WGRAD", "1") == "1"


def _conv3x3x3_k19(x, xb, w, transposed, O, I, dims):
    lib = _lib.lib()
    B = x.shape[0]
    D, H, W = dims
    y = torch.empty(B, O, D, H, W, device=x.device, dtype=torch.float32)
    ws = torch.empty(lib.mlagg_conv3x3x3_workspace_bytes(O, I), device=x.device, dtype=torch.uint8)
    _lib.check(lib.mlagg_conv3x3x3_fwd(_ptr(x), xb, _ptr(w), int(transposed), None, _ptr(y), O * D * H * W, _ptr(ws), B, O, I, D, H, W,
                                       _stream()), "mlagg_conv3x3x3_fwd")
    return y


class Conv3x3x3Fn(torch.autograd.Function):
    """y = conv3d(x, W, padding=1) for a dense 3 x 3 x 3 kernel (stride 1, no bias): forward and data gradient on K19 (27 shifted
    split-bf16 GEMMs straight on the NCDHW volumes, no padded copy), weight gradient on K15 (its two padded copies are made in
    backward)."""

    @staticmethod
    def forward(ctx, x, weight):
        x, xb, P = _planes(x, "x")
        dims = tuple(int(v) for v in x.shape[2:])
        O, I = int(weight.shape[0]), int(weight.shape[1])
        w = _require(weight.contiguous(), "weight")
        y = _conv3x3x3_k19(x, xb, w, False, O, I, dims)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dims = tuple(int(v) for v in x.shape[2:])
        O, I = int(w.shape[0]), int(w.shape[1])
        dx = dW = None
        dy = dy.contiguous()
        lib = _lib.lib()
        if ctx.needs_input_grad[0]:
            if lib.mlagg_conv3x3x3_supported(I, O, *dims):
                dx = _conv3x3x3_k19(dy, O * dims[0] * dims[1] * dims[2], w, True, I, O, dims)
            else:                                               # contraction (the layer's output channels) not a multiple of 16
                dx = torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1, 1), (1, 1, 1), (1, 1, 1), False, (0, 0, 0), 1,
                                                         (True, False, False))[0]
        if ctx.needs_input_grad[1]:
            B = x.shape[0]
            if K19_3D_WGRAD and lib.mlagg_conv3x3x3_wgrad_supported(O, I, *dims):
                dW = torch.empty(O, I, 3, 3, 3, device=x.device, dtype=torch.float32)
                ws = torch.empty(lib.mlagg_conv3x3x3_wgrad_workspace_floats(B, O, I, *dims), device=x.device, dtype=torch.float32)
                _lib.check(lib.mlagg_conv3x3x3_wgrad(_ptr(dy), O * dims[0] * dims[1] * dims[2], _ptr(x), x.stride(0), _ptr(dW), _ptr(ws),
                                                     B, O, I, *dims, _stream()), "mlagg_conv3x3x3_wgrad")
            else:                                               # widths that are not multiples of 8: K15 on padded copies
                dW = conv_weight_grad(x if x.is_contiguous() else x.contiguous(), dy, 3, 1).view(w.shape)
        return dx, dW


# measured and NOT adopted (off): on every two-branch block 35.18 vs 35.10 ms, only where K18 runs the 1 x 1 data gradient anyway 35.14 vs 35.11
# (profiles/round4_m_conv_pair_ab.log) -- the accumulate's extra read of the map costs what the add_ kernel cost
CONV_PAIR = _os.environ.get("MLAGG_CONV_PAIR", "0") == "1"
CONV_PAIR_ALWAYS = _os.environ.get("MLAGG_CONV_PAIR_ALWAYS", "0") == "1"


class ConvPairFn(torch.autograd.Function):
    """(conv3x3(x, W3, padding 1), conv1x1(x, W1)) of ONE input map: conv1 and conv3 of a UnetResBlock whose channel count changes (MONAI
    structure behind T:1340-1368, M:581-667).  Forward products as in Conv3x3Fn / Conv1x1Fn; backward: K19 writes the 3 x 3 data gradient
    and K18 ADDS the 1 x 1 data gradient to it (mlagg_conv1x1_fwd_acc) -- autograd's add_ over the (B, I, H, W) map (114 us at 256 x 256)
    is gone, which pays for K18 on contractions it otherwise leaves to the library."""

    @staticmethod
    def forward(ctx, x, w3, w1, form=_DTYPE_BF16X3):
        x, xb, P = _planes(x, "x")
        B, I, H, W = x.shape
        O3, O1 = w3.shape[0], w1.shape[0]
        w3c = _require(w3.contiguous(), "w3")
        w1c = _require(w1.reshape(O1, I).contiguous(), "w1")
        c3 = _conv3x3_k19(x, xb, w3c, False, O3, I, H, W, form) if _k19_product(O3, I, H, W, form) else _lib_conv_fwd(x, w3c, 1, form)
        if _k18_product(O1, I, P, form):
            c1 = torch.empty(B, O1, H, W, device=x.device, dtype=torch.float32)
            _flop("K18", 2 * B * O1 * I * P)
            _conv1x1_k18(x, xb, w1c, c1, B, O1, I, P, form)
        else:
            c1 = _lib_conv_fwd(x, w1, 0, form)
        ctx.save_for_backward(x, w3c, w1c)
        ctx.form, ctx.w1shape = form, w1.shape
        ctx.leaf, ctx.leaf_params = _leaf_ok(w3, w1), [w3, w1]
        note_leaf_use(w3, w1)
        return c3, c1

    @staticmethod
    def backward(ctx, d3, d1):
        x, w3, w1 = ctx.saved_tensors
        B, I, H, W = x.shape
        O3, O1, P, form = w3.shape[0], w1.shape[0], H * W, ctx.form
        lib = _lib.lib()
        d3, d3b, _ = _planes(d3, "d3")
        d1, d1b, _ = _planes(d1, "d1")
        dx = dW3 = dW1 = None
        if ctx.needs_input_grad[0]:
            dx = _conv3x3_k19(d3, d3b, w3, True, I, O3, H, W, form) if _k19_product(I, O3, H, W, form) else \
                _lib_conv_bwd(d3, x, w3, 1, (True, False, False), form)[0].contiguous()
            if bool(lib.mlagg_conv1x1_supported(I, -(-O1 // 16) * 16, P)):
                wt = transpose_2d(w1.unsqueeze(0))[0]                                  # (I, O1)
                _flop("K18", 2 * B * O1 * I * P)
                _conv1x1_k18(d1, d1b, wt, dx, B, I, O1, P, form, accumulate=True)
            else:
                dx += _lib_conv_bwd(d1, x, w1.view(ctx.w1shape), 0, (True, False, False), form)[0]
        ok = ctx.leaf and leaf_single_use(ctx.leaf_params)
        if ctx.needs_input_grad[1]:
            if _k19_wgrad(O3, I, H, W, form):
                with _LeafStream(d3, x, ok=ok):
                    dW3 = torch.empty(O3, I, 3, 3, device=x.device, dtype=torch.float32)
                    ws = torch.empty(lib.mlagg_conv3x3_wgrad_workspace_floats(B, O3, I, H, W), device=x.device, dtype=torch.float32)
                    _flop("K19", 2 * 9 * B * O3 * I * P)
                    _lib.check(lib.mlagg_conv3x3_wgrad_lp(_ptr(d3), d3b, _ptr(x), x.stride(0), _ptr(dW3), _ptr(ws), B, O3, I, H, W, form,
                                                          _stream()), "mlagg_conv3x3_wgrad_lp")
            else:
                dW3 = _lib_conv_bwd(d3.contiguous(), x, w3, 1, (False, True, False), form)[1]
        if ctx.needs_input_grad[2]:
            with _LeafStream(d1, x, ok=ok):
                dW1 = torch.empty(O1, I, device=x.device, dtype=torch.float32)
                ws = torch.empty(lib.mlagg_conv1x1_wgrad_workspace_floats(B, O1, I, P), device=x.device, dtype=torch.float32)
                _flop("K18", 2 * B * O1 * I * P)
                _lib.check(lib.mlagg_conv1x1_wgrad_lp(_ptr(d1), d1b, _ptr(x), x.stride(0), _ptr(dW1), _ptr(ws), B, O1, I, P, form, _stream()),
                           "mlagg_conv1x1_wgrad_lp")
            dW1 = dW1.view(ctx.w1shape)
        return dx, dW3, dW1, None


def conv_pair_supported(x, conv3, conv1, form=_DTYPE_BF16X3):
    """Both convolutions of the pair are bias-free stride-1 layers K19 / K18 take (at least their weight gradients) on this map, and the
    input wants a gradient (else there is no sum to fuse)."""
    if not (CONV_PAIR and x.is_cuda and x.requires_grad and torch.is_grad_enabled()):
        return False
    if not (conv3.bias is None and conv1.bias is None
            and conv3x3_supported(x, conv3.weight, conv3.stride, conv3.padding, conv3.dilation, conv3.groups, form)
            and conv1x1_supported(x, conv1.weight, conv1.stride, conv1.padding, conv1.dilation, conv1.groups, form)):
        return False
    # only where K18 runs the 1 x 1 data gradient anyway: forcing it onto a 48-deep contraction (96 -> 48 at 256 x 256: 110 vs the
    # library's 88 us) cost more than the saved add_ (35.18 vs 35.10 ms, profiles/round4_m_conv_pair_ab.log)
    O1, I = int(conv1.weight.shape[0]), int(conv1.weight.shape[1])
    return CONV_PAIR_ALWAYS or _k18_product(I, O1, int(x.shape[2] * x.shape[3]), form)


def conv_pair(x, w3, w1, form=_DTYPE_BF16X3):
    return ConvPairFn.apply(x, w3, w1, form)


def conv3x3x3_supported(x, weight, stride, padding):
    if not (K19_3D and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and tuple(weight.shape[2:]) == (3, 3, 3)):
        return False
    if any(int(v) != 1 for v in stride) or any(int(v) != 1 for v in padding):
        return False
    return bool(_lib.lib().mlagg_conv3x3x3_supported(int(weight.shape[0]), int(weight.shape[1]), *(int(v) for v in x.shape[2:])))


def _pad_geometry(D, H, W, stride, wide=False):
    import ctypes
    Dq, Hq, Wq, guard = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_long()
    _lib.check(_lib.lib().mlagg_conv_pad_geometry(D, H, W, stride, int(wide), ctypes.byref(Dq), ctypes.byref(Hq), ctypes.byref(Wq),
                                                  ctypes.byref(guard)), "mlagg_conv_pad_geometry")
    return Dq.value, Hq.value, Wq.value, guard.value


# K15 for the 2-D network's dense convolutions: OFF.  Measured on the 21 convolution shapes of the 256 x 256 step
# (profiles/round3_conv_wgrad_2d_shapes_k15_vs_miopen.log): MIOpen's weight-gradient solvers are 1.1-1.9x faster on the 3x3 shapes
# (K15 incl. its two pad copies reaches 28-59 TFLOP/s there) and 4-7x on the 1x1 shapes (plain GEMMs); step 43.5 -> 56.2 ms with it
# on.  It pays where MIOpen has no tuned solver: the 3-D convolutions (8.2 s -> 66 ms per step at 2 x 96x160x160 voxels).
K15_2D = _os.environ.get("MLAGG_K15_2D", "0") == "1"
# K16 (forward / data gradient of the 3-D stride-1 convolutions on the tap-GEMM kernel) -- MLAGG_K16=0: MIOpen
K16 = _os.environ.get("MLAGG_K16", "1") == "1"


def conv_wgrad_supported(x, weight, stride, padding):
    """Kernel 3 with padding 1 or kernel 1 with padding 0, isotropic, stride 1 (2-D and 3-D) or 2 (3-D), fp32 device maps."""
    nd = x.dim() - 2
    k = weight.shape[2]
    return (x.is_cuda and x.dtype == torch.float32 and nd in (2, 3) and all(int(v) == k for v in weight.shape[2:]) and k in (1, 3)
            and all(int(v) == k // 2 for v in padding) and len(set(int(v) for v in stride)) == 1
            and (int(stride[0]) == 1 or (int(stride[0]) == 2 and nd == 3 and all(int(v) > 1 for v in x.shape[2:]))))


def conv_taps_supported(x, weight, stride, padding):
    """K16: 3-D, stride 1, last extent a multiple of 4, at least 8 input channels (a 1-channel stem would waste 31 / 32 of the MFMAs)."""
    return (K16 and x.dim() == 5 and conv_wgrad_supported(x, weight, stride, padding) and int(stride[0]) == 1 and x.shape[-1] % 4 == 0
            and x.shape[1] >= 8)


class _Padded:
    """A channel-major map copied into the zero-padded box of the tap-GEMM kernels (csrc/conv_wgrad.hip mlagg_volume_pad)."""

    def __init__(self, t, dims, stride, wide, as_output_of=None):
        import ctypes  # noqa: F401
        lib = _lib.lib()
        t = _require(t.contiguous(), "map")
        self.B, self.C = t.shape[:2]
        self.dims = tuple(dims)                                    # geometry of the convolution INPUT
        self.stride, self.wide = stride, wide
        self.Dq, self.Hq, self.Wq, self.guard = _pad_geometry(*self.dims, stride, wide)
        self.Q = self.Dq * self.Hq * self.Wq
        self.row = 2 * self.guard + self.Q
        self.nph = 1 if (stride == 1 or as_output_of is not None) else 8
        self.buf = torch.empty(self.B, self.nph, self.C, self.row, device=t.device, dtype=torch.float32)
        od = tuple(t.shape[2:]) if t.dim() == 5 else (1,) + tuple(t.shape[2:])
        if as_output_of is None:
            _lib.check(lib.mlagg_volume_pad(_ptr(t), _ptr(self.buf), self.B, self.C, *self.dims, stride, int(wide), 0, 0, 0, 0, _stream()),
                       "mlagg_volume_pad")
        else:
            _lib.check(lib.mlagg_volume_pad(_ptr(t), _ptr(self.buf), self.B, self.C, *self.dims, stride, int(wide), 1, *od, _stream()),
                       "mlagg_volume_pad")

    def ptr(self):
        return self.buf.data_ptr() + 4 * self.guard


def _tap_offsets(k, nd, stride, Hq, Wq, I=0, row=0):
    taps = []
    kz_range = range(k) if nd == 3 else (k // 2,)
    for kz in kz_range:
        for ky in range(k):
            for kx in range(k):
                if stride == 1:
                    dz = (kz - k // 2) if nd == 3 else 0
                    taps.append(dz * Hq * Wq + (ky - k // 2) * Wq + (kx - k // 2))
                else:
                    # padded input index 2 z + kz (k = 3) or 2 z + 1 (k = 1): parity phase + shift
                    az, ay, ax = (kz, ky, kx) if k == 3 else (1, 1, 1)
                    ph = ((az & 1) << 2) | ((ay & 1) << 1) | (ax & 1)
                    taps.append(ph * I * row + (az >> 1) * Hq * Wq + (ay >> 1) * Wq + (ax >> 1))
    return taps


def _wgrad_from_padded(xp, dyp, k, nd):
    """K15 on the padded copies of the input (xp) and of the output gradient (dyp, same box geometry): dW (O, I, k^nd)."""
    import ctypes
    lib = _lib.lib()
    O, I, B = dyp.C, xp.C, xp.B
    taps = _tap_offsets(k, nd, xp.stride, xp.Hq, xp.Wq, I, xp.row)
    ntaps = len(taps)
    Q8 = (xp.Q + 7) & ~7
    off = (ctypes.c_long * ntaps)(*taps)
    dW = torch.empty(O, I, ntaps, device=xp.buf.device, dtype=torch.float32)
    ws = torch.empty(lib.mlagg_conv_wgrad_taps_workspace_floats(B, Q8, O, I, ntaps), device=xp.buf.device, dtype=torch.float32)
    _lib.check(lib.mlagg_conv_wgrad_taps(dyp.ptr(), O * dyp.row, dyp.row, xp.ptr(), xp.nph * I * xp.row, xp.row, off, ntaps, Q8, O, I, B,
                                         _ptr(dW), 0, _ptr(ws), _stream()), "mlagg_conv_wgrad_taps")
    return dW


def conv_weight_grad(x, dy, k, stride):
    """dW (O, I, k^nd) of a convolution y = conv(x, W, stride, padding k // 2) from x (B, I, *dims) and dy (B, O, *out_dims)."""
    nd = x.dim() - 2
    dims = tuple(x.shape[2:]) if nd == 3 else (1,) + tuple(x.shape[2:])
    xp = _Padded(x, dims, stride, False)
    dyp = _Padded(dy, dims, stride, False, as_output_of=xp)
    return _wgrad_from_padded(xp, dyp, k, nd)


def _conv_taps(src, weight, O, I, k, flip, dims):
    """K16 on a padded copy `src` (wide stride-1 box): y (B, O, *dims) with weight (O', I', k^3) read as [o][i] (forward) or
    transposed with flipped taps (data gradient: O = the convolution's input channels)."""
    import ctypes
    lib = _lib.lib()
    ntaps = k ** 3
    taps = _tap_offsets(k, 3, 1, src.Hq, src.Wq)
    off = (ctypes.c_long * ntaps)(*taps)
    y = torch.empty(src.B, O, *dims, device=src.buf.device, dtype=torch.float32)
    w = _require(weight.contiguous(), "weight")
    if flip:      # weight (Cout_conv = I here, Cin_conv = O here, taps): output channel o -> stride ntaps, contraction i -> stride O * ntaps
        w_so, w_si = ntaps, O * ntaps
    else:
        w_so, w_si = I * ntaps, ntaps
    _lib.check(lib.mlagg_conv_taps(src.ptr(), src.C * src.row, src.row, _ptr(w), w_so, w_si, int(flip), off, ntaps, _ptr(y), src.B, O, I,
                                   *dims, _stream()), "mlagg_conv_taps")
    return y


class ConvTapsFn(torch.autograd.Function):
    """y = conv3d(x, W) (stride 1; kernel 3 pad 1 or kernel 1; no bias) entirely on this package's tap-GEMM kernels: K16 forward on a
    zero-padded copy of x (kept for backward), K16 data gradient on the padded copy of dy, K15 weight gradient on the two copies."""

    @staticmethod
    def forward(ctx, x, weight):
        dims = tuple(x.shape[2:])
        k = int(weight.shape[2])
        xp = _Padded(x, dims, 1, True)
        y = _conv_taps(xp, weight, weight.shape[0], weight.shape[1], k, False, dims)
        ctx.xp = xp
        ctx.save_for_backward(weight)
        ctx.meta = (dims, k)
        return y

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        dims, k = ctx.meta
        xp = ctx.xp
        dyp = _Padded(dy, dims, 1, True, as_output_of=xp)
        dx = dW = None
        if ctx.needs_input_grad[0]:
            dx = _conv_taps(dyp, weight, weight.shape[1], weight.shape[0], k, True, dims)
        if ctx.needs_input_grad[1]:
            dW = _wgrad_from_padded(xp, dyp, k, 3).view(weight.shape)
        ctx.xp = None
        return dx, dW


class ConvNdFn(torch.autograd.Function):
    """y = conv(x, W) (no bias) on channel-major maps: forward and the data gradient stay MIOpen's (library convolutions, the
    north star's "conv stem / decoder stages live in PyTorch-ROCm"); the weight gradient is K15."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        conv = torch.nn.functional.conv3d if x.dim() == 5 else torch.nn.functional.conv2d
        y = conv(x, weight, None, stride, padding)
        ctx.save_for_backward(x, weight)
        ctx.geom = (tuple(int(v) for v in stride), tuple(int(v) for v in padding))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, padding = ctx.geom
        nd = x.dim() - 2
        dx = dW = None
        dy = dy.contiguous()
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(dy, x, weight, None, stride, padding, (1,) * nd, False, (0,) * nd, 1,
                                                     (True, False, False))[0]
        if ctx.needs_input_grad[1]:
            dW = conv_weight_grad(x, dy, int(weight.shape[2]), stride[0]).view(weight.shape)
        return dx, dW, None, None


def conv_nd(x, weight, stride, padding):
    """Bias-free convolution; the tap-GEMM kernels when the shape is one they are built for (K16 + K15: 3-D stride 1; K15 weight
    gradient behind MIOpen's forward / data gradient: 3-D stride 2), plain torch otherwise."""
    if conv3x3x3_supported(x, weight, stride, padding):
        return Conv3x3x3Fn.apply(x, weight)
    if conv_taps_supported(x, weight, stride, padding):
        return ConvTapsFn.apply(x, weight)
    if conv_wgrad_supported(x, weight, stride, padding):
        return ConvNdFn.apply(x, weight, tuple(stride), tuple(padding))
    conv = torch.nn.functional.conv3d if x.dim() == 5 else torch.nn.functional.conv2d
    return conv(x, weight, None, stride, padding)
