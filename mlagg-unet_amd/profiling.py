"""Host side of the library's per-kernel HIP-event timers + the algorithmic byte counts the roofline
line of bench.py prices each kernel with (derivations in DESIGN.md)."""
import ctypes

from . import _lib


def _names():
    lib = _lib.lib()
    return [lib.mlagg_profile_kernel_name(i).decode() for i in range(lib.mlagg_profile_kernel_count())]


def select_all():
    _lib.check(_lib.lib().mlagg_profile_select(-2), "mlagg_profile_select")


def select(name):
    idx = -1 if name is None else _names().index(name)
    _lib.check(_lib.lib().mlagg_profile_select(idx), "mlagg_profile_select")


def collect():
    """{kernel name: {"ms": summed milliseconds, "count": launches}} since the last collect()."""
    lib = _lib.lib()
    n = lib.mlagg_profile_kernel_count()
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_int * n)()
    _lib.check(lib.mlagg_profile_collect(ms, cnt), "mlagg_profile_collect")
    return {nm: {"ms": ms[i], "count": cnt[i]} for i, nm in enumerate(_names()) if cnt[i] > 0}


def algorithmic_bytes(kernel, batch, img):
    """Bytes ONE launch of `kernel` must move for `batch` images of size img (fp32, D = 384 scan channels,
    N = 16 states, G = 4 directions; C * tokens = 96 * 128 * 128 at every encoder stage)."""
    H, W = img
    n0 = (H // 2) * (W // 2)                      # stage-0 tokens
    l_cat = sum(n0 >> (2 * i) for i in range(4))  # 21760 at 256x256
    D, N, G, R = 384, 16, 4, 3
    # The model runs K1 in its low-rank form (delta is formed in-kernel from the rank-3 rows): delta and its
    # gradient are not op-boundary tensors any more, so the SURVEY 8(d) figures 5120 L / 8704 L (which count them)
    # shrink to 3632 L forward and 5728 L backward.
    scan = {
        "selscan_fwd_kernel<false>": 4 * l_cat * (D + G * R + G * N),                    # read u, dtr, B
        "selscan_fwd_kernel<true>": 4 * l_cat * (2 * D + G * R + 2 * G * N),             # K1 fwd op boundary (3632 L)
        "selscan_bwd_local_kernel": 4 * l_cat * (D + G * R + G * N),                     # read dtr, dy, C
        "selscan_bwd_group_kernel": 4 * l_cat * (3 * D + 2 * G * R + 4 * G * N),         # K1 bwd op boundary (5728 L)
        # K1f (token-major, what the model runs since round 4): per direction and token u / dy / du are 96 floats, a projection
        # row 36 (3 rank values + pad, B, C); every direction reads u and dy for itself and writes its own output block
        "tok_fwd_kernel<false>": 4 * l_cat * G * (96 + 4 + N),                            # read u, rank row, B (1856 L)
        "tok_fwd_kernel<true>": 4 * l_cat * G * (96 + 36 + 96),                           # read u, projection row; write y_k (3648 L)
        "tok_bwd_local_kernel": 4 * l_cat * G * (96 + 4 + N),                             # read dy, rank row, C
        "tok_bwd_group_kernel": 4 * l_cat * G * (3 * 96 + 2 * 36),                        # read u, dy, row; write du_k, d(row) (5760 L)
    }
    if kernel in scan:
        return scan[kernel] * batch
    # (C/2) * N_tokens per attention module HALVES from stage to stage (C doubles, the token count quarters):
    # 48 n0, 24 n0, 12 n0, 6 n0.  These kernels run once per block at each of the 4 stages and their timers average over
    # all launches, so the figure priced against that average is the mean over the four stage shapes.
    dn = sum(48 * n0 / 2 ** i for i in range(4)) / 4
    per_module = {
        "local_attn_fwd_kernel": 16 * dn,          # q, k, v in; out
        "local_attn_bwd_a_kernel": 20 * dn,        # q, k, v, dout in; dq out
        "local_attn_bwd_b_kernel": 16 * dn,        # q, dout in; dk, dv out
        "pooled_attn_fwd_kernel": 8 * dn,          # q in, out (pooled K/V negligible)
        "pooled_attn_bwd1_kernel": 16 * dn,        # q, dout, o_pre in; dq out
        "pooled_attn_bwd2_kernel": 8 * dn,         # q, d(o) in
    }
    # K2 / K2n / K5w / K6 / K1' are launched with several shapes per step (different callers): they have no
    # single bytes-per-launch figure and are reported by time only
    return int(per_module.get(kernel, 0) * batch)


def algorithmic_bytes_sel1(kernel, batch, tokens, C=64, K=12, R=2):
    """Bytes one launch of a K1s kernel must move at the widest stage of the 3-D network (stage 0: `tokens` = D*H*W of the patch,
    C = d_inner 64, K = 12 directions, rank 2; only that stage runs the R = 2 instantiations, so these kernel names see one shape
    per step).  Op-boundary counting in the sense of SURVEY 8(d)'s "fully fused lower bound": u is read once per direction, the
    MERGED y / du is written once (the kernels as built write all K direction blocks and a block sum merges them: that extra
    traffic counts against them), per-step parameters are idx + B + C + R rank rows."""
    par = 4 * (3 + R)
    per_token = {
        "sel1_fwd_kernel<2, false>": K * (4 * C + par),                                  # read u, params
        "sel1_fwd_kernel<2, true>": K * (4 * C + par) + 4 * C,                           # read u, params; write merged y
        "sel1_bwd_local_kernel<2>": K * (4 * C + par),                                   # read dy, params
        "sel1_bwd_kernel<2>": K * (8 * C + par + 4 * (2 + R)) + 4 * C,                   # read u, dy, params; write dB, dC, ddtr, merged du
    }
    return int(per_token.get(kernel, 0)) * tokens * batch
