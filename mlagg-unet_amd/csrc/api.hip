// Library-level entry points of include/mlagg_hip.h: version, error strings, kernel timing.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "mlagg_hip.h"
#include "prof.h"

namespace mlagg_prof {

int g_selected = -1;

namespace {
struct Rec {
    int id;
    hipEvent_t start, stop;
};
std::mutex g_mu;
std::vector<Rec> g_recs;         // event pairs, reused across collect() calls
size_t g_used = 0;
const char *const kNames[K_COUNT] = {
    "selscan_fwd_kernel<false>", "selscan_chunk_prefix", "selscan_fwd_kernel<true>", "selscan_bwd_local_kernel",
    "selscan_bwd_group_kernel", "selscan_reduce_partials", "local_attn_fwd_kernel", "local_attn_bwd_a_kernel",
    "local_attn_bwd_b_kernel", "pooled_attn_fwd_kernel", "pooled_attn_bwd1_kernel",
    "pooled_attn_bwd2_kernel", "dwconv_tiled_kernel<fwd>", "dwconv_tiled_kernel<dgrad>", "dwconv_bwd_weight_kernel", "linear weight gradient (K5w: linear_wgrad_kernel / linear_wgrad_x3_kernel + reduce)", "layernorm_fwd_kernel",
    "layernorm_bwd_kernel", "dwconv_nchw_fwd_kernel", "dwconv_nchw_bwd (data+weight+reduce)",
    "cross_scan_kernel<false>", "cross_scan_kernel<true>", "gate_fwd_kernel", "gate_bwd_kernel",
    "linear forward + data gradient (K5: linear_x3_kernel on weight images; MLAGG_K5_V2=0: linear_lp_kernel<true, MODE> / linear_mfma_kernel<true>)", "linear data gradient (K5: linear_mfma_kernel<false> / linear_lp_kernel<false, MODE>)", "row_scale_kernel", "dice_ce_stats_kernel", "dice_ce_grad_kernel", "transpose_tile_kernel", "bias gradient (plane_sum / column_sum)", "plane_norm_fwd_kernel", "plane_norm_bwd_kernel", "adamw (sumsq + update)", "flash_fwd_kernel", "flash_bwd (dq + dk/dv)", "channel_epilogue (bias + residual + GELU)",
    "sel1_fwd_kernel<2, false>", "sel1_fwd_kernel<2, true>", "sel1_bwd_local_kernel<2>", "sel1_bwd_kernel<2>",
    "sel1_fwd_kernel<R != 2, false>", "sel1_fwd_kernel<R != 2, true>", "sel1_bwd_local_kernel<R != 2>", "sel1_bwd_kernel<R != 2>",
    "sel1_prefix_kernel", "sel1 reductions (step partials + per-chunk rows)",
    "volume_pad_kernel (+ guard fill)", "conv_wgrad_taps_kernel", "conv_wgrad_reduce_kernel", "conv_taps_kernel (forward / data gradient)",
    "gelu_pool (forward + backward)", "conv1x1 (forward / data gradient / weight gradient)",
    "conv3x3 (forward / data gradient / weight gradient, incl. weight image and reduce)",
    "tok_fwd_kernel<false>", "tok_fwd_kernel<true>", "tok_bwd_local_kernel", "tok_bwd_group_kernel"};
}  // namespace

// begin/end pairs of one kernel are issued back to back from one host thread (the launcher), so the
// open record is always the last one.
void record(int id, hipStream_t st, bool begin)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (begin) {
        if (g_used == g_recs.size()) {
            Rec r{id, nullptr, nullptr};
            if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
            g_recs.push_back(r);
        }
        g_recs[g_used].id = id;
        (void)hipEventRecord(g_recs[g_used].start, st);
        ++g_used;
    } else if (g_used > 0) {
        (void)hipEventRecord(g_recs[g_used - 1].stop, st);
    }
}

}  // namespace mlagg_prof

extern "C" const char *mlagg_version(void) { return "mlagg_hip 0.1 (gfx950)"; }

extern "C" const char *mlagg_error_string(int code)
{
    switch (code) {
    case 0: return "success";
    case MLAGG_E_UNSUPPORTED: return "unsupported shape for the gfx950 kernels";
    case MLAGG_E_NULLPTR: return "required pointer argument is NULL";
    case MLAGG_E_WORKSPACE: return "workspace too small";
    default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown mlagg error";
    }
}

extern "C" int mlagg_profile_kernel_count(void) { return mlagg_prof::K_COUNT; }

extern "C" const char *mlagg_profile_kernel_name(int id)
{
    return (id >= 0 && id < mlagg_prof::K_COUNT) ? mlagg_prof::kNames[id] : "?";
}

extern "C" int mlagg_profile_select(int id)
{
    if (id < -2 || id >= mlagg_prof::K_COUNT) return MLAGG_E_UNSUPPORTED;
    mlagg_prof::g_selected = id;
    return 0;
}

// Synchronises with the recorded events (NOT graph-capturable, bench/diagnostics only), adds the
// elapsed milliseconds and launch counts per kernel id into ms[K_COUNT] / counts[K_COUNT], and resets.
extern "C" int mlagg_profile_collect(double *ms, int *counts)
{
    using namespace mlagg_prof;
    if (!ms || !counts) return MLAGG_E_NULLPTR;
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < K_COUNT; ++i) { ms[i] = 0.0; counts[i] = 0; }
    for (size_t i = 0; i < g_used; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(g_recs[i].stop) != hipSuccess) continue;
        if (hipEventElapsedTime(&t, g_recs[i].start, g_recs[i].stop) != hipSuccess) continue;
        ms[g_recs[i].id] += t;
        counts[g_recs[i].id] += 1;
    }
    g_used = 0;
    return 0;
}
