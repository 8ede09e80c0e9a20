// Optional per-kernel HIP-event timing used by bench.py's roofline line (off by default: zero cost).
#pragma once
#include <hip/hip_runtime.h>

namespace mlagg_prof {

enum KernelId {
    K_SELSCAN_FWD_LOCAL = 0, K_SELSCAN_PREFIX, K_SELSCAN_FWD_FINAL, K_SELSCAN_BWD_LOCAL, K_SELSCAN_BWD,
    K_SELSCAN_REDUCE, K_LOCAL_FWD, K_LOCAL_BWD_A, K_LOCAL_BWD_B, K_POOLED_FWD,
    K_POOLED_BWD1, K_POOLED_BWD2, K_DWCONV_FWD, K_DWCONV_BWD_DATA, K_DWCONV_BWD_WEIGHT, K_LINEAR_WGRAD, K_LAYERNORM_FWD, K_LAYERNORM_BWD, K_DWCONV_NCHW_FWD, K_DWCONV_NCHW_BWD, K_CROSS_SCAN, K_CROSS_MERGE, K_GATE_FWD, K_GATE_BWD, K_LINEAR_FWD, K_LINEAR_DGRAD, K_ROW_SCALE, K_LOSS_STATS, K_LOSS_GRAD, K_TRANSPOSE, K_BIAS_GRAD, K_PLANE_NORM_FWD, K_PLANE_NORM_BWD, K_ADAMW, K_FLASH_FWD, K_FLASH_BWD, K_CHANNEL_EPI,
    K_SEL1_FWD_LOCAL_R2, K_SEL1_FWD_FINAL_R2, K_SEL1_BWD_LOCAL_R2, K_SEL1_BWD_R2, K_SEL1_FWD_LOCAL, K_SEL1_FWD_FINAL, K_SEL1_BWD_LOCAL,
    K_SEL1_BWD, K_SEL1_PREFIX, K_SEL1_REDUCE, K_CONV_PAD, K_CONV_WGRAD, K_CONV_WGRAD_REDUCE, K_CONV_TAPS, K_GELU_POOL, K_CONV1X1, K_CONV3X3, K_TOK_FWD_LOCAL, K_TOK_FWD_FINAL, K_TOK_BWD_LOCAL, K_TOK_BWD_GROUP, K_COUNT
};

extern int g_selected;                 // -1: off, -2: every kernel, else one KernelId
void record(int id, hipStream_t st, bool begin);

struct Scope {
    int id;
    hipStream_t st;
    bool on;
    Scope(int id_, hipStream_t st_) : id(id_), st(st_), on(g_selected == -2 || g_selected == id_)
    {
        if (on) record(id, st, true);
    }
    ~Scope()
    {
        if (on) record(id, st, false);
    }
};

}  // namespace mlagg_prof

#define MLAGG_TIMED(id, st) mlagg_prof::Scope mlagg_prof_scope_##id(mlagg_prof::id, st)
