// Launchers shared between translation units of libmlagg_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace mlagg_internal {
size_t dwconv_wgrad_workspace_floats(int batch, int H, int W, int C);
// dw (C, 9) and dbias (C, may be NULL) are ACCUMULATED into; part = workspace of the size above
void dwconv_wgrad_launch(const float *x, int x_stride, const float *dy, int dy_stride, const float *pre, float *dw,
                         float *dbias, float *part, int batch, int H, int W, int C, int silu, hipStream_t st);
}  // namespace mlagg_internal
