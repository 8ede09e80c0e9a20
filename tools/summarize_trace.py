#!/usr/bin/env python
"""Reduce a rocprofv3 --kernel-trace CSV of `bench.py` to per-kernel statistics of the TIMED steps only.

The whole-process --stats table also contains MIOpen's first-use solver search (naive reference
convolutions of tens of milliseconds) and the untimed warm-up, so it cannot be compared with bench.py's
ms_per_step.  A train step is delimited by consecutive dispatches of the scan's backward group kernel (`tok_bwd_group_kernel` / `selscan_bwd_group_kernel`: exactly one
per step); the last `--steps` intervals are the timed region.

    python tools/summarize_trace.py gpurun_out/prof/runc/*_kernel_trace.csv --steps 10 > profiles/...md
"""
import argparse
import collections
import csv
import re


def family(n):
    if n.startswith("Cijk_"):
        return "GEMM (rocBLAS/hipBLASLt)"
    for key, lab in (("linear_lp", "HIP K5 projections (fwd, dx), 16-bit operands"), ("selscan", "HIP K1 selective scan"), ("tok_fwd", "HIP K1 selective scan"), ("tok_bwd", "HIP K1 selective scan"), ("sel1_", "HIP K1s one-state selective scan (3-D)"),
                     ("conv_taps_kernel", "HIP K16 convolution forward / data gradient (tap GEMM)"),
                     ("dwconv", "HIP K2 depthwise conv"), ("gelu_pool", "HIP K17 GELU + window mean"),
                     ("conv1x1_", "HIP K18 1x1 convolution (split bf16)"), ("conv3x3_", "HIP K19 3x3 convolution (split bf16)"),
                     ("linear_wgrad_x3", "HIP K5w linear weight-grad"), ("linear_x3_kernel", "HIP K5 projections (fwd, dx)"),
                     ("weight_image", "HIP K5 projections (fwd, dx)"), ("linear_lp_kernel<true, 2>", "HIP K5 projections (fwd, dx)"),
                     ("linear_lp_kernel<false, 2>", "HIP K5 projections (fwd, dx)"),
                     ("conv_wgrad_", "HIP K15 convolution weight gradient (tap GEMM)"), ("volume_pad_kernel", "HIP K15/K16 padded copies"),
                     ("guard_zero_kernel", "HIP K15/K16 padded copies"), ("pooled_lp_", "HIP K4lp pooled diff-attention (16-bit MFMA)"),
                     ("plane_split_", "HIP K10 plane norm + activation"), ("channel_epilogue", "HIP K13 convolution epilogue"),
                     ("channel_gelu", "HIP K13 convolution epilogue"),
                     ("index_scan_kernel", "HIP K14 index scan / merge"), ("block_sum_kernel", "HIP K14 index scan / merge"), ("cross_scan_kernel", "HIP K1' cross-scan / merge"),
                     ("dwconv", "HIP K2 depthwise conv"),
                     ("local_attn", "HIP K3 local diff-attention"), ("pooled_attn", "HIP K4 pooled diff-attention"),
                     ("linear_wgrad", "HIP K5w linear weight-grad"), ("linear_mfma", "HIP K5 projections (fwd, dx)"),
                     ("layernorm_", "HIP K6 LayerNorm"), ("column_sum", "HIP K6/K8 column sums"), ("gate_", "HIP K7 gate"),
                     ("row_scale", "HIP K8 small ops"), ("diff_lambda", "HIP K8 small ops"),
                     ("transpose_tile", "HIP K8 small ops"), ("plane_sum", "HIP K8 small ops"),
                     ("dice_ce", "HIP K9 fused loss"), ("plane_norm", "HIP K10 plane norm + activation"),
                     ("adamw_", "HIP K11 clip + AdamW")):
        if key in n:
            return lab
    if re.search(r"miopen|igemm|naive_conv|Im2d2Col|Col2Im|batched_transpose|gridwise|conv|Conv|SubTensor|transpose_", n):
        return "MIOpen convolution (+layout)"
    if re.search(r"layer_norm|GammaBeta|GradInput|RowwiseMoments|batch_norm|BatchNorm|GroupNorm|ComputeInternalGradients|FusedParams", n):
        return "norms (LN/GN/IN)"
    if re.search(r"elementwise|FillFunctor|fillBuffer|copyBuffer", n):
        return "elementwise / fill / copy"
    if "reduce_kernel" in n:
        return "reductions"
    if re.search(r"multi_tensor|Optim|lpnorm", n):
        return "optimizer / clip"
    if re.search(r"softmax|nll|scatter", n, re.I):
        return "loss"
    if re.search(r"CatArray|index|flip", n, re.I):
        return "cat / index / flip"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--aten", action="store_true", help="append every at::native / MIOpen helper kernel of the step (the library tail)")
    ap.add_argument("--mark", default=r"(selscan|tok)_bwd_(group_)?kernel",
                    help="regex of the kernel dispatched exactly once per step (3-D network: 'sel1_bwd_kernel<2>')")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if re.search(a.mark, r["Kernel_Name"])]
    if len(marks) < a.steps + 1:
        raise SystemExit(f"only {len(marks)} steps in the trace")
    lo, hi = marks[-a.steps - 1], marks[-1]          # [bwd-scan of step k-1, bwd-scan of the last step)
    sel = rows[lo:hi]
    wall = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6 / a.steps
    per = collections.defaultdict(list)
    for r in sel:
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    fam = collections.defaultdict(lambda: [0.0, 0])
    for k, v in per.items():
        fam[family(k)][0] += sum(v) / 1e3 / a.steps
        fam[family(k)][1] += len(v) / a.steps
    busy = sum(v[0] for v in fam.values())
    print(f"# rocprofv3 kernel trace, timed region only ({a.steps} steps)\n")
    print(f"wall per step between step markers: {wall:.2f} ms; sum of kernel durations per step: {busy:.2f} ms\n")
    print("| kernel family | ms / step | % of kernel time | launches / step |\n|---|---|---|---|")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"| {k} | {v[0]:.2f} | {100 * v[0] / busy:.1f} | {v[1]:.0f} |")
    print("\n| kernel | calls / step | avg us | min us | max us | ms / step |\n|---|---|---|---|---|---|")
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:a.top]:
        name = re.sub(r"\s+", " ", k)[:110]
        print(f"| `{name}` | {len(v) / a.steps:.1f} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} | "
              f"{sum(v) / 1e3 / a.steps:.3f} |")
    if a.aten:
        print("\n## Library tail (at::native kernels and MIOpen's helper kernels), every kernel\n")
        print("| kernel | calls / step | avg us | ms / step |\n|---|---|---|---|")
        tot = 0.0
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            if not re.search(r"at::native|at::cuda|SubTensorOp|batched_transpose", k):
                continue
            m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda|\w+KernelImpl|CatArrayBatchedCopy\w*|reduce_kernel|SubTensorOp\w*|batched_transpose\w*"
                          r"|MeanOps|sum_functor|bernoulli\w*|uniform\w*|where_kernel\w*|clamp\w*|compare\w*)", k)
            lab = (m.group(1) if m else k[:60]) + (" [strided]" if "elementwise_kernel_manual_unroll" in k or "unrolled_elementwise" in k else "")
            tot += sum(v) / 1e3 / a.steps
            print(f"| `{lab}` | {len(v) / a.steps:.1f} | {sum(v) / len(v):.1f} | {sum(v) / 1e3 / a.steps:.3f} |")
        print(f"\ntotal: {tot:.2f} ms / step")


if __name__ == "__main__":
    main()
