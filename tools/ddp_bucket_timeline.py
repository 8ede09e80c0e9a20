#!/usr/bin/env python
"""Where in the backward pass the data-parallel gradient buckets are handed to the collective: from a rocprofv3 --kernel-trace CSV of
`MLAGG_FORCE_DDP=1 python bench.py` (one RCCL rank: the only form of the `nccl` path a one-GPU box can run -- RCCL launches no
kernel for a one-rank all-reduce, so what the trace can show is the ENQUEUE point of every bucket: its multi-tensor gather kernel,
issued from the post-accumulate-grad hook of the bucket's last gradient, directly in front of `dist.all_reduce(async_op=True)`).

    python tools/ddp_bucket_timeline.py trace.csv --steps 5 > profiles/roundN_ddp_bucket_timeline.md
A step runs from the loss gradient kernel (dice_ce_grad_kernel, first kernel of backward) to the optimizer (adamw_update_kernel)."""
import argparse
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
upd = [i for i, r in enumerate(rows) if "adamw_update_kernel" in r["Kernel_Name"]]
print("# Gradient-bucket enqueue points inside the backward pass (single-rank RCCL run)\n")
print("One line per bucket gather (`multi_tensor_apply_kernel<... copy ...>` issued by trainer.BucketedGradSync's hook); times in ms from the "
      "first backward kernel of the step; `backward done` = start of the clip / AdamW kernels.\n")
for s in range(max(1, len(upd) - a.steps), len(upd)):
    lo, hi = upd[s - 1], upd[s]
    seg = rows[lo + 1:hi + 1]
    first = next((i for i, r in enumerate(seg) if "dice_ce_grad_kernel" in r["Kernel_Name"]), None)
    if first is None:
        continue
    t0 = int(seg[first]["Start_Timestamp"])
    bwd = seg[first:]
    sumsq = next((r for r in bwd if "adamw_sumsq_kernel" in r["Kernel_Name"]), bwd[-1])
    t_end = (int(sumsq["Start_Timestamp"]) - t0) / 1e6
    launches = [(j, r) for j, r in enumerate(bwd) if re.search(r"multi_tensor_apply_kernel", r["Kernel_Name"]) and
                re.search(r"[Cc]opy", r["Kernel_Name"]) and int(r["Start_Timestamp"]) < int(sumsq["Start_Timestamp"])]
    # a bucket of many tensors is gathered by several back-to-back launches of the multi-tensor copy: one line per bucket
    gathers = [(j, r) for n, (j, r) in enumerate(launches) if n == 0 or launches[n - 1][0] != j - 1]
    print(f"## step {s}: backward {t_end:.2f} ms, {len(bwd)} kernels, {len(gathers)} buckets ({len(launches)} gather launches)\n")
    print("| bucket | gather starts at (ms) | % of backward elapsed | kernel in front of it | kernels of backward still to run |\n|---|---|---|---|---|")
    for b, (j, r) in enumerate(gathers):
        t = (int(r["Start_Timestamp"]) - t0) / 1e6
        prev = re.sub(r"\s+", " ", bwd[j - 1]["Kernel_Name"])[:70] if j else "-"
        left = sum(1 for q in bwd[j + 1:] if int(q["Start_Timestamp"]) < int(sumsq["Start_Timestamp"]))
        print(f"| {b} | {t:.2f} | {100 * t / t_end:.0f} | `{prev}` | {left} |")
    print()
