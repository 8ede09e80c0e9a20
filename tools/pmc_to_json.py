#!/usr/bin/env python
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (separate passes) into the per-kernel HBM
traffic table bench.py reads for `roofline.traffic`.

gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE counts the 128-byte requests of wide
(16 B/lane) coalesced reads as 64 bytes -> doubled; WRITE_SIZE is exact for 16-byte stores.  Both are
reported in KiB.

    python tools/pmc_to_json.py gpurun_out/pmc2_FETCH_SIZE gpurun_out/pmc2_WRITE_SIZE > profiles/pmc_traffic.json
"""
import collections
import csv
import glob
import json
import re
import sys


# rocprof's template spellings of the low-rank scan kernels -> the library's profile names (bench.py looks these up)
LIB_NAMES = {"selscan_fwd_kernel<false, true>": "selscan_fwd_kernel<false>", "selscan_fwd_kernel<true, true>": "selscan_fwd_kernel<true>",
             "selscan_bwd_kernel<true>": "selscan_bwd_kernel", "selscan_bwd_local_kernel<true>": "selscan_bwd_local_kernel",
             "selscan_bwd_group_kernel": "selscan_bwd_group_kernel"}     # round 2: the group-per-wave form serves the same entry point


def lib_name(base, targs):
    """rocprof's kernel name -> the library's profile name (bench.py's keys): the scan kernels keep only the template
    argument that distinguishes passes (forward: FINAL), every other template argument is a shape specialisation."""
    args = [a.strip() for a in targs.strip("<>").split(",")] if targs else []
    if base == "selscan_fwd_kernel":
        return f"selscan_fwd_kernel<{args[0]}>" if args else base
    if base in ("selscan_bwd_local_kernel", "selscan_bwd_kernel"):
        return base
    if base == "tok_fwd_kernel":
        return f"tok_fwd_kernel<{args[0]}>" if args else base
    if base in ("tok_bwd_local_kernel", "tok_bwd_group_kernel"):
        return base
    if base == "selscan_bwd_group_kernel":
        return "selscan_bwd_group_kernel"                       # round 2: the group-per-wave form serves the same entry point
    name = base + (targs or "")
    return LIB_NAMES.get(name, name)


def load(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"([a-z_0-9]+_kernel)(<[^>]*>)?|(selscan_[a-z_]+)", r["Kernel_Name"])
            if m:
                out[lib_name(m.group(1) or m.group(3), m.group(2) or "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


res = {}
for d in sys.argv[1:]:
    for k, v in load(d).items():
        for c, vals in v.items():
            res.setdefault(k, {})[c] = sum(vals) / len(vals)
kern = {}
for k, v in sorted(res.items()):
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        fetch, write = 2 * v["FETCH_SIZE"] * 1024, v["WRITE_SIZE"] * 1024
        kern[k] = {"fetch_bytes_corrected": round(fetch), "write_bytes": round(write), "traffic_bytes": round(fetch + write)}
import os
WORK = {"scanlr": "tools/bench_ops.py scanlr: B=10, D=384, N=16, G=4, R=3, L=21760 (BASELINE config 2 MSMM scan, (B, D, L) low-rank form)",
        "msmm": "tools/bench_ops.py msmm: B=10, four directions x 96 channels, N=16, R=3, L_cat=21760 (BASELINE config 2 MSMM scan, token-major K1f)"}
json.dump({"workload": WORK.get(os.environ.get("SCAN_OP", "msmm"), os.environ.get("SCAN_OP", "msmm")),
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, per-dispatch averages; "
                     "FETCH_SIZE x2 (gfx950 counts wide coalesced reads at half; an upper bound where a kernel issues narrower loads), both KiB -> bytes",
           "kernels": kern}, sys.stdout, indent=1)
