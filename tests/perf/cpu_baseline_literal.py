"""BASELINE.md section 4, figure (a): one CPU train step of BASELINE configs[0] (batch 2, 128x128, 14 classes, variant B)
with the scan loop written exactly as mamba-ssm's selective_scan_ref writes it (`deltaA[:, :, i]` indexing, O(L^2)
backward), beside figure (b), the `unbind` loop the bench's cpu_baseline uses.  CPU only; takes ~10-20 minutes.
    python tests/perf/cpu_baseline_literal.py [--threads 8] > profiles/roundN_cpu_literal.json.log
(under tests/: oracle code is test infrastructure and is only run from there, from smoke() and from bench.py's cpu_baseline)
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import mlagg_oracle as O  # noqa: E402


def step_seconds(literal, timed):
    O.SCAN_LITERAL_INDEXING = literal
    torch.manual_seed(0)
    net = O.build_reference_config_model((128, 128), 1, 14, True, "B").train()
    opt = O.make_optimizer(net)
    data, target = O.synthetic_batch(2, 1, 128, 128, 14, seed=1234)
    first = None
    if not literal:
        first = float(O.train_step(net, opt, data, target))        # warm-up
    t0 = time.perf_counter()
    for _ in range(timed):
        loss = float(O.train_step(net, opt, data, target))
        first = loss if first is None else first
    return (time.perf_counter() - t0) / timed, first


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(16, len(os.sched_getaffinity(0))))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    fair, loss_b = step_seconds(False, 3)
    literal, loss_a = step_seconds(True, 1)                        # one step, no warm-up: it is ~10 minutes long
    print(json.dumps({"workload": "BASELINE configs[0]: batch 2, 128x128x1, 14 classes, variant B, CPU eager oracle train step",
                      "threads": a.threads,
                      "unbind_loop_s_per_step": round(fair, 2), "unbind_loop_images_per_s": round(2 / fair, 4),
                      "literal_indexing_s_per_step": round(literal, 1), "literal_indexing_images_per_s": round(2 / literal, 5),
                      "first_loss_equal": abs(loss_a - loss_b) < 1e-3 * max(1.0, abs(loss_b))}))


if __name__ == "__main__":
    main()
