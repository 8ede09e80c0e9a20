#!/usr/bin/env python
"""Headline benchmark: train-step images/sec of MLAgg-UNet 2D (nnUNetTrainer_MLAgg_2D_dt_MS) on
synthetic AbdomenMRI-shaped batches -- BASELINE.json config 2: 256x256x1, batch 10 per GPU, fp32,
14 classes, attention variant B (the fp32 path, reference T:762-777).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = zero_grad + forward + Dice/CE deep-supervision loss + backward + clip_grad_norm_(12) + AdamW
(reference nnUNetTrainer.py:833-863), inputs resident in HBM before the timed region
(nnUNetTrainerBenchmark_5epochs_noDataLoading.py:16-22).  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

IMG = (256, 256)
BATCH_PER_GPU = 10
N_CLASSES = 14
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BF16_MFMA_PEAK_TFS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
FP32_MFMA_PEAK_TFS = 157.3   # v_mfma_f32_32x32x2_f32: what an fp32 GEMM on the fp32 matrix instruction is bounded by
# this package's matrix kernels by their profile name; the fp32 products run as SIX bf16 MFMAs each (csrc/bf16x3.h), so the ceiling
# of real fp32 flop is a sixth of the bf16 peak
MFMA_FAMILIES = {"K5": "linear forward + data gradient (K5", "K5w": "linear weight gradient (K5w", "K18": "conv1x1 (", "K19": "conv3x3 ("}
# BASELINE.json configs by their 1-based position; 2 is the headline (the default).  The others print the same JSON line
# for their own shape / arithmetic type and are never the headline number.
CONFIGS = {
    2: dict(img=(256, 256), in_ch=1, classes=14, batch=10, variant="B", precision="fp32", dtype="f32",
            name="AbdomenMRI-shaped 256x256x1, 14 classes, attention variant B (BASELINE.json configs[1])"),
    3: dict(img=(224, 224), in_ch=1, classes=4, batch=10, variant="B", precision="bf16", dtype="bf16",
            name="ACDC-shaped 224x224x1, 4 classes, attention variant B, bf16 operands / fp32 sums (BASELINE.json configs[2])"),
    4: dict(img=(96, 160, 160), in_ch=1, classes=14, batch=2, variant=None, precision="fp32", dtype="f32",
            name="BTCV-shaped 96x160x160 patches, 14 classes, 3-D network with 12-direction selective scans "
                 "(UMambaEnc_SS3D design source; BASELINE.json configs[3])"),
    5: dict(img=(512, 640), in_ch=3, classes=8, batch=4, variant="A", precision="fp16", dtype="f16",
            name="Endovis17-shaped 512x640x3, 8 classes, shipped flash scaling (variant A), fp16 operands / fp32 sums + "
                 "GradScaler (BASELINE.json configs[4])"),
}


def pmc_traffic(kernel, batch):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950-corrected; collected at batch 10), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)["kernels"].get(kernel)
    except (OSError, ValueError, KeyError):
        return None
    if rec is None or batch != BATCH_PER_GPU:
        return None
    return rec["traffic_bytes"]


def host_threads():
    """CPU share of this process: the affinity mask, capped at 16 (the GPU box's per-GPU CPU share; its
    os.cpu_count() reports the whole host and oversubscribing it makes torch crawl)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def _oracle_step_seconds(img, batch, timed):
    """Mean seconds per train step of the CPU oracle: 1 warm-up step, then `timed` timed steps."""
    from oracle import mlagg_oracle as O
    torch.manual_seed(0)
    net = O.build_reference_config_model(img, 1, N_CLASSES, True, "B").train()
    opt = O.make_optimizer(net)
    data, target = O.synthetic_batch(batch, 1, *img, N_CLASSES, seed=1234)
    O.train_step(net, opt, data, target)
    t0 = time.perf_counter()
    for _ in range(timed):
        O.train_step(net, opt, data, target)
    return (time.perf_counter() - t0) / timed


def cpu_baseline(budget_s=40.0):
    """The CPU oracle's train step (plain PyTorch eager, unbind-loop scan) on this
    box's host cores, on a bounded sample: 1 warm-up + 3 timed steps at batch 1 of the 128x128 case (BASELINE
    configs[0]) first; if that predicts 1 + 3 steps of the 256x256 workload fit the budget, those are what is
    reported, else the small case scaled by the pixel ratio."""
    cores = host_threads()
    torch.set_num_threads(cores)
    small = (IMG[0] // 2, IMG[1] // 2)
    dt_small = _oracle_step_seconds(small, 1, 3)
    if 4 * 4.5 * dt_small <= budget_s:
        dt = _oracle_step_seconds(IMG, 1, 3)
        value, sample = 1.0 / dt, (f"1 warm-up + 3 timed train steps (fwd+bwd+clip+AdamW) at batch 1 of the "
                                   f"{IMG[0]}x{IMG[1]} workload: {dt:.2f} s per step; torch eager fp32, {cores} threads")
    else:
        value, sample = 1.0 / (4.0 * dt_small), (
            f"1 warm-up + 3 timed train steps at batch 1 of the {small[0]}x{small[1]} case: {dt_small:.2f} s per step, "
            f"scaled by the 4x pixel (= token) ratio to {IMG[0]}x{IMG[1]} images; torch eager fp32, {cores} threads")
    return {"value": round(value, 5), "unit": "images/sec", "cores": cores, "kind": "port", "sample": sample}


def cpu_baseline_3d(size, strides, n_cls, budget_s=40.0):
    """The CPU oracle of the 3-D network (oracle/umamba3d_oracle.py, torch eager, Python-loop scan) on a bounded sample: one
    warm-up + one timed train step (forward, loss, backward, SGD) at batch 1 of a 1/240 volume (8 x 32 x 32: every stage keeps
    more than one voxel), scaled to the full patch by the voxel ratio (the work is linear in the voxel count)."""
    from oracle import mlagg_oracle as O
    from oracle import umamba3d_oracle as U
    cores = host_threads()
    torch.set_num_threads(cores)
    small = (8, 32, 32)
    sm_strides = [[1, 1, 1], [2, 2, 2], [2, 2, 2], [2, 2, 2], [1, 2, 2], [1, 1, 1]]
    torch.manual_seed(0)
    net = U.build_reference_3d_model(1, n_cls, U.features_for(len(strides)), sm_strides).train()
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    data, target = U.synthetic_batch_3d(1, 1, small, sm_strides, n_cls)

    def step():
        opt.zero_grad()
        loss = O.deep_supervision_loss(net(data), target, batch_dice=False)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        opt.step()

    step()
    t0 = time.perf_counter()
    step()
    dt = time.perf_counter() - t0
    ratio = (size[0] * size[1] * size[2]) / float(small[0] * small[1] * small[2])
    return {"value": round(1.0 / (dt * ratio), 6), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"1 warm-up + 1 timed train step at batch 1 of an {small[0]}x{small[1]}x{small[2]} volume: {dt:.2f} s, scaled by "
                      f"the voxel ratio {ratio:.0f} to {size[0]}x{size[1]}x{size[2]} patches; torch eager fp32, {cores} threads"}


def main_3d(args, cfg, world, rank, dev, dev_index, ddp):
    """BASELINE configs[3]: train step of the 3-D network (model3d.UMambaEnc) with the base trainer's recipe
    (nnUNetTrainer.py:330-352 loss with batch_dice False as 3d_fullres plans set it, :448-452 SGD momentum 0.99 Nesterov,
    :855-857 clip 12), patches resident in HBM.  Never the headline."""
    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import _lib, miopen_tuning, model3d, profiling, trainer
    _lib.lib()
    miopen_tuning.use_tuned_convolutions(enabled=os.environ.get("MLAGG_BENCH_MIOPEN", "auto") == "find")
    size, strides = tuple(cfg["img"]), model3d.BTCV_STRIDES
    n = len(strides)
    torch.manual_seed(0)
    net = model3d.build_network_architecture_3d(cfg["in_ch"], cfg["classes"], [[3, 3, 3]] * n, strides, [2] * n, [2] * (n - 1))
    net = net.to(dev).train()
    opt = torch.optim.SGD(net.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    step_net = trainer.wrap_ddp(net, dev_index) if ddp else net
    data, target = model3d.synthetic_batch_3d(args.batch, cfg["in_ch"], size, strides, cfg["classes"], seed=1234 + rank, device=dev)

    def one_step():
        return trainer.train_step(step_net, opt, data, target, batch_dice=False, ddp=ddp)

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    for i in range(args.warmup):
        one_step()
        torch.cuda.synchronize()
        note(f"warm-up step {i + 1}/{args.warmup} done ({torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB peak)")
    tokens = size[0] * size[1] * size[2]
    dominant, table, fixed = None, {}, []
    if not args.no_roofline:
        profiling.select_all()
        one_step()
        table = profiling.collect()
        fixed = [k for k in table if profiling.algorithmic_bytes_sel1(k, args.batch, tokens) > 0]
        dominant = max(fixed, key=lambda k: table[k]["ms"]) if fixed else None
        profiling.select(dominant)
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    dt = time.perf_counter() - t0
    if ddp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    roof = None
    if dominant is not None:
        res = profiling.collect()[dominant]
        avg_ms = res["ms"] / max(res["count"], 1)
        alg = profiling.algorithmic_bytes_sel1(dominant, args.batch, tokens)
        achieved = alg / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_ms": round(avg_ms, 4),
                "launches_timed": res["count"], "algorithmic_bytes_per_launch": alg,
                "shape": f"stage 0: batch {args.batch} x {tokens} tokens x 64 channels x 12 directions, rank 2, one state",
                "fixed_shape_kernels": {
                    k: {"ms_per_launch": round(table[k]["ms"] / table[k]["count"], 4),
                        "achieved_GBs": round(profiling.algorithmic_bytes_sel1(k, args.batch, tokens)
                                              / (table[k]["ms"] / table[k]["count"] * 1e-3) / 1e9, 1)} for k in sorted(fixed)},
                "all_kernels_ms_per_step": {k: round(v["ms"], 3) for k, v in sorted(table.items())}}
        profiling.select(None)
    if rank == 0:
        n_img = args.batch * world * args.steps
        line = {"metric": f"images/sec (train step) 3-D network {size[0]}x{size[1]}x{size[2]} bs{args.batch}",
                "value": round(n_img / dt, 4), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": cfg["dtype"], "data": "synthetic",
                "config": {"workload": "nnUNetTrainerUMambaEnc_SS3D train step (SGD momentum 0.99 Nesterov, clip 12), " + cfg["name"],
                           "batch_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                           "final_loss": round(float(loss), 5), "launch": "eager", "miopen": "immediate mode",
                           "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
                "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_3d(size, strides, cfg["classes"])
        print(json.dumps(line), flush=True)
    if ddp:
        dist.destroy_process_group()


def matrix_arithmetic(precision):
    """How the dense products of this package's own GEMM / convolution kernels are formed (operands, accumulation and results are
    fp32 in the fp32 mode whichever way the products are issued)."""
    from mlagg_unet_amd import ops
    if precision != "fp32":
        return f"{precision} operands (rounded once), fp32 accumulation"
    on = [n for n, f in (("K5/K5w", ops.K5_X3), ("K18", ops.K18), ("K19", ops.K19)) if f]
    if not on:
        return "fp32 MFMA (v_mfma_f32_32x32x2_f32)"
    return ("fp32 operands as 3 exact bf16 pieces, 6 partial products on v_mfma_f32_32x32x16_bf16, fp32 accumulation (DESIGN 4i; error of an "
            "fp32 GEMM against float64) in " + ", ".join(on) + "; fp32 MFMA / libraries elsewhere")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config (2 = headline)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling); default: the config's")
    ap.add_argument("--precision", default=None, choices=["fp32", "bf16", "fp16"],
                    help="arithmetic type of the dense products; default: the config's (a diagnostic override)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step as one hipGraph (trainer.GraphedTrainStep).  Off by default: at "
                         "config 2 the step is GPU-bound either way (61.58 vs 61.63 ms measured A/B on one box) and "
                         "eager launches let the roofline kernel be event-timed inside the timed region itself")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    cfg = dict(CONFIGS[args.config])
    if args.precision is not None and args.precision != cfg["precision"]:
        cfg["precision"] = args.precision
        cfg["dtype"] = {"fp32": "f32", "bf16": "bf16", "fp16": "f16"}[args.precision]
        cfg["name"] += f" [precision overridden: {args.precision}]"
    global IMG, N_CLASSES
    IMG, N_CLASSES = cfg["img"], cfg["classes"]
    if args.batch is None:
        args.batch = cfg["batch"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # one process per GPU; MLAGG_DIST_BACKEND=gloo + fewer devices than ranks is the single-GPU rehearsal of
    # the multi-process path (ranks share the card, collectives go through the host) -- never a result
    backend = os.environ.get("MLAGG_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # MLAGG_FORCE_DDP=1: the data-parallel code path (RCCL process group, DDP buckets, batch-dice all-reduce) with ONE rank -- the
    # only way to execute the `nccl` branch on a single-GPU box (RCCL refuses two ranks per device); a rehearsal, never a result
    ddp = world > 1 or os.environ.get("MLAGG_FORCE_DDP", "0") == "1"
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    if args.config == 4:
        return main_3d(args, cfg, world, rank, dev, dev_index, ddp)

    import mlagg_unet_amd  # noqa: F401
    from mlagg_unet_amd import _lib, gemm_tuning, miopen_tuning, model, profiling, trainer

    _lib.lib()                                              # fail loudly if the HIP library is missing
    # The reference sets cudnn.benchmark=True (run_training.py:123-125); here: the committed result of that search
    # the committed find-db covers the fp32 convolutions of config 2; the other configurations run MIOpen's immediate-mode
    # choice.  MLAGG_BENCH_MIOPEN=find lets MIOpen search at first use instead -- measured on config 3: more than 7 minutes of
    # solver compilation before the first step finishes, so it is not the default; =off forces immediate mode everywhere
    mode = os.environ.get("MLAGG_BENCH_MIOPEN", "auto")
    miopen_db = miopen_tuning.use_tuned_convolutions(
        enabled=(mode == "find") or (mode == "auto" and args.config == 2 and cfg["precision"] == "fp32"))
    # ... and of the same kind of search over the library GEMM algorithms (PyTorch TunableOp table, tuning off at run time)
    gemm_db = gemm_tuning.use_tuned_gemms(enabled=mode == "auto" and args.precision is None and args.config == 2)
    torch.manual_seed(0)
    net = model.build_network_architecture(IMG, cfg["in_ch"], N_CLASSES, True, cfg["variant"], cfg["precision"]).to(dev).train()
    use_graph = args.graph and not ddp and not args.no_graph and cfg["precision"] != "fp16"      # DDP: eager (trainer.GraphedTrainStep)
    opt, sched = trainer.configure_optimizers(net, capturable=use_graph)
    sched.step(0)
    step_net = trainer.wrap_ddp(net, dev_index) if ddp else net
    data, target = trainer.synthetic_batch(args.batch, cfg["in_ch"], *IMG, N_CLASSES, seed=1234 + rank, device=dev)
    scaler = torch.amp.GradScaler("cuda") if cfg["precision"] == "fp16" else None      # reference B:152

    def eager_step():
        return trainer.train_step(step_net, opt, data, target, batch_dice=True, ddp=ddp, grad_scaler=scaler)

    one_step = eager_step

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    for i in range(args.warmup):
        one_step()
        torch.cuda.synchronize()
        note(f"warm-up step {i + 1}/{args.warmup} done")

    # pick the dominant hand-written kernel from one instrumented (untimed, eager) step; during the timed
    # region ONLY that kernel is bracketed by HIP events on its launch stream (event records are captured
    # into the graph with the kernel, so they time every replay)
    roof = None
    dominant = None
    mfma = None
    if not args.no_roofline:
        from mlagg_unet_amd import ops
        profiling.select_all()
        ops.FLOP_COUNT = {}
        eager_step()
        table = profiling.collect()
        flop, ops.FLOP_COUNT = ops.FLOP_COUNT, None
        if cfg["precision"] == "fp32":
            # the matrix kernels against the MFMA roofline: real flop of the products they ran in this step (counted at the call
            # sites, 2 M N K each) / their HIP-event time in the same step
            split_peak = BF16_MFMA_PEAK_TFS / 6.0
            mfma = {"unit": "TFLOP/s", "peak": round(split_peak, 1), "peak_note": "dense bf16 MFMA peak / 6 (every fp32 product is six "
                    "bf16 MFMAs)", "fp32_mfma_peak": FP32_MFMA_PEAK_TFS, "kernels": {}}
            for fam, prefix in MFMA_FAMILIES.items():
                ms = sum(v["ms"] for k, v in table.items() if k.startswith(prefix))
                if ms > 0 and flop.get(fam):
                    tf = flop[fam] / (ms * 1e-3) / 1e12
                    mfma["kernels"][fam] = {"flop_per_step": flop[fam], "ms_per_step": round(ms, 3), "achieved": round(tf, 1),
                                            "frac": round(tf / split_peak, 3), "vs_fp32_mfma_peak": round(tf / FP32_MFMA_PEAK_TFS, 3)}
        # candidates: kernels with ONE shape per step, hence one algorithmic bytes-per-launch figure
        fixed = [k for k in table if profiling.algorithmic_bytes(k, args.batch, IMG) > 0]
        dominant = max(fixed, key=lambda k: table[k]["ms"]) if fixed else None
        profiling.select(None if use_graph else dominant)
    if use_graph:
        note("capturing the step into a hipGraph")
        graphed = trainer.GraphedTrainStep(net, opt, data, target, batch_dice=True)
        one_step = graphed
        one_step()
        torch.cuda.synchronize()
        note("graph replay ok")

    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    dt = time.perf_counter() - t0
    if ddp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the path nnUNetv2_train drives reads the loss back every step (`l.detach().cpu().numpy()`, reference B:863; the plugin's
    # train_step mirrors it): the same steps once more with that host synchronisation, reported beside the headline
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        float(one_step().cpu())
    dt_synced = time.perf_counter() - t1

    if not args.no_roofline and dominant is not None:
        if use_graph:
            # events cannot be read back per replay from inside a graph: time the dominant kernel live over
            # the same number of eager steps right after the timed region (same process, same inputs)
            profiling.select(dominant)
            for _ in range(args.steps):
                eager_step()
        res = profiling.collect()[dominant]
        avg_ms = res["ms"] / max(res["count"], 1)
        alg = profiling.algorithmic_bytes(dominant, args.batch, IMG)
        achieved = alg / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dominant, args.batch) if args.config == 2 else None,
                "traffic_source": "static: profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                  "tools/bench_ops.py on the same shapes, gfx950 x2 fetch correction); not collected in this run",
                "avg_launch_ms": round(avg_ms, 4), "launches_timed": res["count"],
                "algorithmic_bytes_per_launch": alg,
                "fixed_shape_kernels": {
                    k: {"ms_per_launch": round(table[k]["ms"] / table[k]["count"], 4),
                        "achieved_GBs": round(profiling.algorithmic_bytes(k, args.batch, IMG)
                                              / (table[k]["ms"] / table[k]["count"] * 1e-3) / 1e9, 1)}
                    for k in sorted(fixed)},
                "all_kernels_ms_per_step": {k: round(v["ms"], 3) for k, v in sorted(table.items())},
                "mfma": mfma}
        profiling.select(None)

    if rank == 0:
        n_img = args.batch * world * args.steps
        line = {
            "metric": f"images/sec (train step) MLAgg-UNet-2D {IMG[0]}x{IMG[1]} bs{args.batch}",
            "value": round(n_img / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": "nnUNetTrainer_MLAgg_2D_dt_MS train step, " + cfg["name"],
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}" + (" (single-rank RCCL rehearsal of the DDP path)" if ddp and world == 1 else ""),
                       "final_loss": round(float(loss), 5),
                       "launch": "hipGraph replay of the whole step" if use_graph else "eager",
                       "ms_per_step_with_loss_readback": round(1e3 * dt_synced / args.steps, 3),
                       "images_per_sec_with_loss_readback": round(args.batch * world * args.steps / dt_synced, 3),
                       "distributed": ({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank0_device": str(dev),
                                        "grad_sync": trainer.GRAD_SYNC} if ddp else None),
                       "miopen": "tuned find-db (mlagg-unet_amd/miopen_db)" if miopen_db else "immediate mode",
                       "library_gemm": "TunableOp table (mlagg-unet_amd/gemm_db), tuning off" if gemm_db else "library default",
                       "matrix_arithmetic": matrix_arithmetic(cfg["precision"])},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline and args.config == 2:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
