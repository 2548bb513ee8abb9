#!/usr/bin/env python3
"""Benchmark of the Frustum-PointNet v1 hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one v1 inference pass (instance segmentation net -> masking -> T-Net ->
box net) over a batch of synthetic frustums already resident in HBM, followed - on
N > 1 - by the RCCL all-gather of the per-frustum result records.  Workload:
BASELINE config 1, 1024 frustums x 1024 points x 4 channels per GPU, fp32 (weak
scaling: every rank runs the same per-GPU batch).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "3d-object-detection_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2516.6  # dense bf16: 16x the fp32 rate (v_mfma_f32_32x32x16_bf16, 32 cycles / 16384 MAC)
ALGO_GFLOP_PER_FRUSTUM = None   # filled from the layer table below


def algorithmic_macs_per_frustum(n, m=512, pc_channels=4):
    """MACs of the v1 graph with the frustum-constant 1027 inputs of conv6 applied
    once per frustum instead of once per point (488.5 M at n=1024)."""
    seg_pt = pc_channels * 64 + 64 * 64 + 64 * 64 + 64 * 128 + 128 * 1024 + 64 * 512 + 512 * 256 + 256 * 128 + 128 * 128 + 128 * 2
    seg_fr = 1027 * 512
    tnet = (3 * 128 + 128 * 128 + 128 * 256) * m + 259 * 256 + 256 * 128 + 128 * 3
    box = (3 * 128 + 128 * 128 + 128 * 256 + 256 * 512) * m + 515 * 512 + 512 * 256 + 256 * 59
    return seg_pt * n + seg_fr + tnet + box


class KernelTimer:
    """HIP-event timer around shared-MLP launches (events are recorded on the stream the
    kernels are launched on: torch's current stream).  `only`: names to time (the rest
    pass untouched: a pair of events around every launch costs the step about 2 %)."""

    def __init__(self, only=None):
        self.pending = []
        self.only = only

    def begin(self, name, rows, cin, cout, macs=None):
        if self.only is not None and name not in self.only:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (name, 2.0 * (macs if macs is not None else rows * cin * cout), e0, e1)

    def end(self, tok):
        if tok is None:
            return
        tok[3].record()
        self.pending.append(tok)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, flops, e0, e1 in self.pending:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += e0.elapsed_time(e1) * 1e-3
        return agg


def cpu_baseline(seconds=12.0):
    """The numpy restatement of the v1 graph (oracle/v1_numpy.py) timed on this
    host's cores on BASELINE config 0 batches (32 frustums x 1024 points), repeated
    for about `seconds` of wall time."""
    from fpn import plan, synthetic, variables
    from oracle import v1_numpy as V
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = 1
    st = variables.VariableStore("cpu", seed=1)
    plan.init_variables(st, 4, True)
    st.randomize_batch_norm(2)
    params = st.numpy()
    pc, one_hot = synthetic.make_frustums(32, 1024, 4, seed=0)
    V.get_model(pc, one_hot, params)  # warm-up
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds:
        V.get_model(pc, one_hot, params)
        done += 32
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frustums/s", "cores": int(threads), "kind": "port",
            "sample": "numpy fp32 restatement of v1 get_model, %d frustums x 1024 pts in batches of 32 (%.1f s)"
                      % (done, dt)}


PEAK_HBM_GBS = 8000.0   # MI355X HBM3E spec peak, GB/s

# observer name of a launch -> kernel name in the rocprofv3 counter summaries
PMC_KERNEL = {"v1_seg_tail": "seg_tail_reg_kernel", "v1_seg_head": "seg_head_reg_kernel",
              "v1_obj_chain_box": "obj_chain_reg_kernel<true>", "v1_obj_chain_tnet": "obj_chain_reg_kernel<false>"}


def measured_traffic(name, batch, points):
    """HBM bytes per launch of kernel `name` from the newest committed rocprofv3 PMC
    summary (profiles/*_pmc_traffic.json, made by tools/pmc_summary.py from separate
    FETCH_SIZE and WRITE_SIZE passes over this same command at the default workload).
    Counters cannot be read from inside the timed process, so this is a recorded
    measurement; None when there is no summary or the workload is not the default."""
    import glob
    if (batch, points) != (1024, 1024) or name not in PMC_KERNEL:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        k = json.load(f)["kernels"].get(PMC_KERNEL[name])
    return (k["hbm_bytes_per_launch"], os.path.basename(files[-1])) if k else (None, None)


def pointset_ops(dev, frustums=256, reps=20):
    """Achieved HBM GB/s of the point-set operators at the v2 shapes (2048 points per
    frustum, 128 FPS samples), from the ALGORITHMIC bytes of each operator (inputs read
    once + outputs written once) and HIP-event time on the launch stream."""
    from fpn import synthetic
    from fpn.tf_ops import tf_grouping, tf_interpolate, tf_sampling
    B, N, M, NS, C = frustums, 2048, 128, 64, 64
    xyz = torch.from_numpy(synthetic.unit_cube(B, N)).to(dev)
    feat = torch.randn((B, N, C), device=dev)
    idx = tf_sampling.farthest_point_sample(M, xyz)
    new_xyz = tf_sampling.gather_point(xyz, idx)
    gi, _ = tf_grouping.query_ball_point(0.4, NS, xyz, new_xyz)
    feat2 = torch.randn((B, M, C), device=dev)
    cases = {
        "farthest_point_sample(2048->128)": (lambda: tf_sampling.farthest_point_sample(M, xyz), 12 * N + 4 * M),
        "gather_point(128)": (lambda: tf_sampling.gather_point(xyz, idx), 4 * M + 24 * M),
        "query_ball_point(2048,128,r0.4,ns64)": (lambda: tf_grouping.query_ball_point(0.4, NS, xyz, new_xyz),
                                                 12 * (N + M) + 4 * M * NS + 4 * M),
        "group_point(c=64)": (lambda: tf_grouping.group_point(feat, gi), 4 * M * NS + 2 * 4 * M * NS * C),
        "three_nn(2048,128)": (lambda: tf_interpolate.three_nn(xyz, new_xyz), 12 * (N + M) + 24 * N),
        "three_nn_interpolate(2048,128,c=64)": (lambda: tf_interpolate.three_nn_interpolate(xyz, new_xyz, feat2),
                                                12 * (N + M) + 4 * M * C + 4 * N * C),
    }
    out = {}
    # FPN_POINTSET_SYNC=1 (counter runs): one launch on the device at a time, so that a
    # dispatch's hardware counters hold its own traffic only
    isolate = os.environ.get("FPN_POINTSET_SYNC") == "1"
    for name, (fn, bytes_per_frustum) in cases.items():
        if isolate:
            inner = fn
            fn = lambda inner=inner: (inner(), torch.cuda.synchronize())      # noqa: E731
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbs = B * bytes_per_frustum / ms / 1e6
        out[name] = {"ms": ms, "frustums_per_s": B / ms * 1e3, "algorithmic_bytes_per_frustum": bytes_per_frustum,
                     "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS}
    return out


def settle_steps(warmup):
    """Untimed steps beyond the requested warm-up: the GPU's clocks take a few steps of
    work to reach their sustained state (measured: 126 k frustums/s over the first three
    steps after one warm-up step, 138 k from the fifth step on), so at least five untimed
    steps always precede the timed region."""
    return max(0, 5 - warmup)


REHEARSAL = os.environ.get("FPN_BENCH_BACKEND", "nccl") != "nccl"


def init_rank(local_rank, world):
    """One process per GPU: bind the rank's device and join the RCCL job (backend "nccl").
    FPN_BENCH_BACKEND=gloo is a rehearsal switch for a box with fewer GPUs than ranks: the
    ranks then share the visible devices round-robin and gather through gloo - it checks
    the multi-rank control flow, its timings mean nothing."""
    rehearsal = REHEARSAL
    index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(index)
    dev = torch.device("cuda", index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(os.environ["FPN_BENCH_BACKEND"])
        else:
            dist.init_process_group("nccl", device_id=dev)
    return dev


def side_config(args):
    """The other GPU workloads BASELINE names, with the same timing protocol as the
    headline one: --config v2 (PointNet++ variant: FPS + ball query + grouping + shared
    MLPs + 3-NN interpolation, 2048 points) and --config bayes (Flipout v1, S sampled
    passes per frustum + ensemble statistics).  One JSON line, no roofline object: these
    are mixes of HBM-bound point-set kernels and MFMA layers, the per-operator
    bandwidths are in the `pointset_ops` table of the default run."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = init_rank(local_rank, world)
    from fpn import dist as fdist
    from fpn import synthetic, variables
    if args.config == "v2":
        from fpn.models import frustum_pointnets_v2 as M2
        b_local = args.batch or 256
        points = args.points or 2048
        pc_np, oh_np = synthetic.make_frustums(b_local, points, 4, seed=rank)
        pc, one_hot = torch.from_numpy(pc_np).to(dev), torch.from_numpy(oh_np).to(dev)
        st = variables.VariableStore(dev, seed=1)
        if args.dtype == "bf16":
            from fpn import pointnet_util
            pointnet_util.SA_PRECISION = "bf16"       # the fused set-abstraction kernels
        with variables.use_store(st):
            M2.get_model(pc[:2], one_hot[:2], False)      # creates the variables
        st.randomize_batch_norm(2)

        def step():
            with variables.use_store(st):
                return fdist.pack_results(M2.get_model(pc, one_hot, False))
        label = "Frustum PointNet v2 (SA-MSG x2, SA, FP x3; box net SA x3) inference, %d frustums x %d pts x 4 ch " \
                "per GPU, %s" % (b_local, points, "bf16 MFMA in the set-abstraction kernels, fp32 elsewhere"
                                 if args.dtype == "bf16" else "fp32")
        metric = "frustums/sec (node) FPointNet v2 N=%d" % points
    else:
        from fpn import bayes, uncertainty
        b_local = args.batch or 128
        points = args.points or 1024
        pc_np, oh_np = synthetic.make_frustums(b_local, points, 3, seed=rank)
        pc, one_hot = torch.from_numpy(pc_np).to(dev), torch.from_numpy(oh_np).to(dev)
        st = variables.VariableStore(dev, seed=1)
        bayes.init_variables(st, 3)
        bp = bayes.BayesPlan(st, 3, mc_dropout=args.mc_dropout, precision="bf16" if args.dtype == "bf16" else "fp32")
        gen = torch.Generator(device=dev)
        gen.manual_seed(7 + rank)

        def step():
            ep = bp.forward(pc, one_hot, args.samples, generator=gen)
            return uncertainty.mc_statistics(ep["box_output"], ep["stage1_center"])
        label = "Flipout Frustum PointNet v1, %d Monte-Carlo samples x %d frustums x %d pts per GPU, %s, " \
                "ensemble statistics on device%s" % (args.samples, b_local, points,
                                                     "bf16 MFMA in the convolution stacks (fp32 first layers, "
                                                     "maxima, head and fully connected layers)"
                                                     if args.dtype == "bf16" else "fp32",
                                                     ", Monte-Carlo dropout before the head" if args.mc_dropout else "")
        metric = "frustums/sec (node) FPointNet Bayes S=%d N=%d" % (args.samples, points)
    gathered = None
    if args.graph:
        from fpn.graph import GraphedForward
        if args.config == "v2":
            def fwd(pc_, oh_):
                with variables.use_store(st):
                    return fdist.pack_results(M2.get_model(pc_, oh_, False))
            graphed = GraphedForward(fwd, [pc, one_hot])
            step = lambda: graphed(pc, one_hot)       # noqa: E731
        label += ", HIP-graph replay"
    per_step = b_local
    if args.stream_total:
        # a scene set larger than a batch, resident in (pinned) HOST memory, streamed through
        # the model in chunks of b_local frustums with the copies on a second stream
        if args.config != "v2":
            raise SystemExit("--stream-total goes with --config v2")
        reps = (args.stream_total + b_local - 1) // b_local
        host_pc = pc.cpu().repeat(reps, 1, 1)[:args.stream_total].contiguous().pin_memory()
        host_oh = one_hot.cpu().repeat(reps, 1)[:args.stream_total].contiguous().pin_memory()
        per_step = args.stream_total

        def fwd_ep(pc_, oh_):
            with variables.use_store(st):
                return M2.get_model(pc_, oh_, False)
        step = lambda: fdist.run_streamed(fwd_ep, host_pc, host_oh, b_local, dev)       # noqa: E731
        label += ", %d frustums per GPU streamed from pinned host memory in chunks of %d (host-to-device copies " \
                 "inside the timed region, overlapped with compute)" % (args.stream_total, b_local)

    def full_step():
        nonlocal gathered
        rec = step()
        if world > 1:
            if gathered is None:
                gathered = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=dev)
            dist.all_gather_into_tensor(gathered, rec.contiguous())

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup + settle_steps(args.warmup)):
        full_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        print(json.dumps({"metric": metric, "value": world * per_step * args.steps / dt, "unit": "frustums/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "settle_steps": settle_steps(args.warmup),
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None,
                          "dtype": args.dtype,
                          "data": "synthetic",
                          "config": {"workload": label, "frustums_per_gpu": b_local, "points": points}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="frustums per GPU (default: 1024 for v1, 256 for v2, 128 for bayes)")
    ap.add_argument("--points", type=int, default=None, help="points per frustum (default: 1024; 2048 for v2)")
    ap.add_argument("--chunk", type=int, default=0, help="frustums per chunk (0 = plan default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-pointset", action="store_true", help="skip the point-set operator bandwidth table")
    ap.add_argument("--config", choices=("v1", "v2", "bayes"), default="v1",
                    help="v1: BASELINE config 1 (the headline metric, default); v2: PointNet++ variant, 2048 points "
                         "(config 2); bayes: Flipout v1 with --samples Monte-Carlo passes per frustum (config 3)")
    ap.add_argument("--stream-total", type=int, default=0,
                    help="--config v2: frustums per GPU held in host memory and streamed through the model in "
                         "chunks of --batch (BASELINE config 5's access pattern); a step = the whole set")
    ap.add_argument("--samples", type=int, default=32, help="Monte-Carlo samples per frustum (--config bayes)")
    ap.add_argument("--mc-dropout", action=argparse.BooleanOptionalAction, default=True,
                    help="--config bayes: keep the Dropout(0.5) before the segmentation head active (Monte-Carlo "
                         "dropout, as BASELINE's Bayesian configuration samples; --no-mc-dropout: Flipout noise only)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="operand type of the layer-chain kernels' matrix products (accumulation is always fp32); "
                         "f32 is the headline configuration")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a HIP graph (side configs; removes the host launch overhead)")
    args = ap.parse_args()
    if args.config != "v1":
        return side_config(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = init_rank(local_rank, world)

    from fpn import dist as fdist
    from fpn import ops, plan, synthetic, variables

    # identical weights on every rank (seeded); per-rank shard of one global tensor
    st = variables.VariableStore(dev, seed=1)
    plan.init_variables(st, 4, True)
    st.randomize_batch_norm(2)
    bf16 = args.dtype == "bf16"
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    net = plan.V1Plan(st, 4, True, chunk_frustums=args.chunk or None, precision="bf16" if bf16 else "fp32")
    b_local = args.batch or 1024
    args.points = args.points or 1024
    pc_np, oh_np = synthetic.make_frustums(b_local, args.points, 4, seed=rank)
    pc = torch.from_numpy(pc_np).to(dev)
    one_hot = torch.from_numpy(oh_np).to(dev)
    gathered = torch.empty((world * b_local, fdist.RECORD), dtype=torch.float32, device=dev) if world > 1 else None

    def step():
        ep = net.forward(pc, one_hot)
        rec = fdist.pack_results(ep)
        if world > 1:
            if REHEARSAL:       # gloo: collectives on host copies
                fdist.gather_results(rec.cpu())
            else:
                fdist.gather_results(rec, out=gathered)
        return rec

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup + settle_steps(args.warmup)):
        step()
    # Inside the timed region only the dominant kernel (the segmentation tail, 43 % of the
    # step) is bracketed by HIP events; the per-kernel table comes from three more steps
    # after it, so that event pairs around every launch do not slow the measured step.
    dominant = "v1_seg_tail_bf16" if bf16 else "v1_seg_tail"
    timer = None if args.no_kernel_timer else KernelTimer(only=None if args.chunk else {dominant})
    ops.launch_observer = timer
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ops.launch_observer = None
    table = None
    if timer is not None and rank == 0:
        table = KernelTimer()
        ops.launch_observer = table
        for _ in range(3):
            net.forward(pc, one_hot)
        ops.launch_observer = None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if REHEARSAL else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    macs = algorithmic_macs_per_frustum(args.points)
    total_frustums = world * b_local * args.steps
    value = total_frustums / dt
    out = {
        "metric": "frustums/sec (node) FPointNet v1 N=%d" % args.points,
        "value": value, "unit": "frustums/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "settle_steps": settle_steps(args.warmup),
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "Frustum PointNet v1 inference, %d frustums x %d pts x 4 ch per GPU, %s"
                               % (b_local, args.points, "bf16 MFMA with fp32 accumulation (not the headline "
                                  "configuration, which is fp32)" if bf16 else "fp32 MFMA (BASELINE config 1)"),
                   "frustums_per_gpu": b_local, "points": args.points,
                   "algorithmic_gflop_per_frustum": 2 * macs / 1e9,
                   "model_tflops": value * 2 * macs / 1e12,
                   "model_frac_of_mfma_peak": value * 2 * macs / 1e12 / (peak * world),
                   "parallelism": "frustum-sharded x%d, all_gather of %d-float records" % (world, fdist.RECORD)},
    }
    if rank == 0:
        if timer is not None:
            agg = timer.summary()
            full = table.summary()
            name = max(agg, key=lambda k: agg[k][2])
            launches, flops, secs = agg[name]
            ach = flops / secs / 1e12
            top = max(full, key=lambda k: full[k][2])
            if top != name:      # the timed kernel must be the one that dominates the step
                out["roofline_note"] = "largest share of the step: %s" % top
            traffic, traffic_src = measured_traffic(name, b_local, args.points) if not args.chunk else (None, None)
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": traffic,
                               "traffic_source": traffic_src, "kernel": name,
                               "launches": launches, "avg_launch_ms": secs / launches * 1e3,
                               "algorithmic_gflop_per_launch": flops / launches / 1e9,
                               "share_of_step_time": secs / dt}
            out["kernels"] = {k: {"launches": v[0], "gflop": v[1] / 1e9, "ms": v[2] * 1e3} for k, v in full.items()}
            out["kernels_source"] = "3 steps after the timed region, HIP events around every launch"
        if world == 1 and not args.no_pointset:
            # 256 frustums = one workgroup per CU for the per-frustum kernels (FPS, ball
            # query): latency-bound; 4096 frustums lets several workgroups share a CU
            out["pointset_ops"] = pointset_ops(dev)
            out["pointset_ops_4096_frustums"] = pointset_ops(dev, frustums=4096, reps=5)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
