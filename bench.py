#!/usr/bin/env python3
"""Benchmark of the Frustum-PointNet hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N > 1: starts N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
         --master-port P bench.py --gpus N --steps K --warmup W

One step = one inference pass over this rank's shard of ONE global seeded batch of
synthetic frustums, resident in HBM, followed - on N > 1 - by the single RCCL all-gather
of the per-frustum result records (weak scaling: every rank holds the same number of
frustums).  Default workload: BASELINE config 1, v1 network, 1024 frustums x 1024 points x
4 channels per GPU, fp32.  --config v2 / --config bayes run the other GPU configurations
with the same protocol.  Rank 0 prints ONE JSON line with `roofline` (dominant kernel,
HIP events on the launch stream) and `cpu_baseline` (the CPU oracle on this host's cores).
"""
import argparse
import json
import os
import socket
import subprocess
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
PEAK_HBM_GBS = 8000.0           # MI355X HBM3E spec peak, GB/s

REHEARSAL = os.environ.get("FPN_BENCH_BACKEND", "nccl") != "nccl"


def algorithmic_macs_per_frustum(n, m=512, pc_channels=4):
    """MACs of the v1 graph with the frustum-constant 1027 inputs of conv6 applied
    once per frustum instead of once per point (488.5 M at n=1024)."""
    seg_pt = pc_channels * 64 + 64 * 64 + 64 * 64 + 64 * 128 + 128 * 1024 + 64 * 512 + 512 * 256 + 256 * 128 + 128 * 128 + 128 * 2
    seg_fr = 1027 * 512
    tnet = (3 * 128 + 128 * 128 + 128 * 256) * m + 259 * 256 + 256 * 128 + 128 * 3
    box = (3 * 128 + 128 * 128 + 128 * 256 + 256 * 512) * m + 515 * 512 + 512 * 256 + 256 * 59
    return seg_pt * n + seg_fr + tnet + box


def reference_macs_per_frustum_v1(n, m=512, pc_channels=4):
    """MACs of the v1 graph as the reference multiplies it: conv6 over the tiled 1091-wide
    tensor of every point (frustum_pointnets_v1.py:64-70); 1026.4 M at n=1024, 4 channels."""
    return algorithmic_macs_per_frustum(n, m, pc_channels) - 1027 * 512 + 1027 * 512 * n


def v2_macs_per_frustum(cfg, n, m=512, pc_channels=4):
    """MACs of one frustum through the v2 graph -> (as the reference's modules multiply it,
    as executed here).  Reference form: every gathered row goes through the whole first
    layer (pointnet_util.py:116-121, :177-190) and every dense point through the whole first
    feature-propagation layer on the concatenated tensor (:218-226).  Executed form: the
    features' rows of a first layer meet each POINT once where the features are >= 32 wide
    (the gathered rows add only the 3 coordinate channels), the interpolated features'
    rows of a feature-propagation layer meet each KNOWN point once, and a single known point
    (group-all level) enters as one row per frustum.  Zero padding of narrow layers to the
    kernels' tile widths is not counted in either."""
    def stack(cin, widths):
        total = 0
        for w in widths:
            total += cin * w
            cin = w
        return total
    cf0 = pc_channels - 3
    seg, box = cfg["seg"], cfg["box"]
    ref = exe = 0
    # --- segmentation net
    np1 = seg["sa1"]["npoint"]
    c1 = 0
    for ns, mlp in zip(seg["sa1"]["nsample"], seg["sa1"]["mlp"]):
        rows = np1 * ns
        ref += rows * stack(cf0 + 3, mlp)
        exe += rows * stack(cf0 + 3, mlp) if cf0 < 32 else n * cf0 * mlp[0] + rows * (3 * mlp[0] + stack(mlp[0], mlp[1:]))
        c1 += mlp[-1]
    np2 = seg["sa2"]["npoint"]
    c2 = 0
    for ns, mlp in zip(seg["sa2"]["nsample"], seg["sa2"]["mlp"]):
        rows = np2 * ns
        ref += rows * stack(c1 + 3, mlp)
        exe += np1 * c1 * mlp[0] + rows * (3 * mlp[0] + stack(mlp[0], mlp[1:]))
        c2 += mlp[-1]
    g = np2 * stack(c2 + 3, seg["sa3"]["mlp"])
    ref += g; exe += g
    c3 = seg["sa3"]["mlp"][-1] + 3                       # + one-hot
    f1, f2, f3 = seg["fp1"], seg["fp2"], seg["fp3"]
    ref += np2 * stack(c3 + c2, f1)
    exe += c3 * f1[0] + np2 * (c2 * f1[0] + stack(f1[0], f1[1:]))
    ref += np1 * stack(f1[-1] + c1, f2)
    exe += np2 * f1[-1] * f2[0] + np1 * (c1 * f2[0] + stack(f2[0], f2[1:]))
    head = stack(f3[-1], [128, 2])
    ref += n * (stack(f2[-1] + pc_channels, f3) + head)
    exe += np1 * f2[-1] * f3[0] + n * (pc_channels * f3[0] + stack(f3[0], f3[1:]) + head)
    # --- T-Net (the v1 one)
    t = (3 * 128 + 128 * 128 + 128 * 256) * m + 259 * 256 + 256 * 128 + 128 * 3
    ref += t; exe += t
    # --- box net
    b1, b2 = box["sa1"], box["sa2"]
    r1 = b1["npoint"] * b1["nsample"]
    ref += r1 * stack(3, b1["mlp"]); exe += r1 * stack(3, b1["mlp"])
    r2 = b2["npoint"] * b2["nsample"]
    ref += r2 * stack(b1["mlp"][-1] + 3, b2["mlp"])
    exe += b1["npoint"] * b1["mlp"][-1] * b2["mlp"][0] + r2 * (3 * b2["mlp"][0] + stack(b2["mlp"][0], b2["mlp"][1:]))
    g = b2["npoint"] * stack(b2["mlp"][-1] + 3, box["sa3"]["mlp"])
    ref += g; exe += g
    fc = stack(box["sa3"]["mlp"][-1] + 3, [512, 256, 59])
    ref += fc; exe += fc
    return ref, exe


def bayes_macs_per_frustum(n, samples, m=512):
    """MACs of one frustum's S-sample Flipout ensemble (3-channel v1 topology without batch
    norm, FPointNet_Bayes.py:54-136) -> (textbook form, as executed).  Textbook: every Flipout
    layer is two products per sample (mean and perturbation) over the graph as the reference
    multiplies it; the two output layers are plain.  Executed: all points of a frustum share
    their signs, so a layer is ONE product with the frustum's effective weights, and the
    frustum-constant inputs of conv6 are applied once per (sample, frustum)."""
    ref1 = reference_macs_per_frustum_v1(n, m, 3)
    plain = 128 * 3 + 256 * 59
    return samples * (2 * (ref1 - plain) + plain), samples * algorithmic_macs_per_frustum(n, m, 3)


# ---- launching --------------------------------------------------------------------

def spawn_ranks(gpus):
    """`bench.py --gpus N` started bare (no WORLD_SIZE in the environment): start N ranks,
    one per GPU, as children of this process through torch.distributed.run and return its
    exit status.  Nothing here touches a GPU (device_count() does not initialise one), so
    the children are the only processes that do."""
    if not REHEARSAL:
        have = torch.cuda.device_count()
        if have < gpus:
            print("bench.py: --gpus %d but only %d GPU(s) are visible (FPN_BENCH_BACKEND=gloo rehearses the "
                  "multi-rank control flow on fewer devices)" % (gpus, have), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def init_rank(local_rank, world):
    """One process per GPU: bind the rank's device and join the RCCL job (backend "nccl").
    FPN_BENCH_BACKEND=gloo is a rehearsal switch for a box with fewer GPUs than ranks: the
    ranks then share the visible devices round-robin and gather through gloo - it checks
    the multi-rank control flow, its timings mean nothing."""
    index = local_rank % torch.cuda.device_count() if REHEARSAL else local_rank
    torch.cuda.set_device(index)
    dev = torch.device("cuda", index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if REHEARSAL:
            dist.init_process_group(os.environ["FPN_BENCH_BACKEND"])
        else:
            dist.init_process_group("nccl", device_id=dev)
    return dev


def rank_identity(rank, dev, world):
    """What this rank ran on, for the record: device name, PCI bus id, compute units, the
    collective backend and its library version - gathered from every rank so that an N-GPU
    line shows N distinct devices behind one RCCL job."""
    props = torch.cuda.get_device_properties(dev)
    info = {"rank": rank, "device_index": dev.index, "name": props.name,
            "gcn_arch": getattr(props, "gcnArchName", None), "compute_units": props.multi_processor_count,
            "hbm_gib": round(props.total_memory / 2 ** 30, 1),
            "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                                getattr(props, "pci_device_id", 0)),
            "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid(), "host": socket.gethostname(),
            "backend": dist.get_backend() if world > 1 else None}
    if world > 1 and not REHEARSAL:
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:       # the version query is evidence only; its absence must not end the run
            info["rccl_version"] = "unavailable: %s" % e
    return info


# ---- timing ------------------------------------------------------------------------

class KernelTimer:
    """HIP-event timer around the fused-kernel launches (events are recorded on the stream
    the kernels are launched on: torch's current stream).  `only`: names to time (the rest
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


def settle_steps(warmup):
    """Untimed steps beyond the requested warm-up: the GPU's clocks take a few steps of
    work to reach their sustained state (measured: 126 k frustums/s over the first three
    steps after one warm-up step, 138 k from the fifth step on), so at least five untimed
    steps always precede the timed region."""
    return max(0, 5 - warmup)


# ---- recorded HBM traffic ------------------------------------------------------------

def pmc_kernel_prefix(obs_name):
    """Observer name of a launch -> the prefix of its kernel's name in the rocprofv3 counter
    summaries (tools/pmc_summary.py strips the namespace and the argument list)."""
    fixed = {"v1_seg_tail": "seg_tail_reg_kernel", "v1_seg_head": "seg_head_reg_kernel",
             "v1_obj_chain_box": "obj_chain_reg_kernel<true>", "v1_obj_chain_tnet": "obj_chain_reg_kernel<false>",
             "v1_seg_tail_bf16": ("seg_tail_fused_bf16_kernel", "seg_tail_bf16_kernel"), "v1_seg_head_bf16": "seg_head_bf16_kernel",
             "v1_obj_chain_box_bf16": "obj_chain_bf16_kernel<true", "v1_obj_chain_tnet_bf16": "obj_chain_bf16_kernel<false",
             "fp_head_chain": "fp_head_reg_kernel", "fp_head_chain_pre": "fp_head_pre_kernel", "fp_head_chain_pre_bf16": "fp_head_pre_bf16_kernel", "fp_head_chain_bf16": ("fp_head_lds_bf16_kernel", "fp_head_bf16_kernel")}
    if obs_name in fixed:
        return fixed[obs_name]
    if obs_name.startswith("sa_chain"):
        # "sa_chain[_bf16][cf=C,w1-w2-w3]" -> template arguments <cf/32, padded widths / 32>
        # (a third field ",pre=C": wide features went through the first layer once per point, in another launch)
        inside = obs_name[obs_name.index("[") + 1:-1]
        cf_s, widths = inside.split(",")[:2]
        cf = int(cf_s[3:])
        w = [int(v) for v in widths.split("-")]
        up = lambda v, q: (v + q - 1) // q * q        # noqa: E731
        bf16 = "_bf16" in obs_name
        w1 = up(w[0], 64)
        # the middle layer runs at a multiple of 32 after a 64-wide first layer (96 stays 96)
        w2 = up(w[1], 32) if (cf < 32 and w1 == 64 and up(w[1], 32) == 96) else up(w[1], 64)
        targs = "%d, %d, %d, %d" % (cf // 32 if cf >= 32 else 0, w1 // 32, w2 // 32, up(w[2], 128) // 32)
        if not bf16 and cf < 32 and w == [32, 32, 64]:
            return "sa_small_kernel"
        if not bf16 and cf < 32:  # the persistent kernel of the stacks over coordinates, or its per-group twin
            return ("sa_persist_kernel<%s," % targs[3:], "sa_chain_kernel<%s>" % targs)
        if bf16 and cf < 32:      # the persistent kernel of the stacks over coordinates
            return "sa_first_bf16_kernel<%s," % targs[3:]
        return ("sa_chain_bf16_kernel<%s," if bf16 else "sa_chain_kernel<%s>") % targs
    return None


def measured_traffic(obs_name, workload):
    """HBM bytes per launch of the kernel behind `obs_name` from the newest committed
    rocprofv3 PMC summary OF THIS WORKLOAD that holds the kernel (profiles/*_pmc_traffic.json,
    made by tools/pmc_summary.py from separate FETCH_SIZE and WRITE_SIZE passes over this same
    command).  Counters cannot be read from inside the timed process, so this is a recorded
    measurement: (None, None) when no summary of the workload holds the kernel."""
    import glob
    prefix = pmc_kernel_prefix(obs_name)
    if prefix is None or workload is None:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        # summaries written before the workload tag existed: plain kernel names = the headline workload
        tag = d.get("workload") or ("v1_f32_1024x1024" if not any(" @ " in k for k in d["kernels"]) else None)
        if tag != workload:
            continue
        hits = [v for k, v in d["kernels"].items() if k.split(" @ ")[0].startswith(prefix)]
        n = sum(v["launches_sampled"] for v in hits)
        if n:
            return sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in hits) / n, os.path.basename(path)
    return None, None


def recorded_clock(obs_name, workload):
    """The clock the chip held under the kernel behind `obs_name`, from the newest committed
    counter record of this workload: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the
    kernel's average duration in the kernel trace of the same command (profiles/*_sq.json and
    *_kernel_stats.csv, tools/run_counters.sh).  Dense bf16 MFMA loops on random data run well
    below the 2.4 GHz the peak is quoted at; (None, None) when no record holds the kernel."""
    import csv
    import glob
    prefix = pmc_kernel_prefix(obs_name)
    if prefix is None or workload is None:
        return None, None
    prefixes = prefix if isinstance(prefix, tuple) else (prefix,)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get("workload") != workload:
                continue
            stats = path[:-len("sq.json")] + "kernel_stats.csv"
            with open(stats) as f:
                rows = list(csv.DictReader(f))
        except (OSError, ValueError):
            continue
        for k, v in d["kernels"].items():
            name = k.split(" @ ")[0]
            if not name.startswith(prefixes) or "GRBM_GUI_ACTIVE" not in v.get("counters", {}):
                continue
            for r in rows:       # the same instantiation of the kernel in the trace
                if name in r["Name"].replace("fpn::(anonymous namespace)::", "") and float(r["AverageNs"]) > 0:
                    return v["counters"]["GRBM_GUI_ACTIVE"] / 8.0 / float(r["AverageNs"]), os.path.basename(path)
    return None, None


# ---- CPU baselines (the oracle, timed on this host) ------------------------------------

def cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = 1
    return model, int(threads)


def _timed_loop(fn, units_per_call, seconds):
    fn()            # warm-up
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds:
        fn()
        done += units_per_call
    return done / (time.perf_counter() - t0), done


def cpu_baseline(kind, seconds=8.0, samples=32, params=None):
    """The numpy restatement of the workload's graph (oracle/) timed on this host: once on
    all the threads numpy's BLAS pool has (`value`, `cores`) and once on one thread
    (`one_thread`).  A bounded sample of the same workload: small batches repeated for
    about `seconds` per measurement."""
    import numpy as np
    from fpn import synthetic, variables
    model, threads = cpu_info()
    if kind == "v1":
        from fpn import plan
        from oracle import v1_numpy as V
        st = variables.VariableStore("cpu", seed=1)
        plan.init_variables(st, 4, True)
        st.randomize_batch_norm(2)
        params = st.numpy()
        pc, one_hot = synthetic.make_frustums_range(0, 32, 1024, 4, seed=0)
        fn, units = (lambda: V.get_model(pc, one_hot, params)), 32
        what = "numpy fp32 restatement of v1 get_model, batches of 32 frustums x 1024 pts"
    elif kind == "v2":
        from fpn.models import frustum_pointnets_v2 as M2
        from oracle import pointnet_numpy as P
        from oracle import v1_numpy as V
        pc, one_hot = synthetic.make_frustums_range(0, 4, 2048, 4, seed=0)
        # params: host copy of the GPU run's variables (the v2 layers create theirs on first use)

        def fn():
            logits = P.v2_instance_seg(pc, one_hot, params, M2.CONFIG)
            obj, mean, _, _ = V.point_cloud_masking(pc, logits)
            delta = V.center_regression_net(obj, one_hot, params)
            return P.v2_box_estimation(obj - delta[:, None, :], one_hot, params, M2.CONFIG)
        units = 4
        what = "numpy fp32 + C point-set oracle restatement of the v2 graph, batches of 4 frustums x 2048 pts"
    else:
        from fpn import bayes
        from oracle import bayes_numpy as BN
        st = variables.VariableStore("cpu", seed=1)
        bayes.init_variables(st, 3)
        params = st.numpy()
        b = 8
        pc, one_hot = synthetic.make_frustums_range(0, b, 1024, 3, seed=0)
        rng = np.random.RandomState(0)
        noise = {}
        for scope, cin, cout, _, _ in plan_table(3):
            if scope in bayes.PLAIN:
                continue
            sigma = np.log1p(np.exp(params[scope + "/rho"])) + bayes.FLIPOUT_EPS
            noise[scope] = ((sigma * rng.standard_normal((cin, cout))).astype(np.float32),
                            rng.choice([-1.0, 1.0], (b, cin)).astype(np.float32),
                            rng.choice([-1.0, 1.0], (b, cout)).astype(np.float32))
        keep = rng.random_sample((b, 1024, 128)) < 0.5
        fn = lambda: BN.forward_sample(pc, one_hot, params, noise, dtype=np.float32, dropout_keep=keep)      # noqa: E731
        units = b / float(samples)        # one sampled pass over 8 frustums = 8/S frustums' worth of an S-sample ensemble
        what = "numpy fp32 textbook-Flipout restatement, sampled passes over 8 frustums x 1024 pts (%d passes per " \
               "frustum make one unit)" % samples
    many, done_many = _timed_loop(fn, units, seconds)
    one = None
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            one, _ = _timed_loop(fn, units, seconds)
    except Exception:
        pass
    return {"value": many, "unit": "frustums/s", "cores": threads, "kind": "port",
            "one_thread": {"value": one, "cores": 1}, "cpu_model": model,
            "sample": "%s, repeated for about %.0f s per measurement (%.1f units at %d threads); labelled "
                      "reference-equivalent CPU path (restated), not TensorFlow" % (what, seconds, done_many, threads)}


def plan_table(pc_channels):
    from fpn import plan
    return plan.layer_table(pc_channels)


def pointset_cpu(seconds_per_op=1.0):
    """The point-set operators on ONE host core at the shapes of `pointset_ops`: the scalar C
    oracle (kind "port") and, where the reference's own stand-alone CPU loops were compiled
    into oracle/_ref (ball query, group_point, three_interpolate), those too (kind
    "reference").  4 frustums per call, repeated for about `seconds_per_op`."""
    import numpy as np
    from fpn import synthetic
    from oracle import pointset as O
    B, N, M, NS, C = 4, 2048, 128, 64, 64
    xyz = synthetic.unit_cube(B, N)
    rng = np.random.RandomState(0)
    feat = rng.standard_normal((B, N, C)).astype(np.float32)
    feat2 = rng.standard_normal((B, M, C)).astype(np.float32)
    idx = O.farthest_point_sample(M, xyz)
    new_xyz = O.gather_point(xyz, idx)
    gi, _ = O.query_ball_point(0.4, NS, xyz, new_xyz)
    d3, i3 = O.three_nn(xyz, new_xyz)
    w3 = O.fp_weights(d3)
    cases = {
        "farthest_point_sample(2048->128)": (lambda: O.farthest_point_sample(M, xyz), None),
        "gather_point(128)": (lambda: O.gather_point(xyz, idx), None),
        "query_ball_point(2048,128,r0.4,ns64)": (lambda: O.query_ball_point(0.4, NS, xyz, new_xyz),
                                                 lambda: O.ref_query_ball_point(0.4, NS, xyz, new_xyz)),
        "group_point(c=64)": (lambda: O.group_point(feat, gi), lambda: O.ref_group_point(feat, gi)),
        "three_nn(2048,128)": (lambda: O.three_nn(xyz, new_xyz), None),
        "three_nn_interpolate(2048,128,c=64)": (
            lambda: O.three_interpolate(feat2, *(lambda d, i: (i, O.fp_weights(d)))(*O.three_nn(xyz, new_xyz))),
            None),
        "three_interpolate(2048,128,c=64)": (lambda: O.three_interpolate(feat2, i3, w3),
                                             lambda: O.ref_three_interpolate(feat2, i3, w3)),
    }
    out = {}
    for name, (port, ref) in cases.items():
        rate, _ = _timed_loop(port, B, seconds_per_op)
        out[name] = {"port_frustums_per_s": rate}
        if ref is not None and O.have_ref():
            rate, _ = _timed_loop(ref, B, seconds_per_op)
            out[name]["reference_frustums_per_s"] = rate
    out["cores"] = 1
    out["note"] = "one host core; port = oracle/fpn_oracle.c, reference = the reference's own CPU loops " \
                  "(query_ball_point.cpp, interpolate.cpp) compiled into oracle/_ref"
    return out


# ---- point-set operator bandwidths ----------------------------------------------------

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
    from fpn import inference
    det_logits = torch.randn((B, N, 2), device=dev)
    det_box = torch.randn((B, 59), device=dev)
    det_c1 = torch.randn((B, 3), device=dev)
    det_rot = torch.rand((B,), device=dev)
    cases = {
        "farthest_point_sample(2048->128)": (lambda: tf_sampling.farthest_point_sample(M, xyz), 12 * N + 4 * M),
        "gather_point(128)": (lambda: tf_sampling.gather_point(xyz, idx), 4 * M + 24 * M),
        "query_ball_point(2048,128,r0.4,ns64)": (lambda: tf_grouping.query_ball_point(0.4, NS, xyz, new_xyz),
                                                 12 * (N + M) + 4 * M * NS + 4 * M),
        "group_point(c=64)": (lambda: tf_grouping.group_point(feat, gi), 4 * M * NS + 2 * 4 * M * NS * C),
        "three_nn(2048,128)": (lambda: tf_interpolate.three_nn(xyz, new_xyz), 12 * (N + M) + 24 * N),
        "three_nn_interpolate(2048,128,c=64)": (lambda: tf_interpolate.three_nn_interpolate(xyz, new_xyz, feat2),
                                                12 * (N + M) + 4 * M * C + 4 * N * C),
        "knn_point(2048,128,k=32)": (lambda: tf_grouping.knn_point(32, xyz, new_xyz), 12 * (N + M) + 8 * M * 32),
        # logits + box vector + centre + angle in, 20 doubles + the mask bits out
        "detection_records(2048)": (lambda: inference.detection_records(det_logits, det_box, det_c1, det_rot, True),
                                    8 * N + 59 * 4 + 12 + 4 + 160 + N // 8),
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
    # FPS is bound by the vector ALU, not by memory (its bytes are read once, its work is
    # (m-1) * n distance / minimum / maximum updates): the same time against the vector rate.
    # One update = 10 lane operations (3 subtractions, 3 products, 2 sums - no fused
    # multiply-add: each is rounded on its own, as the reference's arithmetic - 1 minimum,
    # 1 maximum); peak = 256 CUs x 128 lanes x 2.4 GHz single operations per second (half the
    # fused-multiply-add figure the fp32 vector peak of 157.3 TFLOP/s counts).
    fps = out["farthest_point_sample(2048->128)"]
    lane_ops = 10.0 * (M - 1) * N
    fps["algorithmic_lane_ops_per_frustum"] = lane_ops
    fps["achieved_Tlaneops"] = B * lane_ops / fps["ms"] / 1e9
    fps["frac_of_valu_peak"] = fps["achieved_Tlaneops"] / (256 * 128 * 2.4e9 / 1e12)
    # three_nn is bound the same way: every (query, known point) pair costs the squared distance
    # (8 operations, rounded one by one) and the reference's strict-"<" insertion into the three
    # running bests with their indices (3 comparisons, 10 selects): 21 lane operations a pair.
    for key in ("three_nn(2048,128)", "three_nn_interpolate(2048,128,c=64)"):
        if key in out:
            row = out[key]
            row["algorithmic_lane_ops_per_frustum"] = 21.0 * N * M
            row["achieved_Tlaneops"] = B * 21.0 * N * M / row["ms"] / 1e9
            row["frac_of_valu_peak"] = row["achieved_Tlaneops"] / (256 * 128 * 2.4e9 / 1e12)
    return out


# ---- workloads --------------------------------------------------------------------------

def build_v1(args, dev, rank, world):
    from fpn import dist as fdist
    from fpn import plan, synthetic, variables
    st = variables.VariableStore(dev, seed=1)        # identical weights on every rank (seeded)
    plan.init_variables(st, 4, True)
    st.randomize_batch_norm(2)
    bf16 = args.dtype == "bf16"
    net = plan.V1Plan(st, 4, True, chunk_frustums=args.chunk or None, precision="bf16" if bf16 else "fp32")
    b_local = args.batch or 1024
    points = args.points or 1024
    lo, hi = fdist.shard_range(world * b_local, rank, world)     # this rank's block of the global batch
    pc_np, oh_np = synthetic.make_frustums_range(lo, hi, points, 4, seed=0)
    pc, one_hot = torch.from_numpy(pc_np).to(dev), torch.from_numpy(oh_np).to(dev)
    macs = algorithmic_macs_per_frustum(points)
    rot = synthetic_rot_angles(lo, hi, dev)
    return {
        "step": (lambda: fdist.pack_results(net.forward(pc, one_hot))) if args.records == "raw"
        else (lambda: fdist.pack_detections(net.forward(pc, one_hot), rot)),
        "gather": "records" if args.records == "raw" else "detections", "frustums": b_local, "points": points,
        "cpu": "v1",
        "metric": "frustums/sec (node) FPointNet v1 N=%d" % points,
        "label": "Frustum PointNet v1 inference, %d frustums x %d pts x 4 ch per GPU, %s"
                 % (b_local, points, "bf16 MFMA with fp32 accumulation (not the headline configuration, which is "
                    "fp32)" if bf16 else "fp32 MFMA (BASELINE config 1)"),
        "workload_tag": "v1_%s_%dx%d" % (args.dtype, b_local, points) if not args.chunk else None,
        "extra_config": {"algorithmic_gflop_per_frustum": 2 * macs / 1e9,
                         "reference_graph_gflop_per_frustum": 2 * reference_macs_per_frustum_v1(points) / 1e9},
        "model_flops_per_frustum": 2.0 * macs,
    }


def synthetic_rot_angles(lo, hi, dev):
    """Frustum angles of frustums lo..hi-1 of the global set (one seeded stream for the set,
    like the points): what the label fields of a detection record are rotated back by."""
    import numpy as np
    rot = np.random.RandomState(11).uniform(-np.pi, np.pi, hi).astype(np.float32)[lo:hi]
    return torch.from_numpy(np.ascontiguousarray(rot)).to(dev)


def build_v2(args, dev, rank, world):
    from fpn import dist as fdist
    from fpn import synthetic, variables
    from fpn.models import frustum_pointnets_v2 as M2
    # (1024 = the chunk the streamed scene sets use; the per-frustum point-set kernels then put several
    # workgroups on a CU: 37.7 k against 35.7 k frustums/s at 256 in fp32, 132 k against 111 k in bf16)
    b_local = args.batch or 1024
    points = args.points or 2048
    lo, hi = fdist.shard_range(world * b_local, rank, world)
    pc_np, oh_np = synthetic.make_frustums_range(lo, hi, points, 4, seed=0)
    pc, one_hot = torch.from_numpy(pc_np).to(dev), torch.from_numpy(oh_np).to(dev)
    st = variables.VariableStore(dev, seed=1)
    from fpn import pointnet_util
    pointnet_util.SA_PRECISION = "bf16" if args.dtype == "bf16" else "fp32"      # the fused set-abstraction kernels
    with variables.use_store(st):
        M2.get_model(pc[:2], one_hot[:2], False)      # creates the variables
    st.randomize_batch_norm(2)

    def forward(pc_, oh_):
        with variables.use_store(st):
            return M2.get_model(pc_, oh_, False)
    raw = args.records == "raw"
    rot = synthetic_rot_angles(lo, hi, dev)
    pack = fdist.pack_results if raw else (lambda ep: fdist.pack_detections(ep, rot))      # noqa: E731
    step = lambda: pack(forward(pc, one_hot))       # noqa: E731
    label = "Frustum PointNet v2 (SA-MSG x2, SA, FP x3; box net SA x3) inference, %d frustums x %d pts x 4 ch " \
            "per GPU, %s" % (b_local, points, "bf16 MFMA in the set-abstraction kernels, fp32 elsewhere"
                             if args.dtype == "bf16" else "fp32")
    per_step = b_local
    if args.graph:
        from fpn.graph import GraphedForward
        graphed = GraphedForward(lambda p, o: pack(forward(p, o)), [pc, one_hot])
        step = lambda: graphed(pc, one_hot)       # noqa: E731
        label += ", HIP-graph replay"
    if args.stream_total:
        # a scene set larger than a batch, resident in (pinned) HOST memory, streamed through
        # the model in chunks of b_local frustums with the copies on a second stream
        reps = (args.stream_total + b_local - 1) // b_local
        host_pc = pc.cpu().repeat(reps, 1, 1)[:args.stream_total].contiguous().pin_memory()
        host_oh = one_hot.cpu().repeat(reps, 1)[:args.stream_total].contiguous().pin_memory()
        per_step = args.stream_total
        host_rot = None if raw else rot.cpu().repeat(reps)[:args.stream_total].contiguous()
        step = lambda: fdist.run_streamed(forward, host_pc, host_oh, b_local, dev, detections=not raw,       # noqa: E731
                                          rot_angle=host_rot)
        label += ", %d frustums per GPU streamed from pinned host memory in chunks of %d (host-to-device copies " \
                 "inside the timed region, overlapped with compute)" % (args.stream_total, b_local)
    plain = not (args.graph or args.stream_total)
    ref_macs, exe_macs = v2_macs_per_frustum(M2.CONFIG, points)
    return {"step": step, "gather": "records" if raw else "detections", "frustums": per_step, "points": points,
            "cpu": "v2",
            "metric": "frustums/sec (node) FPointNet v2 N=%d" % points, "label": label,
            "workload_tag": "v2_%s_%dx%d" % (args.dtype, b_local, points) if plain else None,
            "extra_config": {"algorithmic_gflop_per_frustum": 2 * exe_macs / 1e9,
                             "reference_graph_gflop_per_frustum": 2 * ref_macs / 1e9},
            "model_flops_per_frustum": 2.0 * exe_macs, "params": st.numpy}


def build_bayes(args, dev, rank, world):
    from fpn import bayes, synthetic, uncertainty, variables
    from fpn import dist as fdist
    b_local = args.batch or 128
    points = args.points or 1024
    lo, hi = fdist.shard_range(world * b_local, rank, world)
    pc_np, oh_np = synthetic.make_frustums_range(lo, hi, points, 3, seed=0)
    pc, one_hot = torch.from_numpy(pc_np).to(dev), torch.from_numpy(oh_np).to(dev)
    st = variables.VariableStore(dev, seed=1)
    bayes.init_variables(st, 3)
    bp = bayes.BayesPlan(st, 3, mc_dropout=args.mc_dropout, precision="bf16" if args.dtype == "bf16" else "fp32")
    gen = torch.Generator(device=dev)
    gen.manual_seed(7 + rank)          # the NOISE differs per rank (independent draws); the data is one global set

    def step():
        ep = bp.forward(pc, one_hot, args.samples, generator=gen)
        return uncertainty.mc_statistics(ep["box_output"], ep["stage1_center"])
    label = "Flipout Frustum PointNet v1, %d Monte-Carlo samples x %d frustums x %d pts per GPU, %s, " \
            "ensemble statistics on device%s" % (args.samples, b_local, points,
                                                 "bf16 MFMA in the convolution stacks (fp32 first layers, "
                                                 "maxima, head and fully connected layers)"
                                                 if args.dtype == "bf16" else "fp32",
                                                 ", Monte-Carlo dropout before the head" if args.mc_dropout else "")
    return {"step": step, "gather": "statistics", "frustums": b_local, "points": points, "cpu": "bayes",
            "metric": "frustums/sec (node) FPointNet Bayes S=%d N=%d" % (args.samples, points), "label": label,
            "workload_tag": "bayes_%s_S%d_%dx%d" % (args.dtype, args.samples, b_local, points),
            "extra_config": {"samples": args.samples,
                             "algorithmic_gflop_per_frustum": 2 * bayes_macs_per_frustum(points, args.samples)[1] / 1e9,
                             "reference_graph_gflop_per_frustum":
                                 2 * bayes_macs_per_frustum(points, args.samples)[0] / 1e9},
            "model_flops_per_frustum": 2.0 * bayes_macs_per_frustum(points, args.samples)[1]}


# ---- main ---------------------------------------------------------------------------------

def measure(args, dev, rank, world, extras=True, cpu_seconds=8.0):
    """Warm-up, timed region and the report of one workload (args.config, args.dtype, ...);
    returns the JSON object (complete on rank 0).  extras: also the point-set operator
    tables and the CPU baseline."""
    from fpn import dist as fdist
    from fpn import ops
    wl = {"v1": build_v1, "v2": build_v2, "bayes": build_bayes}[args.config](args, dev, rank, world)
    step = wl["step"]
    bf16 = args.dtype == "bf16"
    gathered = [None]
    last_work = [None]

    def full_step():
        rec = step()
        if world > 1:
            if REHEARSAL:       # gloo: collectives on host copies
                rec = rec.cpu()
            elif gathered[0] is None:
                gathered[0] = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=dev)
            # RCCL: the gather runs on the backend's stream beside the next step's kernels (the
            # fence at the end of the timed region waits for every stream); gloo: synchronous
            if wl["gather"] == "statistics":
                res = fdist.gather_statistics(rec, out=gathered[0], async_op=not REHEARSAL)
            elif wl["gather"] == "detections":
                res = fdist.gather_detections(rec, out=gathered[0], async_op=not REHEARSAL)
            else:
                res = fdist.gather_results(rec, out=gathered[0], async_op=not REHEARSAL)
            last_work[0] = None if REHEARSAL else res[1]

    def fence():
        if last_work[0] is not None:
            last_work[0].wait()
            last_work[0] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup + settle_steps(args.warmup)):
        full_step()
    # Which fused kernel takes the largest share of the step: untimed steps with HIP
    # events around every fused launch.  Inside the timed region only that kernel is
    # bracketed (event pairs around every launch would slow the measured step by ~2 %);
    # the per-kernel table comes from three more steps after the timed region.
    dominant = None
    if not args.no_kernel_timer:
        # (two probe steps, each kernel's faster one: an event pair also spans whatever the host did
        # between recording and launching, and one descheduled moment must not pick the kernel)
        agg = {}
        for _ in range(2):
            probe = KernelTimer()
            ops.launch_observer = probe
            step()
            ops.launch_observer = None
            for k, v in probe.summary().items():
                agg[k] = min(agg.get(k, float("inf")), v[2])
            probe.pending.clear()
        if agg:
            # (the generic per-layer names are buckets of many different layers, not one kernel shape: the
            # roofline is reported for a fused kernel whenever the step has one)
            fused = {k: v for k, v in agg.items() if not k.startswith("shared_mlp")} or agg
            dominant = max(fused, key=fused.get)
        # Release the probe's events before timing anything: with ~140 timing events of a v2 step
        # still alive every later launch of the process took ~50 us longer on the host (the step
        # became host-bound: 10.4 instead of 6.5 ms), as if the runtime had run out of pooled
        # completion signals; measured with tools/step_jitter.py and by dropping this object.
        del probe
    timer = KernelTimer(only={dominant}) if dominant else None
    ops.launch_observer = timer
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    fence()
    dt = time.perf_counter() - t0
    ops.launch_observer = None
    table = None
    if timer is not None and rank == 0:
        table = [KernelTimer() for _ in range(3)]
        for t in table:
            ops.launch_observer = t
            step()
        ops.launch_observer = None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if REHEARSAL else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ident = rank_identity(rank, dev, world)
    ranks = [ident]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, ident)

    value = world * wl["frustums"] * args.steps / dt
    config = {"workload": wl["label"], "frustums_per_gpu": wl["frustums"], "points": wl["points"],
              "data": "shards of one global seeded batch (fpn.synthetic.make_frustums_range, seed 0)",
              "parallelism": "frustum-sharded x%d, one all_gather of %s" % (
                  world, {"statistics": "%d-float ensemble statistics" % fdist.STATS,
                          "detections": "%d-double detection records (label fields, score, class picks; the record "
                                        "kernel runs inside the timed step)" % fdist.DETECTION,
                          "records": "%d-float result records" % fdist.RECORD}[wl["gather"]])}
    config.update(wl["extra_config"])
    config["workload_tag"] = wl["workload_tag"]
    if wl["model_flops_per_frustum"]:
        # (a bf16 step is priced against the bf16 peak although its first layers, fully connected layers and
        # point-set operators run in fp32: a lower bound of the fraction)
        peak_model = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        config["model_tflops"] = value * wl["model_flops_per_frustum"] / 1e12
        config["model_frac_of_mfma_peak"] = config["model_tflops"] / (peak_model * world)
        config["model_peak_tflops_per_gpu"] = peak_model
    out = {"metric": wl["metric"], "value": value, "unit": "frustums/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "settle_steps": settle_steps(args.warmup), "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
           "data": "synthetic", "config": config, "ranks": ranks,
           "distinct_devices": len({(r["host"], r["pci_bus_id"], r["uuid"]) for r in ranks})}
    if rank == 0:
        if timer is not None:
            agg = timer.summary()
            # an event pair also spans whatever the host did between recording the first
            # event and launching (a descheduled thread shows up as tens of ms on one
            # launch): the table sums the three steps, the ranking uses each kernel's
            # fastest step
            per_step = [t.summary() for t in table]
            full = {}
            for a in per_step:
                for k, v in a.items():
                    f = full.setdefault(k, [0, 0.0, 0.0, float("inf")])
                    f[0] += v[0]; f[1] += v[1]; f[2] += v[2]; f[3] = min(f[3], v[2])
            launches, flops, secs = agg[dominant]
            ach = flops / secs / 1e12
            ranked = [k for k in full if not k.startswith("shared_mlp")] or list(full)
            top = max(ranked, key=lambda k: full[k][3])
            if top != dominant:      # the timed kernel must be the one that dominates the step
                out["roofline_note"] = "largest share of the step in the post-run table: %s" % top
            peak = PEAK_BF16_MFMA_TFLOPS if "bf16" in dominant else PEAK_FP32_MFMA_TFLOPS
            traffic, traffic_src = measured_traffic(dominant, wl["workload_tag"])
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": traffic,
                               "traffic_source": traffic_src, "kernel": dominant,
                               "launches": launches, "avg_launch_ms": secs / launches * 1e3,
                               "algorithmic_gflop_per_launch": flops / launches / 1e9,
                               "share_of_step_time": secs / dt}
            # the peak is quoted at 2.4 GHz; the clock the chip held under this kernel when its
            # counters were recorded (frac itself stays against the nominal peak)
            ghz, ghz_src = recorded_clock(dominant, wl["workload_tag"])
            if ghz:
                out["roofline"].update({"recorded_clock_ghz": ghz, "clock_source": ghz_src,
                                        "frac_of_peak_at_recorded_clock": ach / (peak * ghz / 2.4)})
            out["kernels"] = {k: {"launches": v[0], "gflop": v[1] / 1e9, "ms": v[2] * 1e3,
                                  "ms_fastest_step": v[3] * 1e3,
                                  "tflops": v[1] / v[2] / 1e12 if v[2] > 0 else None} for k, v in full.items()}
            out["kernels_source"] = "3 steps after the timed region, HIP events around every fused launch"
        if extras and not args.no_pointset and args.config in ("v1", "v2"):
            # 256 frustums = one workgroup per CU for the per-frustum kernels (FPS, ball
            # query): latency-bound; 4096 frustums lets several workgroups share a CU
            out["pointset_ops"] = pointset_ops(dev)
            out["pointset_ops_4096_frustums"] = pointset_ops(dev, frustums=4096, reps=5)
            if not args.no_cpu_baseline:
                out["pointset_cpu"] = pointset_cpu()
        if extras and not args.no_cpu_baseline:       # rank 0, any world size: host timing needs no GPU
            out["cpu_baseline"] = cpu_baseline(wl["cpu"], seconds=cpu_seconds, samples=args.samples,
                                               params=wl["params"]() if "params" in wl else None)
    return out



def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="frustums per GPU (default: 1024 for v1 and v2, 128 for bayes)")
    ap.add_argument("--points", type=int, default=None, help="points per frustum (default: 1024; 2048 for v2)")
    ap.add_argument("--chunk", type=int, default=0, help="frustums per chunk (0 = plan default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-pointset", action="store_true", help="skip the point-set operator bandwidth table")
    ap.add_argument("--no-side-workloads", action="store_true",
                    help="default run only: skip the v2 and Bayes workloads reported under `side_workloads`")
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
    ap.add_argument("--records", choices=("detections", "raw"), default="detections",
                    help="what a step ends with and the all-gather carries: detections = the 20-double detection "
                         "records of fpn_detection_records (label fields, score, picks; default), raw = the 65-float "
                         "box vector + centres (no score without the logits)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a HIP graph (--config v2; removes the host launch overhead)")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.stream_total and args.config != "v2":
        raise SystemExit("--stream-total goes with --config v2")
    return args


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))         # this process never touches a GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = init_rank(local_rank, world)

    out = measure(args, dev, rank, world)
    if rank == 0:
        if world == 1 and args.config == "v1" and args.dtype == "f32" and not args.no_side_workloads \
                and not (args.batch or args.points or args.chunk):
            # the other GPU configurations BASELINE names, measured in this same run with the
            # same protocol (fewer steps): each is a complete report of its own under its key
            import copy
            out["side_workloads"] = {}
            for key, config, dtype, steps in (("v1_bf16", "v1", "bf16", 20), ("v2_f32", "v2", "f32", 10),
                                              ("v2_bf16", "v2", "bf16", 10), ("bayes_f32", "bayes", "f32", 3),
                                              ("bayes_bf16", "bayes", "bf16", 3)):
                side = copy.copy(args)
                side.config, side.dtype, side.steps, side.warmup = config, dtype, steps, 2
                side.no_pointset = True
                # (one CPU baseline per graph: the bf16 variants run the same graph as the f32 ones)
                out["side_workloads"][key] = measure(side, dev, rank, world, extras=dtype == "f32", cpu_seconds=4.0)
                torch.cuda.empty_cache()
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
