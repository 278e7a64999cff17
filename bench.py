"""Headline benchmark: images/s of forward + decode_outputs + post_process (NMS), CCNet `network.net.Net`
(s size, 6 classes) at 640x640, 64 images per GPU, bf16 storage / fp32 accumulate, synthetic images and weights.

    python bench.py [--gpus N --steps K --warmup W]      (N > 1 without WORLD_SIZE in the environment: this process starts
                                                          N child ranks itself, before it touches the GPU, and relays rank 0's line)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     (one rank per GPU, weak scaling)

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events around every conv launch of one extra
forward (on the stream the kernels run on); `cpu_baseline` times the CPU oracle (a port of the reference path) on a
bounded sample on rank 0 when N == 1.  Nothing here reads /root/reference.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md
FP32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
OP_CONV = 1
OP_CSP = 15   # a fused CSP layer (ccnet_csp_fused)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--kind", default="net")
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--conf", type=float, default=0.5)
    ap.add_argument("--nms", type=float, default=0.6)
    ap.add_argument("--cand-per-image", type=int, default=0,
                    help="place --conf so that about this many anchors per image survive the filter (BASELINE configs[4]: ~10000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", action="store_true", help="collect a step's detections after the next step has been submitted (post_process_async); "
                    "measured 1.3 %% faster over 100 steps and slower over 10, so the default is the reference's synchronous loop")
    ap.add_argument("--breakdown", action="store_true", help="print a per-op-kind time table to stderr")
    return ap.parse_args()


def build_model(kind, precision):
    import importlib
    from ccnet_amd import synth
    net = importlib.import_module("ccnet_amd.network." + kind).Net(0.33, 0.5, 6, "silu")
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0, bn_calib=kind))
    return net


def cpu_baseline(kind, size, conf, nms):
    """Oracle (CPU port of the reference path) on bounded samples, SURVEY 8(d): median of >= 5 passes after 2 warm-ups with all
    the cores this process may use and again with one thread (the reference server's stated intent,
    tools/socket2springboot.py:53-55); BASELINE configs[0] exactly (net, 224^2, bs1, conf 0.5 / nms 0.6) and the benched
    configuration at a reduced batch scaled to images/s."""
    import importlib
    import platform
    import statistics
    from ccnet_amd import synth
    from oracle import ccnet_oracle as orc
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)

    def measure(knd, sz, bs, threads, warm, reps, cf, nm):
        net = importlib.import_module("ccnet_amd.network." + knd).Net(0.33, 0.5, 6, "silu")
        sd = synth.synth_state_dict(net.state_dict(), seed=0, bn_calib=knd)
        torch.set_num_threads(threads)
        x = synth.synth_images(bs, sz, sz, seed=0)
        info = [[i, sz, sz] for i in range(bs)]
        ts = []
        for i in range(warm + reps):
            t0 = time.perf_counter()
            heads = orc.forward(knd, sd, x, 0.33, 0.5, "silu")
            dec = orc.decode_outputs(heads, [sz, sz])
            orc.post_process(dec, 6, [sz, sz], info, cf, nm)
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        return round(bs / statistics.median(ts), 3)

    cpu_model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    all_cores = measure(kind, size, 8, cores, 2, 5, conf, nms)
    one = measure(kind, size, 2, 1, 1, 3, conf, nms)
    c1_all = measure("net", 224, 1, cores, 2, 7, 0.5, 0.6)
    c1_one = measure("net", 224, 1, 1, 2, 7, 0.5, 0.6)
    torch.set_num_threads(cores)
    return {"value": all_cores, "unit": "images/s", "cores": cores, "kind": "port", "cpu": cpu_model,
            "sample": "%s %dx%d bs8 fp32 fwd+decode+NMS (conf %.3g, nms %.3g), median of 5 passes after 2 warm-ups, torch CPU oracle, %d threads"
                      % (kind, size, size, conf, nms, cores),
            "one_thread": {"value": one, "unit": "images/s", "cores": 1,
                           "sample": "%s %dx%d bs2, median of 3 passes after 1 warm-up, 1 thread" % (kind, size, size)},
            "config1_net_224_bs1": {"value": c1_all, "unit": "images/s", "cores": cores, "one_thread_value": c1_one,
                                    "sample": "BASELINE configs[0]: net 224x224 bs1 conf 0.5 nms 0.6, median of 7 passes after 2 warm-ups"}}


KIND_NAMES = {0: "other", 1: "conv", 2: "stem", 3: "dwconv", 4: "sppf", 5: "maxpool", 6: "cbam", 7: "eca", 8: "asff", 9: "layernorm",
              10: "attention", 14: "transformer_fused", 15: "csp_fused"}


def _entry(bound, work, ms, n, peak, unit, extra=None):
    """One roofline line: achieved = algorithmic work / measured time."""
    ach = work / (ms * 1e-3) / (1e12 if bound == "mfma" else 1e9) if ms > 0 else 0.0
    e = {"bound": bound, "n_launches": n, "ms_per_forward": round(ms, 4), "achieved": round(ach, 1), "peak": peak, "unit": unit,
         "frac": round(ach / peak, 4)}
    if extra:
        e.update(extra)
    return e


def _event_ms(fn, reps=20):
    """Average milliseconds of fn() between two events on torch's current stream (the stream the library launches on)."""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def kernel_rooflines(ms, infos, precision, outs, B, S, nc, conf, nms_thr, dev):
    """Per-kernel roofline entries for everything that is not a conv launch (north_star: achieved HBM GB/s for the elementwise,
    pooling, attention and NMS kernels; SURVEY 8(d) byte / pair definitions).  Plan ops come from the same HIP-event pass as the
    conv roofline; decode / filter / NMS are timed here with event pairs around repeated launches on resident tensors."""
    import math
    import numpy as np
    from ccnet_amd import ops
    from ccnet_amd.utils import bbox
    mfma_peak = BF16_MFMA_PEAK_TFLOPS if precision == "bf16" else FP32_MFMA_PEAK_TFLOPS
    agg = {}
    for m, (k, f, b) in zip(ms, infos):
        if k in (0, 1):
            continue
        a = agg.setdefault(k, [0.0, 0.0, 0.0, 0])
        a[0] += m; a[1] += f; a[2] += b; a[3] += 1
    out = {}
    for k, (m, f, b, n) in agg.items():
        name = KIND_NAMES.get(k, str(k))
        if k in (10, 14):   # attention core / fused Transformer: MFMA work
            out[name] = _entry("mfma", f, m, n, mfma_peak, "TFLOP/s", {"flops": f})
        else:
            extra = {"bytes": b}
            if k == 6:      # CBAM: unique bytes (1R + 1W of x) and the dependency-forced figure (3R + 1W), SURVEY 8(d)
                ach2 = 2.0 * b / (m * 1e-3) / 1e9 if m > 0 else 0.0
                extra["pass_forced_3r1w"] = {"bytes": 2.0 * b, "achieved": round(ach2, 1), "frac": round(ach2 / HBM_PEAK_GBS, 4)}
            out[name] = _entry("hbm", b, m, n, HBM_PEAK_GBS, "GB/s", extra)
    # ---- decode / filter / NMS on the tensors of the last step
    dec = bbox.decode_outputs(outs, [S, S])
    A = dec.shape[1]
    t_dec = _event_ms(lambda: bbox.decode_outputs(outs, [S, S]))
    dbytes = 2.0 * B * A * (5 + nc) * 4 + B * A * 4.0
    out["decode"] = _entry("hbm", dbytes, t_dec, 1, HBM_PEAK_GBS, "GB/s", {"bytes": dbytes})
    cand = torch.empty(B, A, 7, dtype=torch.float32, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    scores = bbox._scores_of(dec, dec, nc)   # what post_process uses: the per-anchor scores decode left beside the rows
    t_fil = _event_ms(lambda: ops.filter_candidates(dec, nc, conf, cand, counts, scores))
    cnt = counts.cpu().numpy()
    ncand = int(cnt.sum())
    fbytes = (B * A * 4.0 + (5 + nc) * 4.0 * ncand if scores is not None else B * A * (5 + nc) * 4.0) + 28.0 * ncand
    out["filter"] = _entry("hbm", fbytes, t_fil, 1, HBM_PEAK_GBS, "GB/s", {"bytes": fbytes, "candidates": ncand, "reads_scores": scores is not None})
    out["decode_plus_filter_ms"] = round(t_dec + t_fil, 4)
    nmax = max(int(cnt.max()), 1)
    hw = torch.full((B, 2), S, dtype=torch.int32, device=dev)
    t_nms = _event_ms(lambda: bbox._nms(cand, counts, nmax, nms_thr, hw))
    cls = cand[..., 6].cpu().numpy()
    pairs = 0.0
    for b_ in range(B):
        c_ = np.bincount(cls[b_, :cnt[b_]].astype(np.int64), minlength=nc).astype(np.float64)
        pairs += float((c_ * (c_ - 1) / 2).sum())
    out["nms"] = {"bound": "pairs", "n_launches": 1, "ms_per_forward": round(t_nms, 4), "candidates_per_image": round(ncand / B, 1),
                  "same_class_iou_pairs": pairs, "iou_pairs_per_s": round(pairs / (t_nms * 1e-3), 1) if t_nms > 0 else 0.0,
                  "note": "host workspace allocation included; latency-bound at this candidate count"}
    # ---- SURVEY 8(d) NMS-only microbench: 10 000 boxes per image x 16 images, log-normal sizes, 6 classes, nms 0.6
    rng = np.random.RandomState(0)
    nb, n = 16, 10000
    cnd = np.zeros((nb, n, 7), np.float32)
    c = rng.uniform(0, 1, (nb, n, 2)); wh = np.exp(rng.normal(math.log(0.05), 0.5, (nb, n, 2)))
    cnd[..., 0:2] = c - wh / 2; cnd[..., 2:4] = c + wh / 2
    cnd[..., 4] = rng.uniform(0, 1, (nb, n)); cnd[..., 5] = 1.0; cnd[..., 6] = rng.randint(0, 6, (nb, n))
    ct = torch.from_numpy(cnd).to(dev)
    cts = torch.full((nb,), n, dtype=torch.int32, device=dev)
    hw2 = torch.ones(nb, 2, dtype=torch.int32, device=dev)
    t_mb = _event_ms(lambda: bbox._nms(ct, cts, n, 0.6, hw2), reps=5)
    pairs_mb = 0.0
    for b_ in range(nb):
        c_ = np.bincount(cnd[b_, :, 6].astype(np.int64), minlength=6).astype(np.float64)
        pairs_mb += float((c_ * (c_ - 1) / 2).sum())
    out["nms_microbench"] = {"workload": "16 images x 10000 candidates, 6 classes, nms 0.6 (SURVEY 8d)", "ms": round(t_mb, 4),
                             "same_class_iou_pairs": pairs_mb, "iou_pairs_per_s": round(pairs_mb / (t_mb * 1e-3), 1),
                             "workspace_bytes": ops.nms_workspace_bytes(nb, n)}
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start N fresh child processes of this same command, one rank per
    GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would set them), wait for all
    of them and exit with the first failure.  The parent never initialises the GPU (torch.cuda.device_count() does not)."""
    import socket
    import subprocess
    rehearse = os.environ.get("CCNET_BENCH_REHEARSE", "0") == "1"
    have = torch.cuda.device_count()
    if have < n and not rehearse:
        sys.exit("bench.py --gpus %d: only %d GPU(s) visible (CCNET_BENCH_REHEARSE=1 runs every rank on cuda:0 with gloo, "
                 "a control-flow rehearsal whose numbers mean nothing)" % (n, have))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0's stdout is read here so that exactly its JSON line reaches this process' stdout (gloo and RCCL print
        # connection notes of their own on stdout); everything else goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    for line in procs[0].stdout:
        (sys.stdout if line.startswith('{"metric"') else sys.stderr).write(line)
        sys.stdout.flush()
    rc = 0
    for p_ in procs:
        rc = rc or p_.wait()
    sys.exit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and "WORLD_SIZE" in os.environ and int(os.environ.get("RANK", "0")) == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size is used" % (args.gpus, world), file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # CCNET_BENCH_REHEARSE=1: every rank on cuda:0 with gloo collectives -- a one-GPU rehearsal of the multi-rank control flow
    # (barriers, max-over-ranks timing, detections all_gather); its numbers mean nothing
    rehearse = world > 1 and os.environ.get("CCNET_BENCH_REHEARSE", "0") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if world > 1 and not rehearse else 0)
    cdev = torch.device("cpu") if rehearse else dev  # where collective payloads live
    torch.cuda.set_device(dev)

    from ccnet_amd import synth
    from ccnet_amd.parallel import all_gather_detections, sharded_post_process
    host_exchange = os.environ.get("CCNET_BENCH_EXCHANGE", "device") == "host"   # fallback: round 1's host-side all_gather
    from ccnet_amd.utils import bbox

    B, S = args.batch, args.size
    net = build_model(args.kind, args.precision).to(dev).set_precision(args.precision)
    # every rank owns its own slice of the global synthetic batch: B DISTINCT images (image i depends only on its global index)
    x = torch.cat([synth.synth_images(1, S, S, seed=1000 + rank * B + i) for i in range(B)], 0).contiguous().to(dev)
    info = [[i, S, S] for i in range(B)]

    if args.cand_per_image > 0:
        # the confidence threshold is placed where ~N of the anchors per image survive (the random-weight
        # net has no meaningful operating point of its own); done once, before anything is timed
        dec0 = bbox.decode_outputs(net(x), [S, S])
        score = (dec0[..., 4] * dec0[..., 5:].max(-1).values).flatten()
        k = min(max(score.numel() - args.cand_per_image * B, 1), score.numel())
        args.conf = float(score.kthvalue(k).values)
        del dec0, score

    def step():
        outs = net(x)
        dec = bbox.decode_outputs(outs, [S, S])
        if world > 1 and host_exchange:
            return all_gather_detections(bbox.post_process(dec, 6, [S, S], info, args.conf, args.nms), device=cdev, max_local=B)
        if world > 1:
            # local filter + NMS, then ONE all_gather of the device result buffer (RCCL with the nccl backend) and one D2H copy
            return sharded_post_process(dec, 6, [S, S], info, args.conf, args.nms)
        return bbox.post_process(dec, 6, [S, S], info, args.conf, args.nms)

    def submit():
        """forward + decode + filter + NMS + device->host copy of one batch, all asynchronous (utils.bbox.post_process_async)"""
        return bbox.post_process_async(bbox.decode_outputs(net(x), [S, S]), 6, [S, S], info, args.conf, args.nms)

    def run_steps(k, pipelined):
        """k steps.  Pipelined (one GPU): a step's detections are collected after the NEXT step has been submitted, so the GPU
        works through the host's round trip; every step's detections are collected before this returns."""
        last = None
        if not pipelined:
            for _ in range(k):
                last = step()
            return last
        pend = None
        for _ in range(k):
            nxt = submit()
            if pend is not None:
                last = pend.result()
            pend = nxt
        return pend.result() if pend is not None else last

    pipelined = world == 1 and args.pipeline

    def timed(k, pipe):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = run_steps(k, pipe)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t0, d

    run_steps(args.warmup, pipelined)
    dt, dets = timed(args.steps, pipelined)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    sync_ms = None
    if pipelined:  # the same K steps with every step's detections collected before the next forward is launched
        sync_ms = timed(args.steps, False)[0] / args.steps * 1e3
    pipe_ms = None
    if world == 1 and not pipelined:
        # beside the headline (synchronous loop): the same K steps with a step's detections collected after the next step has been
        # submitted, the way a serving loop would run them -- reported, never `value`
        pipe_ms = timed(args.steps, True)[0] / args.steps * 1e3
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    # ---- roofline of the dominant kernel family (conv_igemm): HIP events around every op of one forward
    plan, _, _ = net.plan_for(B, S, S, dev)
    plan.run_timed(x)
    ms = None
    for _ in range(3):
        cur = plan.run_timed(x)
        ms = cur if ms is None else [min(a, b) for a, b in zip(ms, cur)]
    infos = plan.op_info()
    # The conv/linear launches fall on both sides of the ridge (peak FLOP/s / peak B/s = 312 FLOP/B for bf16): the 3x3 / 6x6
    # layers are MFMA-bound and dominate the time, the 1x1 layers and transformer linears are HBM-bound.  The headline
    # roofline is the MFMA-bound group; the HBM-bound group is reported beside it against the HBM peak.
    peak = BF16_MFMA_PEAK_TFLOPS if args.precision == "bf16" else FP32_MFMA_PEAK_TFLOPS
    ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    # A fused CSP layer (five convs in one launch) is priced against HBM whatever its FLOP/B says: one read of x and one write of
    # out is what it is built for, and its two-input form sits exactly on the ridge (313 FLOP/B).  It is counted in the HBM-bound
    # group, as in earlier rounds, and has a line of its own under "kernels".
    conv = [(m, f, b, k) for m, (k, f, b) in zip(ms, infos) if k in (OP_CONV, OP_CSP)]
    hot = [(m, f, b) for (m, f, b, k) in conv if k == OP_CONV and f / b >= ridge]
    thin = [(m, f, b) for (m, f, b, k) in conv if k == OP_CSP or f / b < ridge]
    conv = [(m, f, b) for (m, f, b, k) in conv]
    hot_ms, hot_fl = sum(m for m, f, b in hot), sum(f for m, f, b in hot)
    thin_ms, thin_by = sum(m for m, f, b in thin), sum(b for m, f, b in thin)
    conv_ms, conv_fl = sum(m for m, f, b in conv), sum(f for m, f, b in conv)
    achieved = hot_fl / (hot_ms * 1e-3) / 1e12 if hot_ms > 0 else 0.0
    thin_gbs = thin_by / (thin_ms * 1e-3) / 1e9 if thin_ms > 0 else 0.0
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": None,
                "kernel": "conv_halo_kernel (3x3 / SpaceToDepth layers) + bneck_fused_kernel (where taken) + conv_patch_kernel + conv_igemm_kernel, the %d MFMA-bound launches of one forward (>= %d FLOP/B)" % (len(hot), int(ridge)),
                "avg_launch_us": round(hot_ms * 1e3 / max(len(hot), 1), 2), "flops_per_launch": hot_fl / max(len(hot), 1),
                "ms_per_forward": round(hot_ms, 3),
                "hbm_bound_launches": {"n": len(thin), "ms_per_forward": round(thin_ms, 3), "bound": "hbm", "achieved": round(thin_gbs, 1),
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(thin_gbs / HBM_PEAK_GBS, 4)},
                "all_conv_launches": {"n": len(conv), "ms_per_forward": round(conv_ms, 3),
                                      "tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2) if conv_ms > 0 else 0.0},
                "forward_ms_sum_of_ops": round(sum(ms), 3)}
    # HBM bytes per launch come from separate rocprofv3 --pmc passes over this same command (scripts/profile_bench.sh,
    # scripts/traffic_from_pmc.py); the newest committed summary is attached, null if there is none
    import glob
    tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_conv_traffic.json")))
    if tf and args.kind == "net" and args.precision == "bf16" and B == 64 and S == 640:
        with open(tf[-1]) as f:
            tj = json.load(f)
        roofline["traffic"] = round(tj["hot"]["hbm_bytes_per_launch"])
        roofline["traffic_source"] = "committed profile profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command), not measured in this run" % os.path.basename(tf[-1])
        roofline["traffic_note"] = "%s: HBM-side bytes per MFMA-bound launch (%s); algorithmic bytes per launch %d" % (
            os.path.basename(tf[-1]), tj["correction"], round(tj["hot"]["algorithmic_bytes_per_launch"]))
        roofline["hbm_bound_launches"]["traffic"] = round(tj["thin"]["hbm_bytes_per_launch"])
    if args.breakdown and rank == 0:
        def timed(fn, n=5):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n * 1e3, r
        t_fwd, outs = timed(lambda: net(x))
        t_dec, dec = timed(lambda: bbox.decode_outputs(outs, [S, S]))
        t_pp, _ = timed(lambda: bbox.post_process(dec, 6, [S, S], info, args.conf, args.nms))
        print("phases: forward %.3f ms  decode %.3f ms  post_process %.3f ms" % (t_fwd, t_dec, t_pp), file=sys.stderr)
        names = {0: "other", 1: "conv", 2: "stem", 3: "dwconv", 4: "sppf", 5: "maxpool", 6: "cbam", 7: "eca", 8: "asff", 9: "layernorm",
                 10: "attention", 14: "transformer", 15: "csp_fused"}
        agg = {}
        for m, (k, f, b) in zip(ms, infos):
            a = agg.setdefault(k, [0.0, 0.0, 0.0, 0])
            a[0] += m; a[1] += f; a[2] += b; a[3] += 1
        for k, (m, f, b, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            print("%-10s n=%3d  %8.3f ms  %8.1f TFLOP/s  %8.1f GB/s(alg)" % (names.get(k, k), n, m, f / (m * 1e-3) / 1e12 if m else 0,
                                                                             b / (m * 1e-3) / 1e9 if m else 0), file=sys.stderr)
        for i, (m, (k, f, b)) in enumerate(zip(ms, infos)):
            print("  op %3d %-9s %7.3f ms %7.1f TF/s %7.1f GB/s  %8.2f GF %8.2f MB" % (i, names.get(k, k), m, f / (m * 1e-3) / 1e12,
                                                                                     b / (m * 1e-3) / 1e9, f / 1e9, b / 1e6), file=sys.stderr)

    out = {"metric": "images/sec (fwd+decode+NMS) %dx%d bs%d per GPU" % (S, S, B), "value": round(value, 2), "unit": "images/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
           "config": {"workload": "CCNet %s (s: depth 0.33, width 0.5, 6 classes) %dx%d, %d images/GPU, fwd+decode+post_process "
                                  "(conf %.3g, nms %.3g)%s" % (args.kind, S, S, B, args.conf, args.nms,
                                                                (", all_gather of detections (%s)" % ("host buffers" if host_exchange else "gloo rehearsal on one GPU" if rehearse
                                                                                              else "RCCL all_gather_into_tensor")) if world > 1 else ""),
                      "global_batch": world * B, "parallelism": "batch-sharded replicas x%d" % world,
                      "detections_last_step": int(sum(0 if d is None else d.shape[0] for d in dets)),
                      "step_loop": ("pipelined: forward + decode + filter + NMS + D2H of step i are submitted before the detections of step "
                                    "i-1 are collected on the host; all K batches are complete inside the timed region" if pipelined else
                                    "synchronous: each step's detections are on the host before the next forward is launched")},
           "roofline": roofline}
    if sync_ms is not None:
        out["sync_ms_per_step"] = round(sync_ms, 3)  # the same K steps without pipelining (the reference's loop shape)
    if pipe_ms is not None:
        out["pipelined_loop"] = {"ms_per_step": round(pipe_ms, 3), "value": round(B * 1e3 / pipe_ms, 2), "unit": "images/s",
                                 "note": "same K steps, detections of step i collected after step i+1 was submitted (post_process_async); not the headline"}
    if rank == 0:
        out["kernels"] = kernel_rooflines(ms, infos, args.precision, net(x), B, S, 6, args.conf, args.nms, dev)
    if rank == 0 and world == 1:
        # the step in front of the path, reported beside the headline and not part of `value`: B NEU-DET-sized
        # 8-bit images resident in HBM -> the (B,3,S,S) fp32 batch (Pillow-exact bicubic + normalisation + CHW)
        from ccnet_amd.utils import preprocess as pp
        src = torch.randint(0, 256, (B, 200, 200, 3), dtype=torch.uint8, device=dev)
        dst = torch.empty((B, 3, S, S), dtype=torch.float32, device=dev)
        for _ in range(3):
            pp.preprocess_group(src, dst, (S, S))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            pp.preprocess_group(src, dst, (S, S))
        e1.record()
        torch.cuda.synchronize()
        pms = e0.elapsed_time(e1) / 20
        pbytes = B * (200 * 200 * 3 + 3 * S * S * 4)
        out["preprocess"] = {"workload": "%d x 200x200x3 u8 -> %dx%d fp32 CHW, bicubic (Pillow-exact)" % (B, S, S), "ms": round(pms, 4),
                             "bound": "hbm", "achieved": round(pbytes / pms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(pbytes / pms / 1e6 / HBM_PEAK_GBS, 4), "in_value": False}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.kind, S, args.conf, args.nms)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
