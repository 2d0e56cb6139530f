#!/usr/bin/env python
"""Benchmark of the EMM tracker-head hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--tracks 30]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one frame pair through the hot path (BASELINE.json north_star):
``EMM.forward`` on frame t (ROIAlign of the search regions with virtual padding -> depthwise
cross-correlation -> prediction towers + heads -> fused up-sample/decode) followed by
``EMM.extract_cache`` on frame t (template ROIAlign + search regions for frame t+1), at
BASELINE.json configs[1]: DLA-34-FPN feature maps of a 720p frame (net input 704x1280, C=128,
5 FPN levels), 30 active tracks.  Inputs are synthetic, seeded, and RESIDENT IN HBM before the timed
region.  The DLA-34-FPN backbone, RPN and box head are NOT in the timed region (they are PyTorch
modules outside the hot path) — `config.workload` says so.

Multi-GPU (--gpus N): one process per GPU, each running its own independent stream;
the predictor weights are broadcast once from rank 0 over RCCL before timing; no per-frame
collective.  Weak scaling: value = N * K / max-over-ranks(elapsed).

The JSON line also carries:
  roofline     — for the depthwise cross-correlation kernel (the graded kernel): algorithmic bytes
                 4*N*C*(Rx^2+Rz^2+Ho^2) per launch / average launch duration measured with HIP events
                 recorded on the launch stream around every xcorr launch of the timed region.
  cpu_baseline — the CPU oracle (oracle/emm_oracle.py, the reference's torch-CPU ops) timed on this
                 host's cores on the same workload (rank 0, N=1 only), bounded to ~10-20 s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
METRIC = "frame-pairs/sec at 720p, 30 active tracks; EMM xcorr HBM GB/s vs peak"
NET_HW = (704, 1280)       # 720p under MIN_SIZE_TEST 800 / MAX 1280 / divisibility 32
CHANNELS = 128
TIMER_STRIDE = 16          # kernel event brackets on every 16th step of the timed region
TRACK_SIZES = [(32, 64), (64, 128), (100, 200), (160, 320)]    # (w,h): FPN levels 0,0,1,2


def synthetic_boxes(n, image_wh):
    """Non-overlapping grid of boxes cycling the four sizes, fully inside the image."""
    W, H = image_wh
    cols = max(1, W // 180)
    boxes = []
    for i in range(n):
        w, h = TRACK_SIZES[i % 4]
        cx = 90 + 180 * (i % cols)
        cy = 170 + 340 * ((i // cols) % max(1, H // 340))
        # tracks beyond the grid capacity wrap with a small offset (crowd regime, configs[2])
        off = 7.0 * (i // (cols * max(1, H // 340)))
        x1 = min(max(cx - w / 2 + off, 0), W - w - 1)
        y1 = min(max(cy - h / 2 + off, 0), H - h - 1)
        boxes.append([x1, y1, x1 + w, y1 + h])
    return torch.tensor(boxes, dtype=torch.float32)


def synthetic_features(seed, device):
    g = torch.Generator().manual_seed(seed)
    H, W = NET_HW
    return tuple(torch.randn((1, CHANNELS, H // s, W // s), generator=g).to(device) for s in (4, 8, 16, 32, 64))


def fused_algorithmic_bytes(boxes, image_wh, channels, rz=15, rx=30, pad_pixels=512, search_expansion=1.0):
    """Compulsory HBM bytes of the fused search-region-pool + xcorr kernel for these tracks (SURVEY.md §8d
    formula B restricted to that kernel): every feature cell of a track's search window read once
    (virtual zero border excluded), its template read once, its response written once.
        fp(sr, l) = (ceil(x2 s) - floor(x1 s) + 1) * (ceil(y2 s) - floor(y1 s) + 1), clipped to the real map."""
    import math
    W, H = image_wh
    total_cells = 0
    for x1, y1, x2, y2 in boxes.tolist():
        s = math.sqrt((x2 - x1 + 1) * (y2 - y1 + 1))
        lvl = int(min(max(math.floor(4 + math.log2(s / 224 + 1e-6)), 2), 5)) - 2
        scale = 1.0 / (4 * 2 ** lvl)
        w, h = x2 - x1 + 1, y2 - y1 + 1
        wx, hy = w * search_expansion / 2.0, h * search_expansion / 2.0
        sx1, sy1, sx2, sy2 = x1 - wx, y1 - hy, x2 + wx, y2 + hy            # un-padded image coordinates
        mw, mh = W // (4 * 2 ** lvl), H // (4 * 2 ** lvl)
        cx1, cx2 = max(math.floor(sx1 * scale), 0), min(math.ceil(sx2 * scale), mw - 1)
        cy1, cy2 = max(math.floor(sy1 * scale), 0), min(math.ceil(sy2 * scale), mh - 1)
        total_cells += max(cx2 - cx1 + 1, 0) * max(cy2 - cy1 + 1, 0)
    ho = rx - rz + 1
    n = boxes.shape[0]
    return 4.0 * channels * total_cells + 4.0 * n * channels * (rz * rz + ho * ho)


def init_predictor(pred, boxes):
    """Random-init weights of the reference architecture (there are no checkpoints offline), with
    biases that keep the decode non-degenerate (SURVEY.md §7 'Degenerate synthetic weights')."""
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for name, p in pred.named_parameters():
            if name.endswith("0.weight") or name in ("cls.weight", "center.weight", "reg.weight"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        mw = float((boxes[:, 2] - boxes[:, 0]).mean())
        mh = float((boxes[:, 3] - boxes[:, 1]).mean())
        pred.reg.bias.copy_(torch.tensor([0.5 * mw, 0.5 * mh, 0.5 * mw, 0.5 * mh]))


def _cpu_threads():
    """Threads for the CPU leg: the cores this process may actually run on (cgroup/affinity aware —
    os.cpu_count() reports the whole host and oversubscribing OpenMP stalls for minutes), capped at 32
    (the workload's tensors are small; more threads only add fork/join cost)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))


def cpu_baseline_worker(n_tracks, budget_s):
    """Time the oracle (reference torch-CPU ops) on the same frame-pair workload; prints one JSON line."""
    from oracle import emm_oracle as O            # checker / baseline only — never the product path
    threads = _cpu_threads()
    torch.set_num_threads(threads)
    feats = synthetic_features(0, "cpu")
    boxes = synthetic_boxes(n_tracks, (NET_HW[1], NET_HW[0]))
    cfg = O.EMMConfig(channels=CHANNELS)
    from siammot_amd.config import get_default_cfg
    from siammot_amd.emm import EMMPredictor
    pred = EMMPredictor(get_default_cfg(channels=CHANNELS))
    init_predictor(pred, boxes)
    params = {k: v.detach() for k, v in pred.named_parameters()}

    def step():
        z, sr = O.extract_cache(cfg, feats, boxes)
        return O.emm_forward(cfg, params, feats, boxes, sr, z, (NET_HW[1], NET_HW[0]), reference_ops=True)

    with torch.no_grad():
        step()
        t0 = time.perf_counter()
        step()
        one = time.perf_counter() - t0
        reps = int(max(3, min(50, budget_s / max(one, 1e-3))))
        times = []
        t_start = time.perf_counter()
        for _ in range(reps):
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > 2 * budget_s:
                break
    times.sort()
    med = times[len(times) // 2]
    print(json.dumps({
        "value": 1.0 / med, "unit": "frame-pairs/s", "cores": threads, "kind": "port",
        "sample": "%d frame pairs of the same workload (%d tracks, 720p maps), median; oracle/emm_oracle.py "
                  "with the reference's torch-CPU ops (grouped conv2d, F.interpolate, physical pad_feature), "
                  "%d threads" % (len(times), n_tracks, threads),
        "ms_per_step": med * 1e3}))


def cpu_baseline(n_tracks, budget_s=10.0, timeout_s=150.0):
    """Run the CPU leg in a child process with a hard timeout so that bench.py always finishes."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--tracks", str(n_tracks),
           "--cpu-budget", str(budget_s), "--channels", str(CHANNELS), "--net-hw", str(NET_HW[0]), str(NET_HW[1])]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        for line in reversed(res.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "frame-pairs/s", "cores": _cpu_threads(), "kind": "port",
                "sample": "cpu leg failed: %s" % (res.stderr.strip().splitlines() or ["no output"])[-1]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "frame-pairs/s", "cores": _cpu_threads(), "kind": "port",
                "sample": "cpu leg exceeded %.0f s and was stopped" % timeout_s}


def multi_stream_throughput(emm, feats, det, n_streams, dev, steps=600):
    """S independent video streams (own track memory) on S HIP streams of one GPU: the serving configuration of
    SURVEY.md §8(e) when a GPU hosts more than one camera.  Kernels of different streams overlap, filling the
    ramp-up / tail of each other's launches.  Informational: `value` stays the single-stream number."""
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    with torch.no_grad():
        states = []
        for st in streams:
            with torch.cuda.stream(st):
                states.append(emm.extract_cache(feats[1], det))

        def run(k_steps):
            for k in range(k_steps):
                for i, st in enumerate(streams):
                    with torch.cuda.stream(st):
                        z, sr, d = states[i]
                        emm(feats[k & 1], d, sr, template_features=z)
                        states[i] = emm.extract_cache(feats[k & 1], det)
        run(100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"streams": n_streams, "steps_per_stream": steps, "value": n_streams * steps / dt, "unit": "frame-pairs/s",
            "ms_per_step_per_stream": dt / steps * 1e3}


def tracking_loop_throughput(n, dev, feats, steps=300):
    """The whole tracker around the head (siammot_amd.track_head.TrackingLoop): EMM.forward -> merge with this
    frame's detections -> solver (score-banded NMS, id life cycle, ONE host sync) -> EMM.extract_cache + track
    memory.  Detections are the n synthetic track boxes jittered by a pixel, so every track survives and the
    track count stays n.  Informational (the solver is host-bound control logic, not part of the metric)."""
    from siammot_amd.config import get_default_cfg
    from siammot_amd.structures import BoxList
    from siammot_amd.track_head import build_tracking_loop
    image_wh = (NET_HW[1], NET_HW[0])
    boxes = synthetic_boxes(n, image_wh).to(dev)
    loop = build_tracking_loop(get_default_cfg(channels=CHANNELS), device=dev)
    init_predictor(loop.track.tracker.predictor, boxes.cpu())
    loop.track.tracker.to(dev)

    def dets(k):
        d = BoxList(boxes + float(k & 1), image_wh, mode="xyxy")
        d.add_field("ids", torch.full((n,), -1, dtype=torch.int64, device=dev))
        d.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))
        d.add_field("scores", torch.full((n,), 0.9, device=dev))
        return d
    for k in range(30):
        out = loop(feats[k & 1], dets(k))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        out = loop(feats[k & 1], dets(k))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "frames/s", "ms_per_frame": dt / steps * 1e3, "tracks": n,
            "tracked_in_last_frame": int((out.get_field("ids") >= 0).sum().item()),
            "note": "head + solver + track memory, synthetic detections; host-bound (one sync per frame)"}


def tower_roofline(n, total_ms, launches, bracket_us):
    algo = 2.0 * n * 2 * CHANNELS * 256 * 9 * CHANNELS
    executed = algo * 16.0 / 36.0
    sec = total_ms * 1e-3 / launches                     # raw event span (includes part of the bracket's own span)
    return {
        "bound": "mfma", "kernel": "tower_wino_kernel<0> (Winograd F(2x2,3x3), v_mfma_f32_16x16x4_f32)",
        "flops_per_launch": algo, "avg_launch_us": sec * 1e6, "event_bracket_overhead_us": bracket_us,
        "achieved": algo / sec / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": algo / sec / 1e12 / 157.3,
        "executed_mfma_flops_per_launch": executed, "executed_frac_of_peak": executed / sec / 1e12 / 157.3,
        "launches_timed": launches,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="setup phase: run untimed frame pairs for this long so host and device clocks leave their "
                         "idle states before the W warm-up steps (a 200-step run measured 0.29 ms/step, a 2000-step "
                         "run 0.16 ms/step on the same box); 0 disables")
    ap.add_argument("--tracks", type=int, default=30)
    ap.add_argument("--channels", type=int, default=CHANNELS,
                    help="FPN channel count (128 = DLA-34-FPN, the metric's configuration; 256 = R-50-FPN, configs[4])")
    ap.add_argument("--net-hw", type=int, nargs=2, default=list(NET_HW), metavar=("H", "W"),
                    help="network input size (704 1280 = 720p under the default resize rule; 1056 1920 = configs[4])")
    ap.add_argument("--extra-streams", type=int, default=2,
                    help="after the timed region, also measure S independent video streams on S HIP streams of the same "
                         "GPU (reported as `multi_stream`, not as `value`); 0 disables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true",
                    help="skip the HIP-event bracketing of the xcorr launches (roofline fields become null)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=10.0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    globals()["CHANNELS"] = args.channels            # the helpers above read the module constants
    globals()["NET_HW"] = tuple(args.net_hw)
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.tracks, args.cpu_budget)
        return

    from siammot_amd import ops, parallel
    from siammot_amd.config import get_default_cfg
    from siammot_amd.emm import EMM
    from siammot_amd.structures import BoxList
    from siammot_amd.track_utils import build_track_utils

    rank, world, local_rank = parallel.init_distributed()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (the product path has no CPU fallback)"
    dev_index = local_rank % torch.cuda.device_count()       # ranks > GPUs only in the gloo smoke test
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ops.load_library()

    n = args.tracks
    image_wh = (NET_HW[1], NET_HW[0])
    boxes_cpu = synthetic_boxes(n, image_wh)
    cfg = get_default_cfg(channels=CHANNELS)
    emm = EMM(cfg, build_track_utils(cfg)).eval()
    if rank == 0:
        init_predictor(emm.predictor, boxes_cpu)
    emm = emm.to(dev)
    bcast_bytes = parallel.broadcast_module(emm, src=0)       # one RCCL broadcast over xGMI; 0 at N=1
    feats = [synthetic_features(100 + rank * 2 + k, dev) for k in range(2)]   # two alternating frames
    det = BoxList(boxes_cpu.to(dev), image_wh, mode="xyxy")
    det.add_field("ids", torch.arange(n, device=dev))
    det.add_field("labels", torch.ones(n, dtype=torch.int64, device=dev))

    def step(k, state):
        z, sr, d = state
        _, result, _ = emm(feats[k & 1], d, sr, template_features=z)       # frame t: track
        return emm.extract_cache(feats[k & 1], det), result                # frame t: new templates / SRs

    with torch.no_grad():
        state = emm.extract_cache(feats[1], det)
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:     # setup, not part of W or K
            for k in range(32):
                state, _ = step(k, state)
            torch.cuda.synchronize()
        for k in range(args.warmup):
            state, _ = step(k, state)
        if not args.no_kernel_timer:
            # events are created here, outside the timed region.  An event pair costs ~3 us of stream time, so the
            # two instrumented kernels are bracketed on every TIMER_STRIDE-th step only (bracketing all of them
            # slows the frame pair from 80 to 94 us); the samples still come from inside the timed region.
            ops.kernel_timer_begin(ops.TIMER_XCORR, args.steps, TIMER_STRIDE)
            ops.kernel_timer_begin(ops.TIMER_TOWER, args.steps, TIMER_STRIDE)
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            state, result = step(k, state)
        torch.cuda.synchronize()
        parallel.barrier()
        elapsed = time.perf_counter() - t0
        xcorr_total_ms, xcorr_launches = (0.0, 0) if args.no_kernel_timer else ops.kernel_timer_end(ops.TIMER_XCORR)
        tower_total_ms, tower_launches = (0.0, 0) if args.no_kernel_timer else ops.kernel_timer_end(ops.TIMER_TOWER)
    multi = loop_stats = None
    if world == 1 and args.extra_streams > 1:
        multi = multi_stream_throughput(emm, feats, det, args.extra_streams, dev)
        loop_stats = tracking_loop_throughput(n, dev, feats)
    elapsed = parallel.max_over_ranks(elapsed, dev)
    # A bracketed span = kernel + part of the bracket's own span.  The span of an EMPTY bracket on the same stream
    # (two hipEventRecords back to back, ~4.6 us on MI355X) is reported next to the spans as an upper bound of that
    # share; `achieved` uses the RAW spans (conservative: rocprofv3 durations in profiles/ are 2-3 us shorter, and
    # raw minus empty-bracket is 2 us shorter still than rocprofv3).
    bracket_us = 0.0 if args.no_kernel_timer else ops.kernel_timer_bracket_overhead(200)
    xcorr_avg_s = xcorr_total_ms * 1e-3 / max(xcorr_launches, 1)

    if rank != 0:
        return
    rx, rz = emm.rx, emm.rz
    ho = rx - rz + 1
    # the kernel that runs in the pipeline: search-region pooling fused with the cross-correlation
    fused_bytes = fused_algorithmic_bytes(boxes_cpu, image_wh, CHANNELS, rz, rx)
    achieved = fused_bytes / xcorr_avg_s / 1e9 if xcorr_avg_s > 0 else 0.0
    # the stand-alone xcorr operator (SURVEY.md §8d figure A) on the same search regions, timed the same way
    xcorr_bytes = 4.0 * n * CHANNELS * (rx * rx + rz * rz + ho * ho)
    xop = None
    if not args.no_kernel_timer:
        with torch.no_grad():
            z, sr, d = state
            x = ops.roi_align_levels(feats[0], sr[0].bbox, d[0].bbox, rx, emm.feature_extractor.pooler_x.scales, 2,
                                     [int(emm.pad_pixels / ((2 ** i) * 4)) for i in range(4)])
            for _ in range(20):
                ops.xcorr_depthwise(x, z)
            reps = 200
            ops.xcorr_timer_begin(reps)
            for _ in range(reps):
                ops.xcorr_depthwise(x, z)
            torch.cuda.synchronize()
            ms, cnt = ops.xcorr_timer_end()
        t = ms * 1e-3 / max(cnt, 1)
        xop = {"kernel": "xcorr_dw_patch2_kernel<30,15,0>", "algorithmic_bytes_per_launch": xcorr_bytes,
               "avg_launch_us": t * 1e6, "achieved": xcorr_bytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": xcorr_bytes / t / 1e9 / HBM_PEAK_GBS, "launches_timed": cnt,
               "note": "stand-alone operator, input resident in L2/MALL (not part of the frame-pair pipeline, which "
                       "runs the fused kernel)"}
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "xcorr_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(str(n))
        except Exception:
            traffic = None
    out = {
        "metric": METRIC,
        "value": world * args.steps / elapsed,
        "unit": "frame-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "EMM tracker-head frame pair (EMM.forward + EMM.extract_cache) on %s FPN maps "
                        "(net input %dx%d, C=%d, 5 levels), %d tracks, one stream per GPU; hot path only: "
                        "backbone / RPN / box head / solver are outside the timed region"
                        % ("DLA-34-FPN 720p" if (CHANNELS, NET_HW) == (128, (704, 1280)) else "synthetic",
                           NET_HW[0], NET_HW[1], CHANNELS, n),
            "tracks": n, "channels": CHANNELS, "rz": rz, "rx": rx, "prewarm_ms": args.prewarm_ms,
            "parallelism": "streams x%d (weights broadcast once: %d B)" % (world, bcast_bytes),
        },
        "roofline": {
            "bound": "hbm", "kernel": "sr_xcorr_fused8_kernel<30,15,2,true> (search-region ROIAlign + depthwise xcorr)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": fused_bytes,
            "avg_launch_us": xcorr_avg_s * 1e6, "event_bracket_overhead_us": bracket_us, "launches_timed": xcorr_launches, "timer_stride": TIMER_STRIDE,
            "xcorr_op": xop,
        },
        # the kernel with the largest share of GPU time: the two conv3x3 towers.  Algorithmic FLOPs are those of
        # the direct convolution the reference computes; the kernel runs it as Winograd F(2x2,3x3) on the fp32
        # matrix cores, i.e. it EXECUTES 2.25x fewer multiply-adds (reported separately, with the matrix-pipe
        # fraction they amount to).
        "roofline_tower": None if tower_launches == 0 else tower_roofline(n, tower_total_ms, tower_launches, bracket_us),
        "multi_stream": multi,
        "tracking_loop": loop_stats,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
