#!/usr/bin/env python3
"""bench.py -- frame-pairs/s of the dense Farneback hot path (BASELINE.json metric and configs).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}] [--family translated|warped]

With --gpus N > 1 and no torch.distributed environment, bench.py starts its own N ranks: a fresh child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` is spawned BEFORE this
process touches the GPU, and its exit code is returned.  Under torch.distributed.run (RANK / WORLD_SIZE set) it is a rank.

A "step" is one pass of the hot path over one batch of synthetic frame pairs, inputs resident in HBM before the timed
region, flow written to HBM, danger maps computed; with N > 1 the danger maps are all-gathered over RCCL inside the
timed region.  Configs (BASELINE.json `configs`):

  3 (default)  1920x1080, levels=5, iterations=3, 512 pairs PER GPU           -> "scaling": "weak"
  4            the same 512 pairs in total, 512/N per GPU, RCCL gather       -> "scaling": "strong"
  5            3840x2160, levels=6, iterations=5, 64 pairs in total, 64/N per GPU -> "strong"
  2            one 1920x1080 pair through the host-pointer drop-in call (ofarn_calc): latency in ms

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel's algorithmic bytes per launch / its mean launch duration, measured live with hipEvent
                pairs on the launch stream (ofarn_profile_*), against 8 TB/s; `traffic` = HBM bytes per launch from
                the separate rocprofv3 --pmc passes kept in profiles/pmc_traffic.json (a profiler cannot run inside the
                bench), `traffic_frac` = traffic / kernel time / 8 TB/s -- the real bandwidth utilisation.
  cpu_baseline  the CPU oracle (oracle/farneback_oracle.c, kind "port") timed on a bounded sample of the same workload
                on this box's host cores (rank 0, N = 1 only); `cores` = threads used.
  parity        on the CPU sample: EPE vs the oracle (OpenCV-order sums), EPE vs ground truth, and the symmetric
                difference between the danger index sets from GPU flow and from OpenCV-order flow.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

CONFIGS = {
    2: dict(w=1920, h=1080, levels=5, iterations=3, global_pairs=1, scaling="weak", key="1080p_L5_I3"),
    3: dict(w=1920, h=1080, levels=5, iterations=3, per_gpu_pairs=512, scaling="weak", key="1080p_L5_I3"),
    4: dict(w=1920, h=1080, levels=5, iterations=3, global_pairs=512, scaling="strong", key="1080p_L5_I3"),
    5: dict(w=3840, h=2160, levels=6, iterations=5, global_pairs=64, scaling="strong", key="4k_L6_I5"),
}

# Algorithmic bytes per work unit of each stage (declared inputs read once, outputs
# written once).  Units: level pixels x frames for A/B, level pixels x pairs for C/D/E.
STAGE_BYTES = {
    "polyexp": 24.0,          # 4 B in, 20 B out
    "update_matrices": 68.0,  # flow 8 + R0 20 + R1 20 in, M 20 out
    "blur_solve": 28.0,       # M 20 in, flow 8 out
    "flow_upsample": 10.0,    # 8 B out + 8 B/4 in
    "level_vpass": 4.0,       # level image out (the frame read is charged to level_hpass)
    "level_hpass": 8.0,       # row-pass intermediate out: float2 per (frame row, level column); the 1 B/px frame read adds
                              # 1/8 .. 1/2 B per unit at the levels that take this path (scale 1/16 and below)
    "flow_iter": 96.0,        # fused C + D as SURVEY 8(d) charges it: C 68 + D 28, i.e. including the 20 B written + 20 B read
                              # of M that the fused kernel never moves
}
# What the fused iteration kernel itself declares: flow 8 + R0 20 + R1 20 in, flow 8 out.  `roofline.achieved` / `frac` use
# THIS figure for a flow_iter launch (a bandwidth the memory system really has to deliver, so it cannot exceed the peak);
# the SURVEY 8(d) figure above is reported next to it as `survey_8d_achieved` / `survey_8d_frac`.
KERNEL_BYTES = dict(STAGE_BYTES, flow_iter=56.0)


def algorithmic_bytes_per_pair(w, h, plan, iterations):
    """SURVEY.md 8(d) formula: A + B + C + D + E."""
    n0 = w * h
    s = len(plan)
    sig = sum(lw * lh for lw, lh, _, _ in plan)
    ntop = plan[-1][0] * plan[-1][1]
    a = 2 * s * n0 * 1 + 2 * sig * 4
    b = 2 * sig * 24
    c = iterations * sig * 68
    d = iterations * sig * 28
    e = 8 * (sig - n0) + 8 * (sig - ntop)
    return float(a + b + c + d + e)


def self_launch(args, argv):
    """--gpus N > 1 outside torch.distributed.run: start the ranks as a fresh child.  Nothing in this process has
    touched HIP yet (torch is not even imported), and the child is a new process, not an exec of this one."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    rc = 1
    for attempt in range(3):
        # a port that is free now can be taken by the time the launcher binds it (seen once: EADDRINUSE): a child that dies within
        # seconds, before any rank has got as far as its timed region, is started again on another port
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={max(args.gpus, 1)}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
        t0 = time.time()
        # stderr is passed through AND kept: only a rendezvous port that was taken between the probe above and the launcher's bind
        # (EADDRINUSE) is worth another attempt; any other failure -- a rank that faults, a bad argument -- is returned as it is
        proc = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE, text=True, errors="replace")
        tail = []
        for line in proc.stderr:
            sys.stderr.write(line)
            tail.append(line)
            del tail[:-400]
        rc = proc.wait()
        err = "".join(tail)
        if rc == 0 or not ("EADDRINUSE" in err or "address already in use" in err.lower()):
            break
        print(f"[bench] rendezvous port {port} was taken (launcher exited with {rc} after {time.time() - t0:.0f} s); "
              f"attempt {attempt + 1} of 3", file=sys.stderr)
    return rc


def smooth_base_gpu(torch, hh, ww, seed, device, sigma=4.0):
    """Low-passed white noise, float32 [hh, ww] on the GPU."""
    r = 16
    x = torch.arange(-r, r + 1, device=device, dtype=torch.float32)
    k = torch.exp(-0.5 * (x / sigma) ** 2)
    k = (k / k.sum())
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    base = torch.randn((hh, ww), generator=g, device=device)
    # separable circular Gaussian as 2 x 33 shifted multiply-adds (plain elementwise kernels: no MIOpen convolution search, which eight
    # ranks starting at once would all run against one user database)
    for dim in (1, 0):
        acc = torch.zeros_like(base)
        for i in range(2 * r + 1):
            acc.add_(torch.roll(base, shifts=r - i, dims=dim), alpha=float(k[i]))
        base = acc
    lo, hi = base.min(), base.max()
    return (base - lo) * (255.0 / (hi - lo))


def make_frames_gpu(torch, n_pairs, seed0, device, W, H, family):
    """Synthetic pairs generated on the GPU.  Returns frames uint8[2*n_pairs, H, W] (prev0, next0, prev1, ...) and the
    ground-truth flow: shifts int[n_pairs, 2] for "translated" (`next` an integer translation of `prev`), or
    a list of float32[H, W, 2] CPU arrays for "warped" (zoom about the frame centre + rotation + sub-pixel shift, the
    FPV forward-flight field of the reference's use case; bicubic resampling)."""
    frames = torch.empty((2 * n_pairs, H, W), dtype=torch.uint8, device=device)
    if family == "translated":
        pad = 16
        shifts = np.empty((n_pairs, 2), np.int64)
        for i in range(n_pairs):
            img = torch.round(smooth_base_gpu(torch, H + 2 * pad, W + 2 * pad, seed0 + i, device)).to(torch.uint8)
            rs = np.random.default_rng(seed0 + i)
            tx, ty = (int(v) for v in rs.integers(-8, 9, size=2))
            shifts[i] = (tx, ty)
            frames[2 * i] = img[pad:pad + H, pad:pad + W]
            frames[2 * i + 1] = img[pad - ty:pad - ty + H, pad - tx:pad - tx + W]
        return frames, shifts
    gts = []
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    cx, cy = float(W // 2), float(H // 2)
    for i in range(n_pairs):
        rs = np.random.default_rng(seed0 + i)
        zoom, ang = float(rs.uniform(1.005, 1.03)), float(np.deg2rad(rs.uniform(-1.0, 1.0)))
        tx, ty = float(rs.uniform(-3, 3)), float(rs.uniform(-3, 3))
        a00, a01, a10, a11 = zoom * np.cos(ang), -zoom * np.sin(ang), zoom * np.sin(ang), zoom * np.cos(ang)
        fx = a00 * (xs - cx) + a01 * (ys - cy) + tx - (xs - cx)
        fy = a10 * (xs - cx) + a11 * (ys - cy) + ty - (ys - cy)
        pad = int(max(float(fx.abs().max()), float(fy.abs().max()))) + 12
        base = smooth_base_gpu(torch, H + 2 * pad, W + 2 * pad, seed0 + i, device)
        det = a00 * a11 - a01 * a10
        qx, qy = xs - cx - tx, ys - cy - ty
        px = cx + (a11 * qx - a01 * qy) / det + pad
        py = cy + (-a10 * qx + a00 * qy) / det + pad
        grid = torch.stack([(px + 0.5) / (W + 2 * pad) * 2 - 1, (py + 0.5) / (H + 2 * pad) * 2 - 1], -1)[None]
        nxt = torch.nn.functional.grid_sample(base[None, None], grid, mode="bicubic", padding_mode="reflection",
                                              align_corners=False)[0, 0]
        frames[2 * i] = torch.round(base[pad:pad + H, pad:pad + W]).clamp(0, 255).to(torch.uint8)
        frames[2 * i + 1] = torch.round(nxt).clamp(0, 255).to(torch.uint8)
        gts.append(torch.stack([fx, fy], -1).cpu().numpy())
    return frames, gts


def cpu_baseline(frames_np, n_sample, params, threads):
    """Times the CPU oracle on the first n_sample pairs: one pair on one thread, then OpenMP across pairs on `threads`
    threads.  Returns the cpu_baseline object and the oracle flows (OpenCV-order sums) of the sample."""
    from oracle import oracle as O
    O.build()
    fr = np.ascontiguousarray(frames_np[:2 * n_sample])
    t0 = time.perf_counter()
    ref1 = O.farneback_batch(fr[:2], 0, nthreads=1, **params)
    t1 = time.perf_counter() - t0
    cores = max(1, min(threads, O.omp_max_threads(), n_sample))
    t0 = time.perf_counter()
    ref = O.farneback_batch(fr, 0, nthreads=cores, **params)
    tn = time.perf_counter() - t0
    assert np.array_equal(ref[0], ref1[0])
    return {
        "value": round(n_sample / tn, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
        "sample": f"first {n_sample} pairs of the workload, oracle/farneback_oracle.c (OpenCV-algorithm CPU "
                  f"restatement, literal running-sum order), OpenMP across pairs on {cores} threads; single thread: "
                  f"{1 / t1:.4f} pairs/s on 1 pair",
        "single_thread_value": round(1 / t1, 4),
        "host_cpus": os.cpu_count(), "host_cpus_usable": len(os.sched_getaffinity(0)),
    }, ref


def parity_report(ofa, gpu_flow, gpu_mask, ref, gt, W, H, family):
    """EPE of the GPU flow vs the oracle's OpenCV-order flow and vs ground truth, and the symmetric difference of the
    danger index sets (GPU mask from GPU flow vs the reference's NumPy filter on the OpenCV-order flow)."""
    from oracle import oracle as O
    n = len(ref)
    e = np.linalg.norm(gpu_flow.astype(np.float64) - ref.astype(np.float64), axis=-1)
    out = {"pairs": n, "family": family,
           "epe_vs_cpu_oracle": {"mean": float(e.mean()), "p999": float(np.quantile(e, 0.999)), "max": float(e.max())}}
    inner = (slice(32, -32), slice(32, -32))
    g_gpu, g_cpu = [], []
    for i in range(n):
        g = np.asarray(gt[i], np.float64)
        g = g[None, None, :] if g.ndim == 1 else g[inner]
        g_gpu.append(np.linalg.norm(gpu_flow[i][inner] - g, axis=-1).mean())
        g_cpu.append(np.linalg.norm(ref[i][inner] - g, axis=-1).mean())
    out["epe_vs_ground_truth_interior_px"] = {"gpu_mean": float(np.mean(g_gpu)), "cpu_oracle_mean": float(np.mean(g_cpu))}
    sym = same = 0
    for i in range(n):
        m_lit, _ = O.danger_map_numpy(ref[i], W, H, 30)
        m_own, _ = O.danger_map_numpy(gpu_flow[i], W, H, 30)
        sym += int((gpu_mask[i] != m_lit).sum())
        same += int((gpu_mask[i] != m_own).sum())
    out["danger_set_symmetric_difference"] = {
        "gpu_flow_vs_opencv_order_flow": sym, "gpu_filter_vs_numpy_filter_same_flow": same,
        "grid_points": int(n * gpu_mask.shape[1]),
        "note": "index sets from the GPU's flow vs from the CPU oracle's flow in OpenCV's literal summation order "
                "(flows differ by ~1e-6 px); on the same flow the GPU filter equals the reference's NumPy lines"}
    return out


def literal_order_leg(ofa, torch, frames_dev, ref, n, W, H, params, device_index):
    """After the timed region (never `value`): the first n pairs again with "box_order" = 1 -- the box window summed in optflowgf.cpp's
    literal order (k_vsum_running + k_hsum_running_solve) -- compared with the CPU oracle in the same order: bit-exact or not, and
    what the mode costs.  The throughput kernels use restarted sums (the EPE above is between the two orders)."""
    with ofa.FarnebackEngine(W, H, n, device_index, **params) as lit:
        lit.set_option("box_order", 1)
        flow = torch.empty((n, H, W, 2), dtype=torch.float32, device=frames_dev.device)
        st = torch.cuda.current_stream().cuda_stream
        run = lambda: lit.calc_batch_device(frames_dev[:2 * n], 2 * n, W, H, ofa.PAIRS_INDEPENDENT, flow, None, None, stream=st)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        got = flow.cpu().numpy()
    diff = np.abs(got.astype(np.float64) - ref[:n].astype(np.float64))
    return {"pairs": n, "bit_exact_vs_cpu_oracle_in_opencvs_order": bool(np.array_equal(got, ref[:n])), "max_abs_diff_px": float(diff.max()),
            "pairs_per_s": round(n / dt, 2),
            "note": "ofarn_set_option \"box_order\" = 1: FarnebackUpdateFlow_Blur's own running sums (one per column down the image with "
                    "float row differences, one along each row); unfused, a verification mode"}


def opencv_column(frames_np, gpu_flow, params, n):
    """EPE and pairs/s against the REAL cv2.calcOpticalFlowFarneback, only if cv2 is importable on this box."""
    try:
        import cv2
    except Exception:
        return {"available": False}
    es = []
    t0 = time.perf_counter()
    for i in range(n):
        ref = cv2.calcOpticalFlowFarneback(frames_np[2 * i], frames_np[2 * i + 1], None, params["pyr_scale"], params["levels"],
                                           params["winsize"], params["iterations"], params["poly_n"], params["poly_sigma"],
                                           params["flags"])
        es.append(np.linalg.norm(gpu_flow[i].astype(np.float64) - ref, axis=-1))
    dt = time.perf_counter() - t0
    e = np.concatenate([x.ravel() for x in es])
    return {"available": True, "version": cv2.__version__, "pairs": n, "pairs_per_s": round(n / dt, 4),
            "threads": cv2.getNumThreads(), "epe_mean": float(e.mean()), "epe_p999": float(np.quantile(e, 0.999)),
            "epe_max": float(e.max())}


def load_traffic(key):
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        j = json.load(open(tj))
    except Exception:
        return {}
    return j.get(key) or {}


def bench_latency(args, cfg, params):
    """Config 2: one 1080p pair, host pointers in and out (the drop-in call), latency."""
    import hackathonopticalflow_amd as ofa
    from hackathonopticalflow_amd.synth import translated_pair, warped_pair
    W, H = cfg["w"], cfg["h"]
    if args.family == "warped":
        prev, nxt, gt, _ = warped_pair(H, W, 2001, zoom=1.02, angle_deg=0.5)
    else:
        prev, nxt, shift = translated_pair(H, W, 2001)
        gt = np.float32(shift)
    eng = ofa.FarnebackEngine(W, H, 1, 0, **params)
    flow = eng.calc(prev, nxt)
    for _ in range(args.warmup):
        eng.calc(prev, nxt, flow)
    ts, dev = [], []
    t_all = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        flow = eng.calc(prev, nxt, flow)
        ts.append((time.perf_counter() - t0) * 1e3)
        dev.append(eng.last_device_ms)
    t_all = time.perf_counter() - t_all
    plan = ofa.level_plan(W, H, **params)
    alg = algorithmic_bytes_per_pair(W, H, plan, params["iterations"])
    dms = float(np.median(dev))
    out = {
        "metric": "latency of one 1920x1080 frame pair (5-level, 3-iter), host buffers in/out", "value": round(float(np.median(ts)), 4),
        "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t_all / args.steps * 1e3, 4),
        "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config2: one 1920x1080 {args.family} pair through ofarn_calc (host pointers: 4 MB in, 16.6 MB out over "
                               f"PCIe inside the timed call), levels=5 iterations=3 winsize=15 poly_n=5", "global_pairs": 1},
        "device_ms": round(dms, 4), "wall_ms_min": round(min(ts), 4), "pairs_per_s_wall": round(1e3 / float(np.median(ts)), 1),
        "roofline": {"bound": "latency (30 dependent launches; all but the finest level are grids of a few blocks)",
                     "achieved": round(alg / (dms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                     "algorithmic_bytes_per_pair": alg},
    }
    if args.cpu_sample > 0:
        from oracle import oracle as O
        O.build()
        t0 = time.perf_counter()
        ref = O.farneback(prev, nxt, **params)
        tc = time.perf_counter() - t0
        e = np.linalg.norm(flow.astype(np.float64) - ref, axis=-1)
        g = gt[None, None] if np.ndim(gt) == 1 else gt[32:-32, 32:-32]
        out["cpu_baseline"] = {"value": round(tc * 1e3, 2), "unit": "ms", "cores": 1, "kind": "port",
                               "sample": "the same pair, oracle/farneback_oracle.c on one thread (OpenCV's PolyExp / UpdateMatrices / "
                                         "UpdateFlow loops are single-threaded scalar code)"}
        out["mean_epe_vs_cpu_oracle_px"] = float(e.mean())
        out["epe_vs_cpu_oracle"] = {"mean": float(e.mean()), "p999": float(np.quantile(e, 0.999)), "max": float(e.max())}
        out["epe_vs_ground_truth_interior_px"] = float(np.linalg.norm(flow[32:-32, 32:-32] - g, axis=-1).mean())
        out["opencv"] = opencv_column(np.stack([prev, nxt]), flow[None], params, 1)
    print(json.dumps(out))


def bench_stream(args, cfg, params):
    """Config 2 as the reference actually runs it (DenseOF.py:491-525): a frame LOOP, one new 1080p frame per turn, the
    previous one held on the device (ofarn_stream_next through FlowStream: pinned buffers, one upload, stages A + B once).
    One step = one turn = one pair.  Synthetic video: K distinct frames cycled."""
    import hackathonopticalflow_amd as ofa
    from hackathonopticalflow_amd.synth import translated_pair, warped_pair
    W, H = cfg["w"], cfg["h"]
    nuniq = 6
    frames = []
    for i in range(nuniq // 2):
        if args.family == "warped":
            a, b, _, _ = warped_pair(H, W, 2001 + i, zoom=1.02, angle_deg=0.5)
        else:
            a, b, _ = translated_pair(H, W, 2001 + i)
        frames += [a, b]
    pinned = [ofa.pinned_empty((H, W), np.uint8) for _ in range(nuniq)] if args.pinned_frames else None
    if pinned:
        for p_, f_ in zip(pinned, frames):
            p_[...] = f_
        frames_in = pinned
    else:
        frames_in = frames
    st = ofa.FlowStream(**params)
    st.next(frames_in[0])
    for i in range(args.warmup):
        st.next(frames_in[(i + 1) % nuniq])
    st.reset()
    st.next(frames_in[0])
    ts, dev = [], []
    t_all = time.perf_counter()
    flow = None
    for i in range(args.steps):
        t0 = time.perf_counter()
        flow = st.next(frames_in[(i + 1) % nuniq])
        ts.append((time.perf_counter() - t0) * 1e3)
        dev.append(st.last_device_ms)
    t_all = time.perf_counter() - t_all
    # the same loop pipelined (FlowStream(pipelined=True): next(frame t) returns the flow of turn t-1 while turn t runs)
    stp = ofa.FlowStream(pipelined=True, **params)
    wp = max(args.warmup + 2, 40)      # the first few dozen turns of a process run slower (pinned buffers first touched, clocks): steady state
    for i in range(wp):
        stp.next(frames_in[i % nuniq])
    t_p = time.perf_counter()
    for i in range(args.steps):
        stp.next(frames_in[(i + wp) % nuniq])
    stp.flush()
    t_p = (time.perf_counter() - t_p) / args.steps * 1e3
    stp.close()
    # the same loop returning what the reference DRAWS from the flow (danger map + draw_flow's arrow lines, optionally draw_hsv's
    # rainbow) instead of the flow field, which stays in HBM: ofarn_stream_next_view
    view = {}
    with ofa.FlowStream(**params) as stv:
        for rb, key in ((False, "view_ms_per_frame"), (True, "view_rainbow_ms_per_frame")):
            stv.reset()
            for i in range(12):
                stv.next_view(frames_in[i % nuniq], rainbow=rb)
            tv, dv = [], []
            for i in range(args.steps):
                t0 = time.perf_counter()
                stv.next_view(frames_in[(i + 12) % nuniq], rainbow=rb)
                tv.append((time.perf_counter() - t0) * 1e3)
                dv.append(stv.last_device_ms)
            view[key] = round(float(np.median(tv)), 4)
            view[key.replace("_ms_per_frame", "_device_ms")] = round(float(np.median(dv)), 4)
    # the loop through the drop-in itself, as the reference writes it (DenseOF.py:519-525): a NEW gray array per frame, the result
    # rebound every turn
    def dropin_loop(n):
        prev, total = np.array(frames_in[0]), 0.0
        for i in range(1, n + 1):
            gray = np.array(frames_in[i % nuniq])              # what cap.read() + cvtColor hand over: a fresh array
            t0 = time.perf_counter()
            _flow = ofa.calculate_optical_flow(prev, gray, **params)
            total += time.perf_counter() - t0
            prev = gray
        return total / n * 1e3
    dropin_loop(max(args.warmup, 12))
    dropin_ms = dropin_loop(args.steps)
    last_pair = (frames[(args.steps - 1) % nuniq], frames[args.steps % nuniq])
    plan = ofa.level_plan(W, H, **params)
    alg = algorithmic_bytes_per_pair(W, H, plan, params["iterations"])
    dms = float(np.median(dev))
    out = {
        "metric": "latency per frame of a 1920x1080 frame LOOP (5-level, 3-iter): one new frame in, flow of (previous, new) out, host buffers",
        "value": round(float(np.median(ts)), 4), "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(t_all / args.steps * 1e3, 4), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config2 --stream: 1920x1080 {args.family} frames through FlowStream / ofarn_stream_next (2 MB in, 16.6 MB out "
                               f"over PCIe inside the timed call; flow into the stream's pinned buffers"
                               f"{', frames from pinned memory' if pinned else ', frames from pageable NumPy arrays'}), "
                               f"levels=5 iterations=3 winsize=15 poly_n=5", "global_pairs": 1},
        "device_ms": round(dms, 4), "wall_ms": round(float(np.median(ts)), 4), "wall_ms_min": round(min(ts), 4),
        "pairs_per_s_wall": round(1e3 / float(np.median(ts)), 1),
        "zero_copy": os.environ.get("OFARN_STREAM_ZERO_COPY", "1") != "0",
        "pipelined_ms_per_frame": round(t_p, 4), "pipelined_pairs_per_s_wall": round(1e3 / t_p, 1),
        "dropin_loop_ms_per_frame": round(dropin_ms, 4),     # flow = calculate_optical_flow(prev_gray, gray); prev_gray = gray
        "view": dict(view, note="synchronous turn returning the danger map + arrow lines (and, second pair of figures, the rainbow image) "
                                "computed from the flow on the device; the float32 flow field stays in HBM"),
        "roofline": {"bound": "latency (dependent launches on grids of a few blocks) + PCIe (16.6 MB of flow per frame)",
                     "achieved": round(alg / (dms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                     "algorithmic_bytes_per_pair": alg},
    }
    if args.cpu_sample > 0:
        from oracle import oracle as O
        O.build()
        t0 = time.perf_counter()
        ref = O.farneback(last_pair[0], last_pair[1], **params)
        tc = time.perf_counter() - t0
        e = np.linalg.norm(flow.astype(np.float64) - ref, axis=-1)
        out["cpu_baseline"] = {"value": round(tc * 1e3, 2), "unit": "ms", "cores": 1, "kind": "port",
                               "sample": "the last pair of the loop, oracle/farneback_oracle.c on one thread"}
        out["epe_vs_cpu_oracle"] = {"mean": float(e.mean()), "p999": float(np.quantile(e, 0.999)), "max": float(e.max())}
        out["mean_epe_vs_cpu_oracle_px"] = float(e.mean())
        out["bit_exact_vs_oracle_device_order"] = bool(np.array_equal(
            flow, O.farneback(last_pair[0], last_pair[1], box_mode=O.BOX_BLOCKED, **params)))
    st.close()
    print(json.dumps(out))


def inproc_shards(cfg, n_gpus, batch=None):
    """Pairs per device for the in-process multi-GPU bench: (global_pairs, [(start, count)] per device).  Config 3 is weak scaling
    (a fixed batch per GPU), configs 4 and 5 shard a fixed global batch -- exactly ofarn_shard_pairs, which is what ofarn_multi_*
    uses.  Pure arithmetic (tests/test_distributed_cpu.py)."""
    from hackathonopticalflow_amd import distributed as D
    if n_gpus < 1:
        raise ValueError("n_gpus must be >= 1")
    global_pairs = (batch or cfg["per_gpu_pairs"]) * n_gpus if "per_gpu_pairs" in cfg else (batch or cfg["global_pairs"])
    shards = [D.shard_pairs(global_pairs, g, n_gpus) for g in range(n_gpus)]
    if min(c for _, c in shards) < 1:
        raise ValueError(f"{global_pairs} pairs cannot be sharded over {n_gpus} devices")
    return global_pairs, shards


def roofline_object(prof, steps, value_per_gpu, W, H, plan, iterations, traffic_tab):
    """The `roofline` object of the JSON line from one context's per-kernel records (shared by both multi-GPU forms)."""
    dom = max(prof, key=lambda r: r["ms"])
    per_launch_s = dom["ms"] / dom["launches"] / 1e3
    bytes_launch = KERNEL_BYTES.get(dom["stage"], 0.0) * dom["units"] / dom["launches"]
    ach = bytes_launch / per_launch_s / 1e9
    survey_launch = STAGE_BYTES.get(dom["stage"], 0.0) * dom["units"] / dom["launches"]
    traffic = None
    per_unit = (traffic_tab.get(dom["stage"]) or {}).get("hbm_bytes_per_unit") if dom["level"] == 0 else None
    if per_unit:
        traffic = per_unit * dom["units"] / dom["launches"]
    total_ms = sum(r["ms"] for r in prof)
    real_pair = traffic_tab.get("pipeline_hbm_bytes_per_pair")
    alg_pair = algorithmic_bytes_per_pair(W, H, plan, iterations)
    agg = {}
    for r in prof:
        agg[r["stage"]] = agg.get(r["stage"], 0.0) + r["ms"] / steps
    return {
        "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
        "traffic_frac": round(traffic / per_launch_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
        "kernel": f"{dom['stage']}@level{dom['level']}",
        "kernel_avg_ms": round(per_launch_s * 1e3, 4), "kernel_launches": dom["launches"],
        "kernel_share_of_device_time": round(dom["ms"] / total_ms, 4),
        "algorithmic_bytes_per_launch": bytes_launch,
        "algorithmic_bytes_per_unit": KERNEL_BYTES.get(dom["stage"]),
        "survey_8d_bytes_per_launch": survey_launch,
        "survey_8d_achieved": round(survey_launch / per_launch_s / 1e9, 1),
        "survey_8d_frac": round(survey_launch / per_launch_s / 1e9 / HBM_PEAK_GBS, 4),
        "survey_8d_note": "work rate, not a bandwidth: SURVEY 8(d) prices the launch at 96 B/px including 40 B/px of M traffic "
                          "that the fused kernel never moves, so it can exceed 1; `frac` (56 B/px) and `traffic_frac` (PMC) are the bandwidths",
        "pipeline": {
            "algorithmic_bytes_per_pair": alg_pair,
            "achieved": round(alg_pair * value_per_gpu / 1e9, 1),
            "frac": round(alg_pair * value_per_gpu / 1e9 / HBM_PEAK_GBS, 4),
            "real_bytes_per_pair": real_pair,
            "real_frac": round(real_pair * value_per_gpu / 1e9 / HBM_PEAK_GBS, 4) if real_pair else None,
        },
        "stages_ms_per_step": {k: round(val, 3) for k, val in agg.items()},
    }


def bench_inproc(args, cfg, params):
    """`--multi inproc`: SURVEY 8(e) as written.  ONE process drives N GPUs through the C-ABI's ofarn_multi_* (MultiGpuEngine):
    device-resident shards in, flow to each device's HBM, the danger maps of ALL pairs all-gathered onto every device by one
    in-place ncclAllGather group per step -- inside the timed region.  Same JSON schema as the one-process-per-GPU form."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: hackathonopticalflow_amd has no CPU path")
    import hackathonopticalflow_amd as ofa
    N = max(args.gpus, 1)
    ndev = torch.cuda.device_count()
    if N > ndev:
        raise SystemExit(f"--gpus {N} but {ndev} GPU(s) visible (one context per GPU)")
    W, H = cfg["w"], cfg["h"]
    try:
        global_pairs, shards = inproc_shards(cfg, N, args.batch)
    except ValueError as e:
        raise SystemExit(str(e))
    cap = max(c for _, c in shards)
    wave = min(args.wave, cap)
    eng = ofa.MultiGpuEngine(list(range(N)), W, H, wave, **params)
    P = len(ofa.grid_points(W, H, 30))
    plan = ofa.level_plan(W, H, **params)
    uniq = min(args.unique, global_pairs)
    frames, flows, masks, vs = [], [], [], []
    fr_u0 = gt_u0 = None
    for g, (start, cnt) in enumerate(shards):
        device = torch.device("cuda", g)
        with torch.cuda.device(g):
            fr_u, gt_u = make_frames_gpu(torch, uniq, 3000, device, W, H, args.family)     # the same distinct pairs on every device
            f = torch.empty((2 * cnt, H, W), dtype=torch.uint8, device=device)
            for i in range(cnt):
                j = (start + i) % uniq
                f[2 * i:2 * i + 2] = fr_u[2 * j:2 * j + 2]
            frames.append(f)
            flows.append(torch.empty((cnt, H, W, 2), dtype=torch.float32, device=device))
            masks.append(torch.zeros((global_pairs, P), dtype=torch.uint8, device=device))
            vs.append(torch.zeros((global_pairs, P), dtype=torch.uint8, device=device))
            if g == 0:
                fr_u0, gt_u0 = fr_u, gt_u
            else:
                del fr_u, gt_u
            torch.cuda.synchronize(g)

    def step():
        eng.calc_batch_device(frames, global_pairs, W, H, ofa.PAIRS_INDEPENDENT, flows, masks, vs)

    def fence():
        eng.synchronize()

    def timed_steps(k):
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        fence()
        return time.perf_counter() - t0

    for _ in range(args.warmup):
        step()
    fence()
    calls0 = eng.info()["allgather_calls"]
    if not args.no_profile:
        eng.rank_profile_enable(0, True)
    elapsed = timed_steps(args.steps)
    prof = [] if args.no_profile else eng.rank_profile_read(0)
    if not args.no_profile:
        eng.rank_profile_enable(0, False)
    calls = eng.info()["allgather_calls"] - calls0
    # the gather really gathered: device g's own rows of the gathered arrays equal what every OTHER device holds for them, and a
    # single-context run of device 0's shard reproduces its rows
    gather_ok = True
    ref_m, ref_v = masks[0].cpu(), vs[0].cpu()
    for g in range(1, N):
        gather_ok = gather_ok and bool((masks[g].cpu() == ref_m).all()) and bool((vs[g].cpu() == ref_v).all())
    with ofa.FarnebackEngine(W, H, min(wave, 8), 0, **params) as single, torch.cuda.device(0):
        nchk = min(shards[0][1], 8)
        f1 = torch.empty((nchk, H, W, 2), dtype=torch.float32, device="cuda:0")
        m1 = torch.zeros((nchk, P), dtype=torch.uint8, device="cuda:0")
        v1 = torch.zeros_like(m1)
        single.calc_batch_device(frames[0][:2 * nchk], 2 * nchk, W, H, ofa.PAIRS_INDEPENDENT, f1, m1, v1,
                                 stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(0)
        gather_ok = gather_ok and bool((m1 == masks[0][:nchk]).all()) and bool((v1 == vs[0][:nchk]).all()) \
            and bool((f1 == flows[0][:nchk]).all())
        del f1, m1, v1
    value = global_pairs * args.steps / elapsed
    res = "1920x1080" if W == 1920 else f"{W}x{H}"
    if args.config == 3:
        workload = f"config3: {res} batch={cap} {args.family} smooth-noise pairs per GPU"
    else:
        workload = f"config{args.config}: {res} batch={global_pairs} {args.family} smooth-noise pairs in total, {cap} per GPU (sharded)"
    out = {
        "metric": f"frame-pairs/s @{'1080p' if W == 1920 else '4K'} ({cfg['levels']}-level, {cfg['iterations']}-iter)",
        "value": round(value, 2), "unit": "pairs/s",
        "n_gpus": N, "ranks": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": cfg["scaling"],
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload + f" ({uniq} distinct, tiled), levels={cfg['levels']} iterations={cfg['iterations']} winsize=15 "
                               f"poly_n=5, f32 with OpenCV's f64 accumulators, flow + danger maps to HBM, the danger maps of all pairs "
                               f"all-gathered onto every device (in-place ncclAllGather group) inside the timed region",
                   "pairs_per_gpu": cap, "global_pairs": global_pairs, "wave": wave,
                   "workspace_bytes_per_gpu": eng.rank_workspace_bytes(0),
                   "parallelism": f"{N} devices in one process, ncclCommInitAll"},
        "roofline": roofline_object(prof, args.steps, value / N, W, H, plan, params["iterations"], load_traffic(cfg["key"])) if prof else None,
        "gathered_danger_maps_checked": gather_ok,
        "collective": {"backend": "rccl (ncclCommInitAll, in-place ncclAllGather, one group per batch)", "world": N, "calls": int(calls),
                       "calls_per_step": calls / max(args.steps, 1), "bytes_per_rank": int(2 * cap * P),
                       "rccl_version": eng.info()["rccl_version"]},
    }
    if N == 1 and args.cpu_sample > 0:
        ns = min(args.cpu_sample, uniq, shards[0][1])
        npar = min(16, ns)
        fr_np = fr_u0[:2 * ns].cpu().numpy()
        cb, ref = cpu_baseline(fr_np, ns, params, args.cpu_threads)
        out["cpu_baseline"] = cb
        par = parity_report(ofa, flows[0][:npar].cpu().numpy(), masks[0][:npar].cpu().numpy(), ref[:npar],
                            [gt_u0[i] for i in range(npar)], W, H, args.family)
        out["parity"] = par
        out["mean_epe_vs_cpu_oracle_px"] = par["epe_vs_cpu_oracle"]["mean"]
        out[f"speedup_vs_cpu_{cb['cores']}_threads"] = round(value / cb["value"], 1)
        out["speedup_vs_cpu_1_thread"] = round(value / cb["single_thread_value"], 1)
    print(json.dumps(out))
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json config number (see the module docstring)")
    ap.add_argument("--batch", type=int, default=None, help="override the config's pair count (per GPU for config 3, in total for 4 and 5)")
    ap.add_argument("--wave", type=int, default=512,
                    help="pairs resident per wave (ofarn max_batch); 512 pairs of 1080p = 63 GB of workspace, one wave per step "
                         "(+1.5 %% over two waves of 256: half the launches and kernel tails)")
    ap.add_argument("--unique", type=int, default=64, help="distinct synthetic pairs generated, then tiled")
    ap.add_argument("--family", default="translated", choices=["translated", "warped"],
                    help="synthetic input: integer translations (SURVEY 8(d)) or the FPV-like zoom + rotation + sub-pixel shift field")
    ap.add_argument("--cpu-sample", type=int, default=64, help="pairs timed on the CPU oracle (0 = skip); parity is reported on the first 16")
    ap.add_argument("--cpu-threads", type=int, default=16, help="OpenMP threads of the CPU baseline (the GPU box's CPU share per GPU is 16)")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel hipEvent timing")
    ap.add_argument("--prof-table", action="store_true", help="print per-(stage, level) timing rows to stderr")
    ap.add_argument("--no-two-stream", action="store_true",
                    help="skip the informational second measurement with per-kernel timing off (two internal streams)")
    ap.add_argument("--no-family-check", action="store_true",
                    help="skip the informational third measurement on the other input family (data-independence check)")
    ap.add_argument("--stream", action="store_true", help="config 2 only: the frame loop (one new frame per step) instead of one pair per step")
    ap.add_argument("--pinned-frames", action="store_true", help="--stream: input frames in page-locked memory too")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the distributed leg at ANY world size, 1 included: bench.py starts its rank(s) through "
                         "torch.distributed.run, initialises the process group and all-gathers the danger maps inside the timed "
                         "region.  `--gpus 1 --backend nccl --force-dist` is how RCCL's all_gather_into_tensor is exercised on a one-GPU box")
    ap.add_argument("--multi", default="ranks", choices=["ranks", "inproc"],
                    help="how N GPUs are driven: 'ranks' = one process per GPU (torch.distributed / RCCL, what the driver launches); "
                         "'inproc' = SURVEY 8(e) as written: ONE process, ofarn_multi_* (a context + persistent host thread + stream "
                         "per device, ncclCommInitAll, one in-place ncclAllGather group per batch)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (gloo: rehearsal of the multi-rank path with "
                         "several ranks sharing one GPU; the gather then goes through host memory)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.steps is None:
        args.steps = 20 if args.config == 2 else 5
    if args.warmup is None:
        args.warmup = (40 if args.stream else 5) if args.config == 2 else 2
    if args.steps < 1 or args.warmup < 0 or args.wave < 1 or (args.batch is not None and args.batch < 1):
        raise SystemExit("--steps, --batch and --wave must be >= 1, --warmup >= 0")

    if args.multi == "inproc":
        if args.config == 2:
            raise SystemExit("--multi inproc drives batches (configs 3, 4, 5)")
        if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("--multi inproc is ONE process for all GPUs: start it without torch.distributed.run")
        params = dict(pyr_scale=0.5, levels=cfg["levels"], winsize=15, iterations=cfg["iterations"], poly_n=5, poly_sigma=1.2, flags=0)
        return bench_inproc(args, cfg, params)

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if (args.gpus > 1 or args.force_dist) and "RANK" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))      # fresh child; this process never initialises HIP
    if env_world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env_world}")

    params = dict(pyr_scale=0.5, levels=cfg["levels"], winsize=15, iterations=cfg["iterations"], poly_n=5, poly_sigma=1.2, flags=0)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: hackathonopticalflow_amd has no CPU path")
    if args.config == 2:
        if args.gpus != 1:
            raise SystemExit("config 2 is a single pair on a single GPU")
        return bench_stream(args, cfg, params) if args.stream else bench_latency(args, cfg, params)

    import hackathonopticalflow_amd as ofa
    from hackathonopticalflow_amd import distributed as D

    rank, local_rank, world = D.env_rank_world()
    W, H = cfg["w"], cfg["h"]
    ndev = torch.cuda.device_count()
    share = os.environ.get("OFARN_BENCH_SHARE_GPU") == "1" and args.backend == "gloo"
    if world > ndev and not share:
        raise SystemExit(f"{world} ranks but {ndev} GPU(s) visible: one rank per GPU.  (Rehearsing the multi-rank path with several "
                         f"ranks on one GPU needs OFARN_BENCH_SHARE_GPU=1 and --backend gloo; the line then reports the number of "
                         f"distinct devices as n_gpus.)")
    dev_index = local_rank % ndev                           # only ever folds under the gloo rehearsal above
    n_devices = min(world, ndev)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = D.init_process_group(args.backend, force=args.force_dist) if (world > 1 or args.force_dist) else None

    # pairs: config 3 = a fixed batch per GPU (weak); configs 4, 5 = a fixed global batch sharded over the ranks (strong)
    if "per_gpu_pairs" in cfg:
        B = args.batch or cfg["per_gpu_pairs"]
        global_pairs, pair_start = B * world, rank * B
    else:
        global_pairs = args.batch or cfg["global_pairs"]
        pair_start, B = D.shard_pairs(global_pairs, rank, world)
        if B < 1:
            raise SystemExit(f"{global_pairs} pairs cannot be sharded over {world} ranks")
    wave = min(args.wave, B)
    eng = ofa.FarnebackEngine(W, H, wave, dev_index, **params)
    P = len(ofa.grid_points(W, H, 30))
    plan = ofa.level_plan(W, H, **params)

    # synthetic input, resident in HBM before anything is timed; global pair g uses distinct pair g % unique
    uniq = min(args.unique, global_pairs)
    fr_u, gt_u = make_frames_gpu(torch, uniq, 3000, device, W, H, args.family)

    def tile(fr):
        frames = torch.empty((2 * B, H, W), dtype=torch.uint8, device=device)
        for i in range(B):
            j = (pair_start + i) % uniq
            frames[2 * i:2 * i + 2] = fr[2 * j:2 * j + 2]
        return frames

    frames = tile(fr_u)
    flow = torch.empty((B, H, W, 2), dtype=torch.float32, device=device)
    mask = torch.zeros((B, P), dtype=torch.uint8, device=device)
    v = torch.zeros((B, P), dtype=torch.uint8, device=device)
    # a stream of its own (non-blocking), not torch's legacy null stream: launches on the null stream carry its implicit
    # synchronisation, which a small latency-bound batch notices (2 pairs of 4K: 4.16 vs 3.08 ms per step, round 4); the
    # collective is issued under the same stream context, so it is ordered behind the kernels without a device-wide fence
    tstream = torch.cuda.Stream(device=device)
    stream = tstream.cuda_stream
    gathered = [None]
    # the one collective of the path, buffers allocated once (device tensors under nccl = RCCL; host tensors under gloo)
    gather = D.DangerGather(global_pairs, P, device if args.backend == "nccl" else "cpu", dist) if dist is not None else None

    def step():
        with torch.cuda.stream(tstream):
            eng.calc_batch_device(frames, 2 * B, W, H, ofa.PAIRS_INDEPENDENT, flow, mask, v, stream=stream)
            if gather is not None:
                gathered[0] = gather(mask.cpu(), v.cpu()) if args.backend == "gloo" else gather(mask, v)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_steps(k):
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    torch.cuda.synchronize()           # the inputs were made on torch's default stream
    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_profile:
        eng.profile_enable(True)
    elapsed = timed_steps(args.steps)
    prof = [] if args.no_profile else eng.profile_read()
    eng.profile_enable(False)

    # the gather really gathered: every rank finds its own shard at its place in the global arrays
    gather_ok = None
    if dist is not None:
        gm, gv = gathered[0]
        ok = (gm.shape == (global_pairs, P) and bool((gm[pair_start:pair_start + B].to(mask.device) == mask).all())
              and bool((gv[pair_start:pair_start + B].to(v.device) == v).all()))
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        gather_ok = bool(t.item())

    # Informational second measurement (never `value`): the same K steps with per-kernel timing off.  The library then
    # alternates the waves of a batch over two internal streams, so the VALU-bound stages of one wave overlap the HBM-bound
    # iterations of the other; per-kernel durations lose their meaning there, which is why the timed region above keeps
    # everything on one stream.
    overlapped = None
    # (gated on a quantity every rank agrees on: the leg contains collectives)
    if not args.no_profile and not args.no_two_stream and global_pairs // world >= 64:
        eng_main = eng
        if wave >= B:       # the timed region ran the batch as ONE wave: a second context with waves of B/2 for this leg
            eng = ofa.FarnebackEngine(W, H, (B + 1) // 2, dev_index, **params)
        step()
        overlapped = global_pairs * args.steps / timed_steps(args.steps)
        if eng is not eng_main:
            eng.close()
            eng = eng_main

    # Informational third measurement (never `value`): the same K steps on the OTHER input family.  The one data-dependent
    # part of the pipeline is the bilinear gather of FarnebackUpdateMatrices (address pattern and its out-of-image branch).
    other = None
    if not args.no_family_check and world == 1:
        other_family = "warped" if args.family == "translated" else "translated"
        u2 = min(8, uniq)
        fr_o, _ = make_frames_gpu(torch, u2, 7000, device, W, H, other_family)
        keep = frames
        frames = torch.empty_like(keep)
        for i in range(B):
            frames[2 * i:2 * i + 2] = fr_o[2 * (i % u2):2 * (i % u2) + 2]
        torch.cuda.synchronize()
        step()
        if not args.no_profile:
            eng.profile_enable(True)
        e3 = timed_steps(args.steps)
        prof_o = [] if args.no_profile else eng.profile_read()
        eng.profile_enable(False)
        other = {"family": other_family, "pairs_per_s": round(global_pairs * args.steps / e3, 2)}
        fi = [r for r in prof_o if r["stage"] == "flow_iter" and r["level"] == 0]
        if fi:
            other["flow_iter_level0_avg_ms"] = round(fi[0]["ms"] / fi[0]["launches"], 4)
        frames = keep
        del fr_o
        step()          # leave flow / mask of the primary family in the buffers for the parity report
        fence()

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.prof_table:
        for r in sorted(prof, key=lambda r: -r["ms"]):
            per = r["ms"] / r["launches"]
            gbs = KERNEL_BYTES.get(r["stage"], 0.0) * r["units"] / r["launches"] / (per * 1e-3) / 1e9
            print(f"  {r['stage']:16s} L{r['level']} launches={r['launches']:4d} total={r['ms']:9.3f} ms "
                  f"avg={per:8.4f} ms  alg={gbs:8.1f} GB/s", file=sys.stderr)
    pairs_total = global_pairs * args.steps
    value = pairs_total / elapsed
    traffic_tab = load_traffic(cfg["key"])

    # roofline: the dominant kernel's own bytes per launch / its mean launch duration (hipEvent pairs on the launch stream); HBM
    # bytes from the PMC counters (separate rocprofv3 --pmc passes, profiles/pmc_traffic.json) scaled to this run's units per launch
    roofline = roofline_object(prof, args.steps, value / world, W, H, plan, params["iterations"], traffic_tab) if prof else None

    res = "1920x1080" if W == 1920 else f"{W}x{H}"
    if args.config == 3:
        workload = f"config3: {res} batch={B} {args.family} smooth-noise pairs per GPU"
    else:
        workload = f"config{args.config}: {res} batch={global_pairs} {args.family} smooth-noise pairs in total, {B} per GPU (sharded)"
    out = {
        "metric": f"frame-pairs/s @{'1080p' if W == 1920 else '4K'} ({cfg['levels']}-level, {cfg['iterations']}-iter)",
        "value": round(value, 2), "unit": "pairs/s",
        "n_gpus": n_devices, "ranks": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": cfg["scaling"],
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload + f" ({uniq} distinct, tiled), levels={cfg['levels']} iterations={cfg['iterations']} winsize=15 "
                               f"poly_n=5, f32 with OpenCV's f64 accumulators, flow + danger maps to HBM"
                               + (f", {'RCCL' if args.backend == 'nccl' else 'gloo'} all_gather_into_tensor of the danger maps "
                                  f"inside the timed region ({world} rank(s))" if dist is not None else ""),
                   "pairs_per_gpu": B, "global_pairs": global_pairs, "wave": wave,
                   "workspace_bytes_per_gpu": eng.workspace_bytes,
                   "parallelism": f"pairs sharded over {world} rank(s) on {n_devices} GPU(s)"
                                  + (" [ranks share a GPU: gloo rehearsal]" if world > n_devices else "")},
        "roofline": roofline,
    }
    if gather_ok is not None:
        out["gathered_danger_maps_checked"] = gather_ok
        out["collective"] = {"backend": args.backend, "world": world, "calls": gather.calls,
                             "bytes_per_rank": int(gather.send.numel()), "device": str(gather.send.device)}
    if overlapped is not None:
        out["two_stream_pairs_per_s"] = round(overlapped, 2)   # per-kernel timing off: waves overlap on two streams (informational)
    if other is not None:
        out["other_family"] = other
    if world == 1 and args.cpu_sample > 0:
        ns = min(args.cpu_sample, uniq, B)
        npar = min(16, ns)
        fr_np = fr_u[:2 * ns].cpu().numpy()
        cb, ref = cpu_baseline(fr_np, ns, params, args.cpu_threads)
        out["cpu_baseline"] = cb
        par = parity_report(ofa, flow[:npar].cpu().numpy(), mask[:npar].cpu().numpy(), ref[:npar],
                            [gt_u[i] for i in range(npar)], W, H, args.family)
        par["opencv"] = opencv_column(fr_np, flow[:min(npar, 4)].cpu().numpy(), params, min(npar, 4))
        par["literal_order_mode"] = literal_order_leg(ofa, torch, fr_u, ref, min(npar, 8), W, H, params, dev_index)
        out["parity"] = par
        out["mean_epe_vs_cpu_oracle_px"] = par["epe_vs_cpu_oracle"]["mean"]
        out[f"speedup_vs_cpu_{cb['cores']}_threads"] = round(value / cb["value"], 1)
        out["speedup_vs_cpu_1_thread"] = round(value / cb["single_thread_value"], 1)
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
