#!/usr/bin/env python3
"""bench.py -- frame-pairs/s of the dense Farneback hot path at 1920x1080 (levels=5, iterations=3).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--wave V]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of B synthetic translated-noise frame pairs
per GPU (BASELINE config 3: B = 512 at 1920x1080, inputs resident in HBM before the timed region,
flow written to HBM, danger maps computed; with N > 1 ranks the danger maps are all-gathered over
RCCL inside the timed region).  Weak scaling: every rank processes its own B pairs.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel's algorithmic bytes per launch / its mean launch duration, measured
                live with hipEvent pairs on the launch stream (ofarn_profile_*), against 8 TB/s.
  cpu_baseline  the CPU oracle (oracle/farneback_oracle.c, kind "port") timed on a bounded sample
                of the same workload on this box's host cores (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 1920, 1080
PARAMS = dict(pyr_scale=0.5, levels=5, winsize=15, iterations=3, poly_n=5, poly_sigma=1.2, flags=0)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# Algorithmic bytes per work unit of each stage (declared inputs read once, outputs
# written once).  Units: level pixels x frames for A/B, level pixels x pairs for C/D/E.
STAGE_BYTES = {
    "polyexp": 24.0,          # 4 B in, 20 B out
    "update_matrices": 68.0,  # flow 8 + R0 20 + R1 20 in, M 20 out
    "blur_solve": 28.0,       # M 20 in, flow 8 out
    "flow_upsample": 10.0,    # 8 B out + 8 B/4 in
    "level_vpass": 4.0,       # level image out (the frame read is charged to level_hpass)
    "flow_iter": 96.0,        # fused C + D: 68 + 28 (M no longer reaches HBM; SURVEY 8(d) keeps the figure)
}


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


def make_frames_gpu(torch, n_pairs, seed0, device):
    """Translated smooth-noise pairs generated on the GPU (SURVEY 8(d) construction: low-passed
    white noise, sigma 4 px, quantised to uint8, `next` an integer translation of `prev`).
    Returns frames uint8[2*n_pairs, H, W] (prev0, next0, prev1, ...) and shifts int[n_pairs, 2]."""
    pad = 16
    r = 16
    x = torch.arange(-r, r + 1, device=device, dtype=torch.float32)
    k = torch.exp(-0.5 * (x / 4.0) ** 2)
    k = (k / k.sum())
    frames = torch.empty((2 * n_pairs, H, W), dtype=torch.uint8, device=device)
    shifts = np.empty((n_pairs, 2), np.int64)
    g = torch.Generator(device=device)
    for i in range(n_pairs):
        g.manual_seed(seed0 + i)
        base = torch.randn((1, 1, H + 2 * pad, W + 2 * pad), generator=g, device=device)
        base = torch.nn.functional.pad(base, (r, r, 0, 0), mode="circular")
        base = torch.nn.functional.conv2d(base, k.view(1, 1, 1, -1))
        base = torch.nn.functional.pad(base, (0, 0, r, r), mode="circular")
        base = torch.nn.functional.conv2d(base, k.view(1, 1, -1, 1))[0, 0]
        lo, hi = base.min(), base.max()
        img = torch.round((base - lo) * (255.0 / (hi - lo))).to(torch.uint8)
        rs = np.random.default_rng(seed0 + i)
        tx, ty = (int(v) for v in rs.integers(-8, 9, size=2))
        shifts[i] = (tx, ty)
        frames[2 * i] = img[pad:pad + H, pad:pad + W]
        frames[2 * i + 1] = img[pad - ty:pad - ty + H, pad - tx:pad - tx + W]
    return frames, shifts


def cpu_baseline(frames_np, gpu_flow_np, n_sample, shifts=None):
    """Times the CPU oracle on the first n_sample pairs: one thread, then OpenMP across pairs.
    Also returns the endpoint errors of SURVEY 8(d): GPU vs oracle over all pixels, and both vs the ground truth
    (the integer translation of the synthetic pair) on the interior, >= 32 px from the border."""
    from oracle import oracle as O
    O.build()
    fr = np.ascontiguousarray(frames_np[:2 * n_sample])
    t0 = time.perf_counter()
    ref1 = O.farneback_batch(fr[:2], 0, nthreads=1, **PARAMS)
    t1 = time.perf_counter() - t0
    cores = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), O.omp_max_threads(), n_sample))
    t0 = time.perf_counter()
    ref = O.farneback_batch(fr, 0, nthreads=cores, **PARAMS)
    tn = time.perf_counter() - t0
    epe = np.linalg.norm(gpu_flow_np[:n_sample].astype(np.float64) - ref.astype(np.float64), axis=-1)
    assert np.array_equal(ref[0], ref1[0])
    gt = None
    if shifts is not None:
        g = np.asarray(shifts[:n_sample], np.float64)[:, None, None, :]
        inner = (slice(None), slice(32, -32), slice(32, -32))
        gt = {"gpu_mean": float(np.linalg.norm(gpu_flow_np[:n_sample][inner] - g, axis=-1).mean()),
              "cpu_oracle_mean": float(np.linalg.norm(ref[inner] - g, axis=-1).mean())}
    return {
        "value": round(n_sample / tn, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
        "sample": f"first {n_sample} pairs of the workload, oracle/farneback_oracle.c (OpenCV-algorithm CPU "
                  f"restatement, literal running-sum order), OpenMP across pairs; single thread: "
                  f"{1 / t1:.4f} pairs/s on 1 pair",
        "single_thread_value": round(1 / t1, 4),
        "host_cpus": os.cpu_count(),
    }, {"mean": float(epe.mean()), "p999": float(np.quantile(epe, 0.999)), "max": float(epe.max())}, gt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="frame pairs per GPU per step (config 3: 512)")
    ap.add_argument("--wave", type=int, default=256, help="pairs resident per wave (ofarn max_batch); 256 pairs = 53 GB of workspace")
    ap.add_argument("--unique", type=int, default=32, help="distinct synthetic pairs generated, then tiled")
    ap.add_argument("--cpu-sample", type=int, default=16, help="pairs timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel hipEvent timing")
    ap.add_argument("--prof-table", action="store_true", help="print per-(stage, level) timing rows to stderr")
    ap.add_argument("--no-two-stream", action="store_true",
                    help="skip the informational second measurement with per-kernel timing off (two internal streams)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (gloo: rehearsal of the multi-rank path with "
                         "several ranks sharing one GPU; the gather then goes through host memory)")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0 or args.batch < 1 or args.wave < 1:
        raise SystemExit("--steps and --batch and --wave must be >= 1, --warmup >= 0")

    import torch
    import hackathonopticalflow_amd as ofa
    from hackathonopticalflow_amd import distributed as D

    rank, local_rank, world = D.env_rank_world()
    if world != max(args.gpus, 1):
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: hackathonopticalflow_amd has no CPU path")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = D.init_process_group(args.backend) if world > 1 else None

    B = args.batch
    eng = ofa.FarnebackEngine(W, H, min(args.wave, B), dev_index, **PARAMS)
    P = len(ofa.grid_points(W, H, 30))
    plan = ofa.level_plan(W, H, **PARAMS)

    # synthetic input, resident in HBM before anything is timed
    uniq = min(args.unique, B)
    fr_u, shifts = make_frames_gpu(torch, uniq, 3000 + rank * B, device)
    frames = torch.empty((2 * B, H, W), dtype=torch.uint8, device=device)
    for i in range(B):
        frames[2 * i:2 * i + 2] = fr_u[2 * (i % uniq):2 * (i % uniq) + 2]
    flow = torch.empty((B, H, W, 2), dtype=torch.float32, device=device)
    mask = torch.zeros((B, P), dtype=torch.uint8, device=device)
    v = torch.zeros((B, P), dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.calc_batch_device(frames, 2 * B, W, H, ofa.PAIRS_INDEPENDENT, flow, mask, v, stream=stream)
        if dist is not None:
            if args.backend == "gloo":
                return D.gather_danger_maps(mask.cpu(), v.cpu(), B * world, dist)
            return D.gather_danger_maps(mask, v, B * world, dist)
        return mask, v

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_profile:
        eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = [] if args.no_profile else eng.profile_read()
    eng.profile_enable(False)

    # Informational second measurement (never `value`): the same K steps with per-kernel timing off.  The library then
    # alternates the waves of a batch over two internal streams, so the VALU-bound stages of one wave overlap the HBM-bound
    # iterations of the other; per-kernel durations lose their meaning there, which is why the timed region above keeps
    # everything on one stream.
    overlapped = None
    if not args.no_profile and not args.no_two_stream and min(args.wave, B) < B:
        step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        e2 = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([e2], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e2 = float(t.item())
        overlapped = B * world * args.steps / e2

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.prof_table:
        for r in sorted(prof, key=lambda r: -r["ms"]):
            per = r["ms"] / r["launches"]
            gbs = STAGE_BYTES.get(r["stage"], 0.0) * r["units"] / r["launches"] / (per * 1e-3) / 1e9
            print(f"  {r['stage']:16s} L{r['level']} launches={r['launches']:4d} total={r['ms']:9.3f} ms "
                  f"avg={per:8.4f} ms  alg={gbs:8.1f} GB/s", file=sys.stderr)
    pairs_total = B * world * args.steps
    value = pairs_total / elapsed
    alg_pair = algorithmic_bytes_per_pair(W, H, plan, PARAMS["iterations"])

    roofline = None
    if prof:
        dom = max(prof, key=lambda r: r["ms"])
        per_launch_s = dom["ms"] / dom["launches"] / 1e3
        bytes_launch = STAGE_BYTES.get(dom["stage"], 0.0) * dom["units"] / dom["launches"]
        ach = bytes_launch / per_launch_s / 1e9
        # HBM bytes from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as
        # MI355X_MICROARCH.md prescribes; provenance inside profiles/pmc_traffic.json), measured per work unit at
        # level 0 and scaled to this run's units per launch.
        traffic = None
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj) and dom["level"] == 0:
            try:
                per_unit = json.load(open(tj)).get(dom["stage"], {}).get("hbm_bytes_per_unit")
                if per_unit:
                    traffic = per_unit * dom["units"] / dom["launches"]
            except Exception:
                traffic = None
        total_ms = sum(r["ms"] for r in prof)
        roofline = {
            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
            "kernel": f"{dom['stage']}@level{dom['level']}",
            "kernel_avg_ms": round(per_launch_s * 1e3, 4), "kernel_launches": dom["launches"],
            "kernel_share_of_device_time": round(dom["ms"] / total_ms, 4),
            "algorithmic_bytes_per_launch": bytes_launch,
            "pipeline": {
                "algorithmic_bytes_per_pair": alg_pair,
                "achieved": round(alg_pair * value / world / 1e9, 1),
                "frac": round(alg_pair * value / world / 1e9 / HBM_PEAK_GBS, 4),
            },
            "stages_ms_per_step": {},
        }
        agg = {}
        for r in prof:
            agg[r["stage"]] = agg.get(r["stage"], 0.0) + r["ms"] / args.steps
        roofline["stages_ms_per_step"] = {k: round(val, 3) for k, val in agg.items()}

    out = {
        "metric": "frame-pairs/s @1080p (5-level, 3-iter)", "value": round(value, 2), "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config3: 1920x1080 batch={B} translated smooth-noise pairs per GPU "
                               f"({uniq} distinct, tiled), levels=5 iterations=3 winsize=15 poly_n=5, f32 with OpenCV's f64 "
                               f"accumulators, "
                               f"flow + danger maps to HBM" + (", RCCL all-gather of danger maps" if world > 1 else ""),
                   "pairs_per_gpu": B, "global_pairs": B * world, "wave": min(args.wave, B),
                   "parallelism": f"pairs sharded over {world} GPU(s)"},
        "roofline": roofline,
    }
    if overlapped is not None:
        out["two_stream_pairs_per_s"] = round(overlapped, 2)   # per-kernel timing off: waves overlap on two streams (informational)
    if world == 1 and args.cpu_sample > 0:
        ns = min(args.cpu_sample, uniq)
        cb, epe, gt = cpu_baseline(fr_u[:2 * ns].cpu().numpy(), flow[:ns].cpu().numpy(), ns, shifts)
        out["cpu_baseline"] = cb
        out["mean_epe_vs_cpu_oracle_px"] = epe["mean"]
        out["epe_vs_cpu_oracle"] = epe
        if gt:
            out["epe_vs_ground_truth_interior_px"] = gt
        out["speedup_vs_cpu_all_cores"] = round(value / cb["value"], 1)
        out["speedup_vs_cpu_1thread"] = round(value / cb["single_thread_value"], 1)
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
