"""Multi-GPU at the C-ABI (ofarn_multi_*): one process, one context + host thread + stream per device, pairs sharded, ONE RCCL
all-gather of the danger maps.  The GPU box has one MI355X: the n = 1 leg runs the whole path -- ncclCommInitAll, the
in-place ncclAllGather inside an RCCL group, the shard arithmetic with world = 1 -- and must equal ofarn_calc_batch bit for bit.
With more devices visible the same tests run on all of them."""
import os

import numpy as np
import pytest

from hackathonopticalflow_amd.synth import translated_pair, translated_pairs, warped_pairs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import hackathonopticalflow_amd as H
    H.load_library()
    return H


def _ndev():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("n_pairs,mode", [(6, 0), (5, 0), (7, 1), (1, 0)])
def test_multi_host_batch_equals_single_context(H, oracle, n_pairs, mode):
    w, h = 320, 240
    n_frames = n_pairs + 1 if mode == 1 else 2 * n_pairs
    frames, _ = translated_pairs(n_frames // 2 + 1, h, w, 5100, max_shift=3)
    frames = frames[:n_frames]
    G = max(1, min(_ndev(), 4))
    with H.MultiGpuEngine(G, w, h, 4, levels=3) as multi, H.FarnebackEngine(w, h, 4, levels=3) as one:
        flow, mask, v = multi.calc_batch(frames, mode)
        f1, m1, v1 = one.calc_batch(frames, mode)
        np.testing.assert_array_equal(flow, f1)
        np.testing.assert_array_equal(mask, m1)
        np.testing.assert_array_equal(v, v1)
        info = multi.info()
        assert info["allgather_calls"] == 2 * G and info["rccl_version"] > 0 and info["last_device_ms"] > 0
        # danger maps only, then flow only
        _, mask2, v2 = multi.calc_batch(frames, mode, want_flow=False)
        np.testing.assert_array_equal(mask2, m1)
        np.testing.assert_array_equal(v2, v1)
        flow3, m3, _ = multi.calc_batch(frames, mode, want_danger=False)
        assert m3 is None
        np.testing.assert_array_equal(flow3, f1)
        assert multi.info()["allgather_calls"] == 4 * G
    i = n_pairs - 1
    a, b = (frames[i], frames[i + 1]) if mode == 1 else (frames[2 * i], frames[2 * i + 1])
    np.testing.assert_array_equal(flow[i], oracle.farneback(a, b, levels=3, box_mode=oracle.BOX_BLOCKED))


def test_multi_device_resident_shards(H, oracle):
    torch = pytest.importorskip("torch")
    w, h, n_pairs = 200, 150, 5
    frames, _ = warped_pairs(n_pairs, h, w, 5200)
    G = max(1, min(_ndev(), 4))
    P = len(H.grid_points(w, h, 30))
    d_frames, d_flow, d_m, d_v = [], [], [], []
    for g in range(G):
        s, c = H.shard_pairs_c(n_pairs, g, G)
        dev = torch.device("cuda", g)
        d_frames.append(torch.from_numpy(frames[2 * s:2 * (s + c)].copy()).to(dev) if c else torch.empty(0, dtype=torch.uint8, device=dev))
        d_flow.append(torch.empty((c, h, w, 2), dtype=torch.float32, device=dev))
        d_m.append(torch.zeros((n_pairs, P), dtype=torch.uint8, device=dev))
        d_v.append(torch.zeros((n_pairs, P), dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    with H.MultiGpuEngine(G, w, h, 8, levels=2) as multi:
        multi.calc_batch_device(d_frames, n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, d_m, d_v)
        multi.synchronize()
        for g in range(G):
            s, c = H.shard_pairs_c(n_pairs, g, G)
            got = d_flow[g].cpu().numpy()
            for i in range(c):
                ref = oracle.farneback(frames[2 * (s + i)], frames[2 * (s + i) + 1], levels=2, box_mode=oracle.BOX_BLOCKED)
                np.testing.assert_array_equal(got[i], ref)
                m_ref, v_ref = oracle.danger_map_numpy(ref, w, h, 30)
                for q in range(G):                     # every device holds every pair's map after the gather
                    np.testing.assert_array_equal(d_m[q][s + i].cpu().numpy(), m_ref)
                    np.testing.assert_array_equal(d_v[q][s + i].cpu().numpy(), v_ref)
        with pytest.raises(ValueError):
            multi.calc_batch_device(d_frames[:0], n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, d_m, d_v)


def test_multi_create_errors(H):
    with pytest.raises(ValueError):
        H.MultiGpuEngine([0, 0], 64, 48, 2)                   # one rank per GPU
    with pytest.raises(ValueError):
        H.MultiGpuEngine([_ndev()], 64, 48, 2)
    with pytest.raises(ValueError):
        H.MultiGpuEngine([], 64, 48, 2)
    with H.MultiGpuEngine([0], 64, 48, 2) as multi:
        with pytest.raises(ValueError):
            multi.calc_batch(np.zeros((2, 100, 100), np.uint8))   # larger than the contexts
        with pytest.raises(ValueError):
            multi.calc_batch(np.zeros((3, 48, 64), np.uint8))     # odd number of frames in independent mode
        f, m, v = multi.calc_batch(np.zeros((0, 48, 64), np.uint8))
        assert f.shape == (0, 48, 64, 2) and m.shape[0] == 0


def test_c_program_stream_and_multi(H, oracle, tmp_path):
    """examples/c_abi_stream_multi.c, a C99 program with no Python or torch in the process: the frame loop through
    ofarn_stream_next (pinned output) and the same pairs as a batch through ofarn_multi_calc_batch (RCCL opened by dlopen from a
    plain C process) give identical flows, and those equal the oracle's."""
    import os
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_abi import _build_c_example
    exe = _build_c_example(tmp_path, "c_abi_stream_multi")
    w, h, n = 320, 240, 5
    frames, _ = translated_pairs(3, h, w, 5300, max_shift=3)
    frames = frames[:n]
    (tmp_path / "frames.raw").write_bytes(frames.tobytes())
    r = subprocess.run([exe, str(tmp_path / "frames.raw"), str(w), str(h), str(n), str(tmp_path / "out"), "3", "1"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical" in r.stdout and "ncclAllGather calls" in r.stdout
    fs = np.fromfile(tmp_path / "out.stream.raw", np.float32).reshape(n - 1, h, w, 2)
    fm = np.fromfile(tmp_path / "out.multi.raw", np.float32).reshape(n - 1, h, w, 2)
    np.testing.assert_array_equal(fs, fm)
    P = len(H.grid_points(w, h, 30))
    mask = np.fromfile(tmp_path / "out.mask.raw", np.uint8).reshape(n - 1, P)
    for i in range(n - 1):
        ref = oracle.farneback(frames[i], frames[i + 1], levels=3, box_mode=oracle.BOX_BLOCKED)
        np.testing.assert_array_equal(fs[i], ref)
        np.testing.assert_array_equal(mask[i], oracle.danger_map_numpy(ref, w, h, 30)[0])


@pytest.mark.parametrize("config,batch", [(3, 16), (4, 16), (5, 2)])
def test_bench_inproc_at_one_device(config, batch):
    """SURVEY 8(e) as written has a bench line -- `bench.py --multi inproc --gpus N`: one process,
    MultiGpuEngine.calc_batch_device in the timed region (device-resident shards, the in-place ncclAllGather group inside).  At
    N = 1 (what this box has) the value must be that of the single-context line on the same box (within a few per cent at this small
    batch; the full-size comparison is profiles/r04_bench_inproc_config3.json vs r04_bench_final.json), the gather check true, the
    collective counted, same JSON schema."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    common = ["--config", str(config), "--batch", str(batch), "--steps", "4", "--warmup", "2", "--cpu-sample", "0", "--no-family-check",
              "--no-two-stream"]
    outs = {}
    for mode in ("inproc", "ranks"):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--multi", mode] + common
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", f"bench_{mode}_n1_config{config}.log"), "w") as f:
            f.write("$ " + " ".join(cmd) + "\n" + r.stdout + "\n---- stderr ----\n" + r.stderr[-4000:])
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        outs[mode] = json.loads(lines[0])
    a, b = outs["inproc"], outs["ranks"]
    assert a["n_gpus"] == 1 and a["ranks"] == 1 and a["scaling"] == b["scaling"] and a["metric"] == b["metric"] and a["unit"] == b["unit"]
    assert a["config"]["parallelism"] == "1 devices in one process, ncclCommInitAll"
    assert a["config"]["global_pairs"] == batch and a["config"]["pairs_per_gpu"] == batch
    assert a["gathered_danger_maps_checked"] is True
    assert a["collective"]["calls"] == 2 * 4 and a["collective"]["world"] == 1          # mask + V, one group per step
    assert a["roofline"]["frac"] <= 1.0 and a["roofline"]["kernel"].startswith("flow_iter")
    assert set(b) - {"two_stream_pairs_per_s", "other_family"} <= set(a) | {"cpu_baseline", "parity"}
    assert abs(a["value"] / b["value"] - 1) < 0.10, (a["value"], b["value"])


def test_multi_leaves_the_callers_device_and_thread_alone(H, oracle):
    """ofarn_multi_* does all device work on its persistent per-device worker threads (created by ofarn_multi_create, joined by
    ofarn_multi_destroy): the calling thread's current device is never changed -- checked with the device-scope hook, which would
    record a restore if any ofarn_ctx entry point ran on the CALLING thread -- and the number of threads of the process does not
    grow from call to call (round 3 started 2 x G threads per call)."""
    import ctypes as C
    import threading
    lib = H.load_library()
    frames = np.stack([f for s in (1, 2, 3) for f in translated_pair(120, 160, s, max_shift=2)[:2]])
    scopes, restored = C.c_int(0), C.c_int(0)
    with H.MultiGpuEngine([0], 160, 120, 4, levels=2) as eng:
        eng.calc_batch(frames)
        n_threads = threading.active_count()
        os_threads = len(os.listdir("/proc/self/task"))
        lib.ofarn_debug_device_scope(-1, C.byref(scopes), C.byref(restored))
        n0 = scopes.value
        for _ in range(5):
            flow, mask, v = eng.calc_batch(frames)
        lib.ofarn_debug_device_scope(-1, C.byref(scopes), C.byref(restored))
        assert scopes.value == n0                     # no context entry point ran on this thread
        assert threading.active_count() == n_threads and len(os.listdir("/proc/self/task")) <= os_threads + 1
    with H.FarnebackEngine(160, 120, 4, levels=2) as single:
        f1, m1, v1 = single.calc_batch(frames)
    np.testing.assert_array_equal(flow, f1)
    np.testing.assert_array_equal(mask, m1)
    np.testing.assert_array_equal(v, v1)


@pytest.mark.parametrize("G", [2, 3, 4, 8])
def test_multi_rank_branches_run_on_one_gpu_in_loopback(H, oracle, monkeypatch, G):
    """the G > 1 branches of ofarn_multi_* -- one worker thread per rank, ragged and empty shards, the in-place gather at
    rank * cap * P, the `even` shortcut straight into caller arrays, the compaction of ragged shards, rank 0's host copy -- had never
    run on hardware: RCCL cannot put two ranks on one GPU, and the box has one.  OFARN_MULTI_LOOPBACK=1 lists the same device G
    times (one context, worker and stream per rank) and replaces the all-gather by the copies it stands for (every rank fetches every
    rank's block behind an event of the producer); everything else is the production path.  Results must equal the single context's
    for even, ragged and fewer-pairs-than-ranks batches, both pair modes, host and device-resident variants."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("OFARN_MULTI_LOOPBACK", "1")
    w, h, kw = 200, 150, dict(levels=2)
    P = len(H.grid_points(w, h, 30))
    with H.MultiGpuEngine([0] * G, w, h, 3, **kw) as multi, H.FarnebackEngine(w, h, 8, **kw) as one:
        assert multi.info()["devices"] == [0] * G
        for n_pairs, mode in ((2 * G, 0), (2 * G + 1, 0), (G - 1, 0), (G + 2, 1), (1, 1)):
            n_frames = n_pairs + 1 if mode == 1 else 2 * n_pairs
            frames, _ = warped_pairs(n_frames // 2 + 1, h, w, 5300 + n_pairs)
            frames = frames[:n_frames]
            f1, m1, v1 = one.calc_batch(frames, mode)
            flow, mask, v = multi.calc_batch(frames, mode)                       # host variant: rank 0's gathered maps come back
            np.testing.assert_array_equal(flow, f1)
            np.testing.assert_array_equal(mask, m1)
            np.testing.assert_array_equal(v, v1)
            if mode == 1:
                continue
            # device-resident variant: every rank's gathered arrays must hold ALL pairs' maps in global order
            d_frames, d_flow, d_m, d_v = [], [], [], []
            for g in range(G):
                s, c = H.shard_pairs_c(n_pairs, g, G)
                d_frames.append(torch.from_numpy(frames[2 * s:2 * (s + c)].copy()).cuda() if c else torch.empty(0, dtype=torch.uint8, device="cuda"))
                d_flow.append(torch.empty((c, h, w, 2), dtype=torch.float32, device="cuda"))
                d_m.append(torch.full((n_pairs, P), 0xEE, dtype=torch.uint8, device="cuda"))
                d_v.append(torch.full((n_pairs, P), 0xEE, dtype=torch.uint8, device="cuda"))
            torch.cuda.synchronize()
            multi.calc_batch_device(d_frames, n_pairs, w, h, H.PAIRS_INDEPENDENT, d_flow, d_m, d_v)
            multi.synchronize()
            for g in range(G):
                s, c = H.shard_pairs_c(n_pairs, g, G)
                np.testing.assert_array_equal(d_flow[g].cpu().numpy(), f1[s:s + c])
                np.testing.assert_array_equal(d_m[g].cpu().numpy(), m1)
                np.testing.assert_array_equal(d_v[g].cpu().numpy(), v1)
    # the first pair of the last batch against the oracle, so that "equal to the single context" means "right"
    np.testing.assert_array_equal(f1[0], oracle.farneback(frames[0], frames[1], box_mode=oracle.BOX_BLOCKED, **kw))
    monkeypatch.delenv("OFARN_MULTI_LOOPBACK")
    with pytest.raises(ValueError):
        H.MultiGpuEngine([0, 0], 64, 48, 2)                   # without the test switch: one rank per GPU
