"""Shard arithmetic and the danger-map gather, on CPU: fake communicator + 2 gloo ranks."""
import os
import socket
import sys

import numpy as np
import pytest

from hackathonopticalflow_amd import distributed as D
from hackathonopticalflow_amd.ofarn import PAIRS_CONSECUTIVE, PAIRS_INDEPENDENT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,world", [(512, 8), (64, 8), (10, 4), (3, 8), (0, 2), (7, 1)])
def test_shard_pairs_partition(n, world):
    spans = [D.shard_pairs(n, r, world) for r in range(world)]
    assert spans[0][0] == 0
    for (s0, c0), (s1, _) in zip(spans, spans[1:]):
        assert s1 == s0 + c0            # contiguous
    assert sum(c for _, c in spans) == n
    counts = [c for _, c in spans]
    assert max(counts) - min(counts) <= 1


def test_shard_frames_modes():
    # video order: neighbouring ranks overlap by one frame (DenseOF.py:525)
    s0 = D.shard_frames(9, PAIRS_CONSECUTIVE, 0, 2)
    s1 = D.shard_frames(9, PAIRS_CONSECUTIVE, 1, 2)
    assert (s0.pair_start, s0.pair_count, s0.frame_start, s0.frame_count) == (0, 4, 0, 5)
    assert (s1.pair_start, s1.pair_count, s1.frame_start, s1.frame_count) == (4, 4, 4, 5)
    s = D.shard_frames(16, PAIRS_INDEPENDENT, 3, 4)
    assert (s.pair_start, s.pair_count, s.frame_start, s.frame_count) == (6, 2, 12, 4)
    with pytest.raises(ValueError):
        D.shard_frames(15, PAIRS_INDEPENDENT, 0, 2)
    assert D.shard_frames(1, PAIRS_CONSECUTIVE, 0, 2).frame_count == 0


def test_fake_communicator_gather_order():
    world, n, P = 3, 8, 11
    rng = np.random.default_rng(0)
    full_m = rng.integers(0, 2, (n, P)).astype(np.uint8)
    full_v = rng.integers(0, 256, (n, P)).astype(np.uint8)
    parts_m, parts_v = [], []
    for r in range(world):
        s, c = D.shard_pairs(n, r, world)
        parts_m.append(full_m[s:s + c])
        parts_v.append(full_v[s:s + c])
    m, v = D.FakeCommunicator(world).gather(parts_m, parts_v, n)
    np.testing.assert_array_equal(m, full_m)
    np.testing.assert_array_equal(v, full_v)


def _worker(rank, world, port, n_pairs, P, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    from hackathonopticalflow_amd import distributed as DD
    dist = DD.init_process_group("gloo")
    s, c = DD.shard_pairs(n_pairs, rank, world)
    rng = np.random.default_rng(123)
    full_m = rng.integers(0, 2, (n_pairs, P)).astype(np.uint8)
    full_v = rng.integers(0, 256, (n_pairs, P)).astype(np.uint8)
    m, v = DD.gather_danger_maps(torch.from_numpy(full_m[s:s + c].copy()), torch.from_numpy(full_v[s:s + c].copy()),
                                 n_pairs, dist)
    ok = np.array_equal(m.numpy(), full_m) and np.array_equal(v.numpy(), full_v)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("n_pairs", [8, 5])
def test_gather_danger_maps_two_gloo_ranks(n_pairs):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, 352, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside torch.distributed.run must spawn the two ranks itself (a fresh child through
    torch.distributed.run, before anything touches HIP) and return the child's exit code.  Without a GPU the ranks stop
    at "bench.py needs a GPU": seeing THAT message from a rank proves the launch path (the launcher stops the other rank
    as soon as one has failed, so it may appear once or twice); with a GPU the same
    command is tests/test_gpu_parity.py::test_bench_two_gloo_ranks_share_the_gpu."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by the gpu-marked rehearsal")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "4",
                        "--batch", "4", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr and "torch.distributed" in r.stderr, r.stderr[-2000:]
    # a rank count that does not match the launch is refused
    env2 = dict(env, RANK="0", WORLD_SIZE="3", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env2)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


def _ws1_worker(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    from hackathonopticalflow_amd import distributed as DD
    assert DD.init_process_group("gloo") is None                    # world size 1 without force: no group
    dist = DD.init_process_group("gloo", force=True)                # bench.py --force-dist: a one-rank group still runs the collective
    P, n = 352, 6
    rng = np.random.default_rng(5)
    ok = True
    g = DD.DangerGather(n, P, "cpu", dist)
    ptrs = (g.send.data_ptr(), g.recv.data_ptr(), g.mask_all.data_ptr(), g.v_all.data_ptr())
    ok &= g.even and g.mask_all.data_ptr() == g.recv.data_ptr()     # even shards: the results are views of the receive block
    for it in range(3):
        m = torch.from_numpy(rng.integers(0, 2, (n, P)).astype(np.uint8))
        v = torch.from_numpy(rng.integers(0, 256, (n, P)).astype(np.uint8))
        ma, va = g(m, v)
        ok &= bool((ma == m).all()) and bool((va == v).all()) and ma.shape == (n, P)
        ok &= ptrs == (g.send.data_ptr(), g.recv.data_ptr(), g.mask_all.data_ptr(), g.v_all.data_ptr())   # nothing reallocated
        ma2, _ = DD.gather_danger_maps(m, v, n, dist)               # the cached gather of the functional form
        ok &= bool((ma2 == m).all())
    ok &= g.calls == 3
    try:
        g(torch.zeros((n - 1, P), dtype=torch.uint8), torch.zeros((n - 1, P), dtype=torch.uint8))
        ok = False
    except ValueError:
        pass
    DD.reset_gathers()
    dist.destroy_process_group()
    q.put(ok)


def test_gather_at_world_size_one_is_allocation_free():
    """The collective at world size 1 (what `bench.py --gpus 1 --force-dist` runs, there over RCCL on device tensors): buffers
    allocated once, results are views of the receive block, wrong shard sizes are refused."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_ws1_worker, args=(port, q))
    p.start()
    assert q.get(timeout=120) is True
    p.join(timeout=60)
    assert p.exitcode == 0


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("n_pairs", [0, 1, 5, 8, 13, 64, 67, 512])
def test_c_abi_gather_plan_against_the_python_mirror(n_pairs, world):
    """the G > 1 branches of ofarn_multi_* (in-place all-gather at rank * cap * P, the `even` shortcut, compaction of
    ragged shards) cannot run on a one-GPU box.  Their index arithmetic lives in one host-only function (ofarn_gather_plan); here it
    is checked on the CPU for ragged and even pair counts against the Python mirror, and the whole data movement is replayed with
    NumPy: every rank's rows written at gather_off[rank] of a padded array, "all-gathered", compacted to global_off -- the result
    must be the maps in global pair order on every rank."""
    import ctypes as C
    from hackathonopticalflow_amd import ofarn
    lib = ofarn.load_library()
    P = 37
    start, count = (C.c_int * world)(), (C.c_int * world)()
    goff, xoff = (C.c_uint64 * world)(), (C.c_uint64 * world)()
    cap, even = C.c_int(), C.c_int()
    assert lib.ofarn_gather_plan(n_pairs, world, P, start, count, C.byref(cap), C.byref(even), goff, xoff) == 0
    spans = [D.shard_pairs(n_pairs, r, world) for r in range(world)]
    assert [(start[r], count[r]) for r in range(world)] == spans
    assert cap.value == max(c for _, c in spans)
    assert bool(even.value) == (n_pairs == cap.value * world)
    assert [goff[r] for r in range(world)] == [r * cap.value * P for r in range(world)]
    assert [xoff[r] for r in range(world)] == [spans[r][0] * P for r in range(world)]
    # replay: what the devices do with these offsets
    rng = np.random.default_rng(n_pairs * 31 + world)
    full = rng.integers(0, 256, (n_pairs, P)).astype(np.uint8)
    padded_bytes = world * cap.value * P
    per_rank = []
    for r in range(world):                                   # phase 1: each rank writes its shard at its own rows of its own buffer
        buf = np.full(padded_bytes, 0xEE, np.uint8)
        s, c = spans[r]
        buf[goff[r]:goff[r] + c * P] = full[s:s + c].ravel()
        per_rank.append(buf)
    gathered = np.full(padded_bytes, 0xEE, np.uint8)         # phase 2: in-place all-gather = every rank's cap * P block, in rank order
    for r in range(world):
        gathered[goff[r]:goff[r] + cap.value * P] = per_rank[r][goff[r]:goff[r] + cap.value * P]
    if even.value:                                           # the padded array IS the global one
        np.testing.assert_array_equal(gathered.reshape(n_pairs, P), full)
    else:                                                    # phase 3: compaction
        out = np.zeros(n_pairs * P, np.uint8)
        for r in range(world):
            c = spans[r][1]
            out[xoff[r]:xoff[r] + c * P] = gathered[goff[r]:goff[r] + c * P]
        np.testing.assert_array_equal(out.reshape(n_pairs, P), full)
    # the torch.distributed form's gather (DangerGather's layout) agrees with it
    parts = [full[s:s + c] for s, c in spans]
    m, _ = D.FakeCommunicator(world).gather(parts, parts, n_pairs)
    np.testing.assert_array_equal(m, full)
    assert lib.ofarn_gather_plan(-1, world, P, None, None, None, None, None, None) < 0
    assert lib.ofarn_gather_plan(4, 0, P, None, None, None, None, None, None) < 0


def test_bench_inproc_argument_plumbing():
    """`bench.py --multi inproc`: shards per device for configs 3 (weak), 4 and 5 (strong) and the argument checks that need no GPU."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    g, sh = bench.inproc_shards(bench.CONFIGS[3], 8)
    assert g == 4096 and sh == [(512 * r, 512) for r in range(8)]
    g, sh = bench.inproc_shards(bench.CONFIGS[4], 8)
    assert g == 512 and sh == [(64 * r, 64) for r in range(8)]
    g, sh = bench.inproc_shards(bench.CONFIGS[5], 8)
    assert g == 64 and sh == [(8 * r, 8) for r in range(8)]
    g, sh = bench.inproc_shards(bench.CONFIGS[4], 3, batch=10)
    assert g == 10 and sh == [(0, 4), (4, 3), (7, 3)]
    with pytest.raises(ValueError):
        bench.inproc_shards(bench.CONFIGS[5], 8, batch=5)
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--multi", "inproc", "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "ONE process" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--multi", "inproc", "--config", "2"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "configs 3, 4, 5" in r.stderr
