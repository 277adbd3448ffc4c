"""world_size-2 `gloo` rehearsal (CPU) of the multi-GPU path: contiguous frame sharding with a one-frame halo
(no collective on the data path), the max-over-ranks clock bench.py uses, and the optional keyframe-descriptor
all-gather for loop-closure candidates (on the GPU box the same code runs over RCCL/xGMI with backend "nccl")."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aria_slam_amd import loopdb, shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = shard.shard_plan(n_frames, rank, world)
        # every rank "extracts" its frames: descriptor rows are a deterministic function of the frame index
        def fake_desc(i):
            rng = np.random.default_rng(1000 + i)
            return rng.integers(0, 256, (16, 32), dtype=np.uint8)
        lo, hi = plan["extract"]
        local = {i: fake_desc(i) for i in range(lo, hi)}
        pairs_ok = all(q_ in local and t_ in local for q_, t_ in plan["pairs"])
        # max-over-ranks clock (bench.py)
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # keyframe DB: every 4th owned frame is a keyframe; all-gather the padded slots
        db = loopdb.KeyframeDB(k_cap=8, rows=16, device="cpu")
        for i in range(plan["lo"], plan["hi"]):
            if i % 4 == 0:
                db.add(i, torch.from_numpy(local[i]), 16 - (i % 3))
        g = db.all_gather()
        q.put((rank, plan, pairs_ok, float(t.item()), g.ids.numpy().copy(), g.counts.numpy().copy(),
               g.desc.numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_frame_ranges_partition_the_sequence():
    for n in (0, 1, 7, 8192, 32768, 12345):
        for world in (1, 2, 3, 8):
            rs = [shard.frame_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
            pairs = [p for r in range(world) for p in shard.shard_plan(n, r, world)["pairs"]]
            assert pairs == [(i, i - 1) for i in range(1, n)]          # every consecutive pair exactly once


def test_world2_gloo_shard_halo_and_allgather():
    world, n_frames = 2, 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, plan0, ok0, t0, ids0, cnt0, desc0), (r1, plan1, ok1, t1, ids1, cnt1, desc1) = res
    assert ok0 and ok1
    assert plan0["extract"] == (0, 18) and plan1["extract"] == (17, 37)      # rank 1 recomputes frame 17 (halo)
    assert t0 == 2.0 and t1 == 2.0                                          # max over ranks
    # both ranks hold the same gathered DB: rank-0 slots then rank-1 slots
    assert np.array_equal(ids0, ids1) and np.array_equal(cnt0, cnt1) and np.array_equal(desc0, desc1)
    want_ids = [0, 4, 8, 12, 16, -1, -1, -1, 20, 24, 28, 32, 36, -1, -1, -1]
    assert ids0.tolist() == want_ids
    for slot, kid in enumerate(want_ids):
        if kid >= 0:
            c = 16 - (kid % 3)
            assert cnt0[slot] == c
            ref = np.random.default_rng(1000 + kid).integers(0, 256, (16, 32), dtype=np.uint8)
            assert np.array_equal(desc0[slot, :c], ref[:c]) and desc0[slot, c:].max(initial=0) == 0
        else:
            assert cnt0[slot] == 0


def test_score_candidates_matches_oracle_rules(oracle):
    rng = np.random.default_rng(4)
    q = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    blocks, ids = [], []
    for k in range(8):
        b = rng.integers(0, 256, (200, 32), dtype=np.uint8)
        n_copy = [0, 10, 30, 60, 100, 150, 200, 200][k]
        b[:n_copy] = q[:n_copy]
        blocks.append(b)
        ids.append(k * 10)
    good = [oracle.count_good_matches_f64(q, b, 0.7) for b in blocks]
    for qid, mfb in ((1000, 30), (75, 30), (60, 0)):
        ci, cs = oracle.loop_candidates(q, qid, blocks, ids, mfb)
        got = loopdb.score_candidates(good, 200, qid, ids, [200] * 8, mfb)
        assert [i for i, _ in got] == ci.tolist()
        assert [s for _, s in got] == cs.tolist()
