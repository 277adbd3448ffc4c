"""bench.py as a program: the N-rank launcher (CPU, gloo), the rehearsal of the N>1 path on one GPU, the verification of
the 8192-frame one-chunk two-stream path the headline number comes from, and the RCCL keyframe-descriptor exchange
(BASELINE.json configs[4]; reference semantics src/legacy/LoopClosure.cpp:28-30, 72-114)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr.decode()


@pytest.mark.parametrize("n", [1, 2, 3])
def test_gpus_flag_launches_n_ranks(n):
    """`bench.py --gpus N` without WORLD_SIZE starts N rank processes that rendezvous (probe mode: gloo, no GPU)."""
    rc, out, err = _run(["--gpus", str(n), "--pairs", "8"], {"ARIA_BENCH_PROBE": "1"}, timeout=300)
    assert rc == 0, err
    assert out["n_gpus"] == n and out["ranks_seen"] == n and out["rank_sum"] == n * (n - 1) // 2
    assert out["rank0_frames"] == [0, 16]


def test_launcher_reports_a_failing_rank():
    rc, out, err = _run(["--gpus", "2", "--pairs", "8", "--bogus-flag"], {"ARIA_BENCH_PROBE": "1"}, timeout=300)
    assert rc != 0 and out is None


def test_external_world_size_wins_over_gpus_flag():
    """Under torch.distributed.run the ranks exist already: no second level of processes is started."""
    rc, out, err = _run(["--gpus", "1", "--pairs", "8"],
                        {"ARIA_BENCH_PROBE": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=300)
    assert rc == 0 and out["n_gpus"] == 1


def test_bench_prefix_golden_matches_the_oracle(aria, oracle):
    """tests/golden/bench_prefix.json (what bench.py's `verified` compares with) is reproducible from the oracle."""
    sys.path.insert(0, ROOT)
    import bench as B
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_prefix.json")))["640x480_n2000"]
    p = oracle.default_params(2000)
    a, b = aria.synth_frame_pair(2, 640, 480)          # frames 2 and 3 of the default sequence
    (ka, da), (kb, db) = oracle.orb_extract(a, p), oracle.orb_extract(b, p)
    assert B.frame_digest(len(ka), ka, da)[:16] == g["frame"][2]
    assert B.frame_digest(len(kb), kb, db)[:16] == g["frame"][3]
    m = oracle.match_ratio(db, da, g["ratio"])
    assert B.match_digest(len(m), m)[:16] == g["match"][3]


@pytest.mark.gpu
def test_default_bench_chunk8192_two_streams_is_verified():
    """The path the headline number uses (one 8192-frame chunk, extractor and matcher on two streams, double-buffered
    outputs): hashed frame by frame, first 256 frames against the committed oracle digests, sampled frames against
    the single-frame entry points."""
    rc, out, err = _run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {}, timeout=1200)
    assert rc == 0, err[-2000:]
    assert out["verified"] is True and out["n_gpus"] == 1
    v = out["verification"]
    assert v["frames_hashed"] == 8192 and v["oracle_prefix_frames"] == 256 and not v["problems"]
    assert out["config"]["chunk_frames"] == 8192 and "configs[2]" in out["config"]["workload"]
    assert out["loop_closure"]["keyframes"] == 500 and "error" not in out["loop_closure"]
    # VERDICT r3 items 5, 8 and ADVICE r3: staging figures are medians of warmed copies (pinned is not slower than pageable),
    # the host-fed leg reproduces the resident step, the PMC traffic is tied to the kernel source it was measured on, and the
    # matcher label comes from the handle
    st = out["h2d_staging"]
    assert len(st["pageable_GBs_all"]) == 3 and len(st["pinned_GBs_all"]) == 3
    assert st["pinned_GBs"] >= 0.9 * st["pageable_GBs"], st
    assert st["streamed"]["equals_resident_step"] is True and st["streamed_frames_per_s"] > 10000, st["streamed"]
    assert out["roofline"]["traffic_stale"] in (True, False) and out["roofline"]["traffic_source"]
    assert out["roofline"]["matcher"]["launched"] == "k_knn2_fp4" and out["roofline"]["matcher"]["peak"] == 10000.0


@pytest.mark.gpu
def test_two_rank_rehearsal_on_one_gpu():
    """N > 1 code path end to end on the one-GPU box (both ranks on cuda:0, gloo): prints n_gpus 2, every rank verifies
    its shard (rank 1 against its recomputed halo frame), the all-gathered DB is scanned."""
    rc, out, err = _run(["--gpus", "2", "--pairs", "128", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                        {"ARIA_BENCH_REHEARSAL": "1"}, timeout=900)
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 2 and out["rehearsal"] is True and out["verified"] is True
    assert out["loop_closure"]["keyframes"] == 500 and out["loop_closure"]["allgather_backend"].startswith("gloo")
    # the N > 1 line can be audited -- per-rank rates and verification, the slowest rank named, the
    # all-gather's received-bytes rate beside the one-shot xGMI bound
    r = out["ranks"]
    assert len(r["frames_per_s"]) == 2 and len(r["verified"]) == 2 and all(r["verified"])
    assert r["min_frames_per_s"] <= r["max_frames_per_s"] and r["slowest_rank"] in (0, 1)
    assert out["value"] <= sum(r["frames_per_s"]) * 1.001            # the barrier-to-barrier clock is never shorter than a rank's own
    lc = out["loop_closure"]
    assert lc["xgmi_one_shot_bound_GBs"] == 153.0 and lc["allgather_GBs"] > 0 and len(lc["allgather_ms_all"]) == 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_keyframe_db_allgather_over_rccl_and_scan_equals_oracle(aria, oracle):
    """configs[4] slot sizes: 500 keyframe slots x 2064 rows (2000 kp + tie slack) resident in HBM, all-gathered with
    backend "nccl" (= RCCL; world size 1 on the one-GPU box, the collective still runs through RCCL), then scanned on the
    device for one query; good-match counts and the top-5 candidates equal the oracle's restatement of
    LoopClosureDetector::findCandidates."""
    import torch
    import torch.distributed as dist
    from aria_slam_amd import loopdb
    dev = torch.device("cuda", 0)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(77)
        rows, k_cap, nq = 2064, 500, 2000
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        db = loopdb.KeyframeDB(k_cap, rows, dev)
        blocks, ids = [], []
        for k in range(k_cap):
            n = 2000 if k % 25 == 0 else int(rng.integers(40, 260))
            b = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            n_copy = min(n, [0, 150, 300, 450, 700, 1000, 2000][k % 7]) if k % 25 == 0 else min(n, int(rng.integers(0, 40)))
            idx = rng.permutation(nq)[:n_copy]
            b[:n_copy] = q[idx]
            b[:n_copy, 0] ^= rng.integers(0, 2, n_copy, dtype=np.uint8)        # a little noise
            blocks.append(b)
            ids.append(3 * k)
            db.add(3 * k, torch.from_numpy(b).to(dev), n)
        g = db.all_gather()                                   # RCCL
        torch.cuda.synchronize(dev)
        assert g.k_cap == k_cap and torch.equal(g.desc, db.desc) and torch.equal(g.counts, db.counts)
        m = aria.HipMatcher(device=0, max_query=rows, max_train=rows)
        try:
            good = torch.zeros((k_cap,), dtype=torch.int32, device=dev)
            m.match_db_device(torch.from_numpy(q).to(dev), nq, g.desc, g.counts, k_cap, rows * 32, 0.7, good)
            m.sync()
            good = good.cpu().numpy()
            want_good = [oracle.count_good_matches_f64(q, b, 0.7) for b in blocks]
            assert good.tolist() == want_good
            for qid, mfb in ((3 * k_cap + 500, 200), (900, 200), (3 * k_cap, 0)):
                got = g.find_candidates(m, torch.from_numpy(q).to(dev), nq, qid, mfb, 0.7)
                ci, cs = oracle.loop_candidates(q, qid, blocks, ids, mfb)
                assert [i for i, _ in got] == ci.tolist()
                assert [s for _, s in got] == cs.tolist()
        finally:
            m.close()
    finally:
        dist.destroy_process_group()
