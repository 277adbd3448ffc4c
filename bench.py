#!/usr/bin/env python3
"""Headline benchmark: frames/s of ORB extract + brute-force Hamming match on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): a recorded sequence of 4096
synthetic 640x480 frame pairs = 8192 grayscale frames, 2000 keypoints per frame, resident in HBM before the
timed region. One "step" = one pass of the hot path over the whole sequence: every frame is extracted (pyramid,
FAST, Harris, IC angle, rBRIEF) and matched (kNN-2 + ratio 0.75) against the previous frame's descriptors
(query = current, train = previous; the first frame of a rank matches the frame before its range -- the halo frame
of aria_slam_amd/shard.py; rank 0 wraps around to its own last frame so that every frame costs 1 extract + 1 match).
Data: synthetic (SURVEY.md 8d generator; the node-wide sequence is seeds 1..4096*N, rank r owns a contiguous range).

Multi-GPU: one process per GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks clock);
the sequence shards by contiguous frame ranges (aria_slam_amd/shard.py), each rank owns 4096 pairs, no data-path
collective -> weak scaling. `python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks
itself (launch_ranks: fresh child processes, the parent never touches the GPU); under torch.distributed.run the
ranks already exist and --gpus only has to agree with WORLD_SIZE. After the timed region, untimed: the last step's
results are verified (`verified`) and the loop-closure leg of BASELINE.json configs[4] runs once (keyframe
descriptor DB, RCCL all-gather across the ranks, device scan: `loop_closure`).

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GOLDEN_PREFIX = os.path.join(ROOT, "tests", "golden", "bench_prefix.json")


def host_cores():
    """Cores this process may really use: affinity, capped by the cgroup CPU quota and by the GPU box's share of
    16 per GPU (sched_getaffinity reports all 256 host threads there)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("ARIA_CPU_BASELINE_THREADS", "16"))))


KERNEL_SOURCES = {"k_fast_blur_stream": "fast_blur_stream.hip", "k_fast_blur_band": "fast_blur_band.hip"}


def kernel_source_sha256(kernel):
    """sha256 of the source file of the FAST/blur kernel (what tools/pmc_traffic.py stores next to its counters)."""
    import hashlib
    f = os.path.join(ROOT, "aria_slam_amd", "csrc", KERNEL_SOURCES.get(kernel, ""))
    return hashlib.sha256(open(f, "rb").read()).hexdigest() if os.path.isfile(f) else None


def load_traffic(chunk, kernel="k_fast_blur_band"):
    """HBM bytes per frame of the FAST/blur kernel from the committed PMC passes (tools/pmc_traffic.sh), the file's name,
    and whether the file was measured on ANOTHER build of the kernel (its stored source hash differs from the tree's, or it
    stores none): a constant read from profiles/, not a measurement of this run -- `traffic_stale` says when not to trust it."""
    for name in ("pmc_traffic_r4.json", "pmc_traffic_r3.json", "pmc_traffic_%d.json" % chunk, "pmc_traffic_1024.json"):
        f = os.path.join(ROOT, "profiles", name)
        if os.path.exists(f):
            j = json.load(open(f))
            k = j["kernels"].get(kernel)
            if k:
                have = (j.get("kernel_source_sha256") or {}).get(kernel)
                stale = have is None or have != kernel_source_sha256(kernel)
                return k["fetch_bytes_per_frame"] + k["write_bytes_per_frame"], name, stale
    return None, None, None


def cpu_baseline(width, height, nfeatures, budget_s=12.0):
    """Time the CPU oracle (a port of the reference's CPU OpenCV path, oracle/orb_oracle.cpp) on this host:
    frame-parallel, one worker thread per core, each unit = 1 extract + 1 match (same accounting as the GPU)."""
    import threading
    import aria_slam_amd as A
    from oracle import oracle_py as O
    O.build()
    O.lib()
    cores = host_cores()
    p = O.default_params(nfeatures)
    done = [0] * cores
    t_end = time.time() + budget_s

    def work(t):
        seed = 1 + t
        a, b = A.synth_frame_pair(seed, width, height)
        _, prev = O.orb_extract(a, p)
        cur_img = b
        while True:
            _, cur = O.orb_extract(cur_img, p)
            O.match_ratio(cur, prev, 0.75)
            done[t] += 1
            prev = cur
            if time.time() >= t_end:
                break
            seed += cores
            a, b = A.synth_frame_pair(seed, width, height)
            cur_img = a if (done[t] & 1) else b

    t0 = time.time()
    th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.time() - t0
    n = sum(done)
    return {"value": round(n / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frames (1 extract + 1 match each) of the same synthetic %dx%d/%d-kp workload in %.1f s, "
                      "%d threads; port = oracle/orb_oracle.cpp (scalar C++ restatement of OpenCV CPU ORB + BFMatcher)"
                      % (n, width, height, nfeatures, dt, cores)}


# ---------------------------------------------------------------------------------------------------------------
# N-rank launcher (parent process: no torch import, no GPU call)
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Start n fresh child processes of this script, one rank per GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set as
    torch.distributed.run would), relay rank 0's JSON line, exit non-zero if any rank fails. The parent makes no GPU
    call at all, so nothing that initialised the GPU is ever exec'ed or forked."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = None
    for ln in (out0 or b"").decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if any(codes) or line is None:
        print("bench.py: rank exit codes %s%s" % (codes, "" if line else "; rank 0 printed no result line"), file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


# ---------------------------------------------------------------------------------------------------------------
# Verification helpers (untimed)
# ---------------------------------------------------------------------------------------------------------------
def frame_digest(count, kps_rows, desc_rows):
    """sha256 over (count, keypoint records, descriptor rows) of one frame -- tests/golden/make_golden.py writes the
    same digest from the oracle."""
    h = hashlib.sha256()
    h.update(np.int32(count).tobytes())
    h.update(np.ascontiguousarray(kps_rows[:count]).tobytes())
    h.update(np.ascontiguousarray(desc_rows[:count]).tobytes())
    return h.hexdigest()


def match_digest(nm, match_rows):
    h = hashlib.sha256()
    h.update(np.int32(nm).tobytes())
    h.update(np.ascontiguousarray(match_rows[:nm]).tobytes())
    return h.hexdigest()


def verify_last_step(A, torch, dev, images, last, halo_img, B, W, H, NF, cap, ratio, rank, seed0):
    """Check what the timed region produced (the one-chunk, two-stream batch path), untimed:
    (1) every frame's keypoints/descriptors/matches are hashed (checksum of checksums);
    (2) rank 0's first frames are compared with the committed ORACLE digests (tests/golden/bench_prefix.json, written
        by tests/golden/make_golden.py) when the workload is the one they were made for;
    (3) sampled frames (first, second, middle, last, a few in between) are recomputed through the single-frame
        host-buffer entry points (aria_orb_extract / aria_matcher_match -- the path the parity tests pin against the
        oracle) and must be byte-identical, including the (frame 0, halo frame) pair."""
    kps = last["kps"].cpu().numpy()
    desc = last["desc"].cpu().numpy()
    cnt = last["counts"].cpu().numpy()
    mts = last["matches"].cpu().numpy()
    nms = last["nmatches"].cpu().numpy()
    fd = [frame_digest(int(cnt[f]), kps[f], desc[f]) for f in range(B)]
    md = [match_digest(int(nms[f]), mts[f]) for f in range(B)]
    allh = hashlib.sha256()
    for f in range(B):
        allh.update(bytes.fromhex(fd[f]))
        allh.update(bytes.fromhex(md[f]))
    res = {"checksum_of_checksums": allh.hexdigest(), "frames_hashed": B, "problems": []}

    # (2) oracle prefix
    res["oracle_prefix_frames"] = 0
    if rank == 0 and seed0 == 1 and os.path.exists(GOLDEN_PREFIX):
        g = json.load(open(GOLDEN_PREFIX))
        key = "%dx%d_n%d" % (W, H, NF)
        if key in g and abs(g[key]["ratio"] - ratio) < 1e-9:
            n = min(B, g[key]["frames"])
            bad = [f for f in range(n) if fd[f][:len(g[key]["frame"][f])] != g[key]["frame"][f]]
            badm = [f for f in range(1, n) if md[f][:len(g[key]["match"][f])] != g[key]["match"][f]]
            res["oracle_prefix_frames"] = n
            if bad:
                res["problems"].append("frames %s differ from the oracle digests" % sorted(set(bad))[:8])
            if badm:
                res["problems"].append("matches of pairs (f, f-1), f in %s differ from the oracle digests" % sorted(set(badm))[:8])
            full = g[key].get("full", {}).get(str(B))
            if full is not None:
                res["full_checksum_matches_committed"] = (full == res["checksum_of_checksums"])
                if full != res["checksum_of_checksums"]:
                    res["problems"].append("checksum of checksums differs from the committed one")

    # (3) sampled frames through the single-frame host path
    rng = np.random.default_rng(12345 + rank)
    sample = sorted(set([0, 1, B // 2 - 1, B // 2, B - 2, B - 1] + rng.integers(0, B, 10).tolist()) & set(range(B)))
    e1 = A.OrbHipExtractor(max_features=NF, device=dev.index, max_width=W, max_height=H, max_batch=1)
    m1 = A.HipMatcher(device=dev.index, max_query=cap, max_train=cap)
    try:
        memo = {}

        def single(f):
            if f not in memo:
                img = halo_img if f < 0 else images[f].cpu().numpy()
                memo[f] = e1.extract(img)
            return memo[f]
        for f in sample:
            fr = single(f)
            n = len(fr["keypoints"])
            if n != int(cnt[f]) or fr["keypoints"].tobytes() != kps[f, :n].tobytes() or \
                    fr["descriptors"].tobytes() != desc[f, :n].tobytes():
                res["problems"].append("frame %d: batch result differs from the single-frame path" % f)
                continue
            prev = single(f - 1)          # f == 0: the halo frame
            want = m1.match(fr, prev, None, ratio)
            if len(want) != int(nms[f]) or want.tobytes() != mts[f, :len(want)].tobytes():
                res["problems"].append("pair (%d, %d): batch matches differ from the single-pair path" % (f, f - 1))
        res["sampled_frames"] = sample
    finally:
        e1.close()
        m1.close()
    res["ok"] = not res["problems"]
    return res


def single_frame_latency(A, dev, images, W, H, NF, n=48):
    """What a frame-at-a-time caller of IFeatureExtractor::extract / IMatcher::match sees (one frame in flight, pageable host
    buffers in and out, PCIe both ways, sync per call): the host entry points replay a hipGraph of a latency schedule.
    Three figures: extract alone; extract + match with the descriptors handed over on the device (getGpuDescriptors /
    matchGpu, what aria::pipeline::FrontEnd does with the HIP adapters: the match is queued behind extractAsync on a shared
    stream, one wait per frame); extract + match through host buffers only (the plain port calls).
    Never `value`: reported beside it."""
    import ctypes as C
    import torch
    n = min(n, images.shape[0])
    seq = images[:n].cpu().numpy()
    stream = torch.cuda.Stream(device=dev)
    e = A.OrbHipExtractor(max_features=NF, device=dev.index, max_width=W, max_height=H)
    m = A.HipMatcher(device=dev.index)
    e2 = A.OrbHipExtractor(max_features=NF, device=dev.index, max_width=W, max_height=H, stream=stream.cuda_stream)
    m2 = A.HipMatcher(device=dev.index, stream=stream.cuda_stream)
    try:
        L = e._L
        cap = e.kp_capacity()
        kp = [np.empty(cap, A.KP_DTYPE) for _ in range(2)]
        ds = [np.empty((cap, 32), np.uint8) for _ in range(2)]
        mt = np.empty(cap, A.MATCH_DTYPE)
        cnt = [C.c_int(), C.c_int()]
        nm = C.c_int()

        def ext(i, slot, h=None):
            rc = L.aria_orb_extract((h or e)._h, seq[i].ctypes.data, W, H, W, kp[slot].ctypes.data, ds[slot].ctypes.data, cap, C.byref(cnt[slot]))
            assert rc == 0, rc
        for i in range(min(8, n)):
            ext(i, 0)
        reps = 4
        t0 = time.perf_counter()
        for _ in range(reps):
            for i in range(n):
                ext(i, 0)
        t1 = time.perf_counter()
        ext(0, 1)
        for _ in range(reps):
            for i in range(n):
                s = i & 1
                ext(i, s)
                rc = L.aria_matcher_match(m._h, ds[s].ctypes.data, cnt[s].value, ds[1 - s].ctypes.data, cnt[1 - s].value,
                                          C.c_float(0.75), mt.ctypes.data, cap, C.byref(nm))
                assert rc == 0, rc
        t2 = time.perf_counter()
        host_matches = mt[:nm.value].copy()
        # device hand-off: extractAsync, the match queued behind it on the same stream, ONE wait, then the results
        ext(0, 1, e2)
        _, d_desc, d_cnt, n0, rows = e2.device_result()
        m2.retain_device(d_desc, n0)
        dev_matches = None
        for rep in range(reps + 1):
            if rep == 1:
                t3 = time.perf_counter()
            for i in range(n):
                s = i & 1
                rc = L.aria_orb_extract_async(e2._h, seq[i].ctypes.data, W, H, W)
                assert rc == 0, rc
                rc = L.aria_matcher_match_device_async(m2._h, d_desc, d_cnt, rows, 1, C.c_float(0.75))
                assert rc == 0, rc
                rc = L.aria_orb_sync(e2._h, kp[s].ctypes.data, ds[s].ctypes.data, cap, C.byref(cnt[s]))
                assert rc == 0, rc
                rc = L.aria_matcher_finish(m2._h, cnt[s].value, mt.ctypes.data, cap, C.byref(nm))
                assert rc == 0, rc
        t4 = time.perf_counter()
        dev_matches = mt[:nm.value].copy()
        assert dev_matches.tobytes() == host_matches.tobytes(), "device hand-off and host-buffer matches differ"
        return {"what": "aria_orb_extract(+ match), one frame in flight, pageable host buffers, PCIe-inclusive, never `value`. "
                        "extract_plus_match_us: descriptors handed over on the device (aria_orb_last_device + "
                        "aria_matcher_match_device_async queued behind aria_orb_extract_async on one stream, one wait per "
                        "frame) = what aria::pipeline::FrontEnd runs; extract_plus_match_host_buffers_us: the two port "
                        "calls through host buffers, a wait each",
                "extract_us": round(1e6 * (t1 - t0) / (reps * n), 1),
                "extract_plus_match_us": round(1e6 * (t4 - t3) / (reps * n), 1),
                "extract_plus_match_host_buffers_us": round(1e6 * (t2 - t1) / (reps * n), 1), "frames": reps * n}
    finally:
        e.close()
        m.close()
        e2.close()
        m2.close()


def loop_closure_leg(A, torch, dist, dev, mat, last, B, cap, rank, world, rehearsal, lo_frame):
    """BASELINE.json configs[4]'s exchange, once, untimed by the headline clock: every rank turns evenly spaced frames
    of its shard into keyframes (500 node-wide, the reference's database cap, src/legacy/LoopClosure.cpp:28-30),
    the padded descriptor slots are all-gathered (RCCL over xGMI; gloo via host memory in rehearsal mode), and each
    rank scans the whole database for its newest frame (LoopClosure.cpp:72-114 with euroc_eval's min_frames_between
    = 200, src/euroc_eval.cpp:103)."""
    from aria_slam_amd import loopdb
    k_local = max(1, loopdb.MAX_KEYFRAMES // world)
    stride = max(1, B // k_local)
    db = loopdb.KeyframeDB(k_local, cap, dev)
    counts_host = last["counts"].cpu().numpy()
    for k in range(k_local):
        f = min(k * stride, B - 1)
        db.add(lo_frame + f, last["desc"][f], int(counts_host[f]))
    torch.cuda.synchronize(dev)
    gather_s = None
    if dist is not None:
        db.all_gather(via_host=rehearsal)          # warm-up: the first collective of a size pays RCCL's channel setup
        torch.cuda.synchronize(dev)
        times = []
        for _ in range(3):
            dist.barrier()
            t0 = time.perf_counter()
            g = db.all_gather(via_host=rehearsal)
            torch.cuda.synchronize(dev)
            times.append(time.perf_counter() - t0)
        gather_s = sorted(times)
    else:
        g = db
    t_scan0 = time.perf_counter()
    q = B - 1
    cands = g.find_candidates(mat, last["desc"][q], int(counts_host[q]), lo_frame + q, 200, 0.7)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    scan_s = t2 - t_scan0
    ids = g.ids.cpu().numpy()
    out = {"keyframes": int((g.counts > 0).sum().item()), "slot_rows": cap, "db_bytes": int(g.desc.numel()),
           "allgather_ms": round(1e3 * gather_s[1], 3) if gather_s else None,
           "allgather_ms_all": [round(1e3 * t, 3) for t in gather_s] if gather_s else None,
           "allgather_backend": (None if dist is None else ("gloo via host (rehearsal)" if rehearsal else "nccl (RCCL)")),
           "scan_ms": round(1e3 * scan_s, 3),
           "query_frame": int(lo_frame + q),
           "candidates": [[int(ids[i]), round(float(s), 6)] for i, s in cands]}
    if dist is not None:
        # bytes each rank RECEIVES (world - 1 slices of the padded database) over the median time, against the one-shot bound of
        # on a fully connected xGMI node every peer pushes its slice over its own link (~153 GB/s each), so a
        # rank can take in (world - 1) x 153 GB/s; a ring would be bound by ONE link. Reported, not asserted (message sizes of a
        # few MB per rank are latency-dominated).
        recv = g.desc.numel() * (world - 1) / world
        out["allgather_GBs"] = round(recv / max(gather_s[1], 1e-9) / 1e9, 2)
        out["allgather_slice_bytes"] = int(g.desc.numel() // world)
        out["xgmi_one_shot_bound_GBs"] = round(153.0 * (world - 1), 1)
        out["allgather_frac_of_bound"] = round(out["allgather_GBs"] / max(out["xgmi_one_shot_bound_GBs"], 1e-9), 4)
        if rehearsal:
            out["allgather_note"] = "rehearsal: gloo through host memory on one GPU -- not an xGMI measurement"
    return out


def streamed_leg(torch, dev, ext, mat, host_pinned, B, W, H, cap, sets, last, se, sm, args, halo_desc, halo_cnt, chunk_frames=1024):
    """h2d_staging.streamed_*: one step whose frames are NOT resident -- the shard comes from pinned host memory in chunks of
    `chunk_frames` on a copy stream (two device buffers: chunk c + 1 is copied while chunk c is extracted), the matcher runs
    on its stream once the last chunk is described. Untimed by `value`; run three times, the median is reported. The results
    must equal the resident step's (same frames)."""
    C = min(chunk_frames, B)
    sc = torch.cuda.Stream(device=dev)
    bufs = [torch.empty((C, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]
    d = sets[0] if (len(sets) == 1 or sets[0] is not last) else sets[1]
    dstride = cap * 32

    def run():
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for c, lo in enumerate(range(0, B, C)):
            n, s = min(C, B - lo), c & 1
            with torch.cuda.stream(sc):
                if c >= 2:
                    sc.wait_event(free[s])
                bufs[s][:n].copy_(host_pinned[lo:lo + n], non_blocking=True)
                ready[s].record(sc)
            with torch.cuda.stream(se):
                se.wait_event(ready[s])
                ext.extract_batch_device(bufs[s].data_ptr(), n, W, H, d["kps"].data_ptr() + lo * cap * 24,
                                         d["desc"].data_ptr() + lo * cap * 32, d["counts"].data_ptr() + 4 * lo, cap)
                free[s].record(se)
        done = torch.cuda.Event()
        done.record(se)
        with torch.cuda.stream(sm):
            sm.wait_event(done)
            mat.match_batch_device(d["desc"], d["counts"], halo_desc, halo_cnt, 1, dstride, args.ratio, d["matches"], d["nmatches"], cap)
            mat.match_batch_device(d["desc"].data_ptr() + dstride, d["counts"].data_ptr() + 4, d["desc"], d["counts"], B - 1,
                                   dstride, args.ratio, d["matches"].data_ptr() + cap * 12, d["nmatches"].data_ptr() + 4, cap)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    run()                                           # warm-up (buffers touched, streams created)
    ts = sorted(run() for _ in range(3))
    ext.check()
    mat.sync()
    same = None
    if d is not last:
        nl = last["counts"].cpu().numpy()
        same = bool(torch.equal(d["counts"], last["counts"]) and torch.equal(d["nmatches"], last["nmatches"]))
        if same:                                     # rows beyond a frame's count are undefined: compare the defined ones
            import numpy as _np
            idx = _np.linspace(0, B - 1, 64).astype(int)
            for f in idx:
                n = int(nl[f])
                same = same and bool(torch.equal(d["desc"][f, :n], last["desc"][f, :n])) and bool(torch.equal(d["kps"][f, :n], last["kps"][f, :n]))
    return {"what": "one step with the frames streamed from pinned host memory in chunks on a copy stream beside the extraction "
                    "(double-buffered), matcher behind the last chunk; median of 3 runs after a warm-up; never `value`",
            "chunk_frames": C, "frames": B, "seconds": round(ts[1], 4), "frames_per_s": round(B / ts[1], 1),
            "frames_per_s_all": [round(B / t, 1) for t in ts], "h2d_GBs_sustained": round(B * W * H / ts[1] / 1e9, 2),
            "equals_resident_step": same}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=4096, help="frame pairs per GPU (sequence length = 2*pairs)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--chunk", type=int, default=0,
                    help="frames per internal extractor pass (scratch is sized for it: ~3.5 MB per 640x480 frame); "
                         "0 = the whole per-GPU sequence, at most 8192 -- with 288 GB of HBM one pass per step is the "
                         "cheapest (1024: 216k, 2048: 222k, 4096: 225k, 8192: 230k frames/s)")
    ap.add_argument("--ratio", type=float, default=0.75)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed verification of the last step")
    ap.add_argument("--no-loop-closure", action="store_true", help="skip the untimed keyframe-DB exchange + scan leg")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false",
                    help="extractor and matcher on one stream (default: matcher of step s beside the extractor of step s+1)")
    ap.add_argument("--match-beside", choices=["band", "select"], default="band",
                    help="two-stream pipeline: the matcher of step s runs beside the FAST/blur launches of step s+1 (band: it "
                         "starts as soon as step s is described) or beside select + describe of step s+1 (select: it also "
                         "waits for the event the extractor records before its select stage, aria_orb_set_stage_event)")
    ap.add_argument("--lanes", type=int, default=1,
                    help="extractor lanes: the per-GPU sequence is cut into this many contiguous parts, each with its own "
                         "extractor handle and stream, so that the VALU-bound FAST/blur launches of one part run beside the "
                         "memory-bound select/describe launches of another (chunk-level pipelining; results unchanged)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # we are the parent: start the ranks, relay rank 0's line. Nothing below runs in this process.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size is used" % (args.gpus, world), file=sys.stderr)

    if os.environ.get("ARIA_BENCH_PROBE") == "1":
        # launcher self-test (tests/test_host_logic.py, no GPU): the ranks rendezvous over gloo, agree on the world
        # size and rank 0 prints a line of the usual shape. Nothing is measured.
        import torch
        import torch.distributed as dist
        from aria_slam_amd import shard
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([1, rank], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t)
        lo, hi = shard.frame_range(world * 2 * args.pairs, rank, world)
        if rank == 0:
            print(json.dumps({"metric": "probe", "n_gpus": world, "ranks_seen": int(t[0]), "rank_sum": int(t[1]),
                              "rank0_frames": [lo, hi], "value": None}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    import torch
    import aria_slam_amd as A
    from aria_slam_amd import shard

    # ARIA_BENCH_REHEARSAL=1: run the N>1 code path on a box with fewer GPUs than ranks (every rank on cuda:0, gloo
    # for the barrier, the clock and the all-gather). The numbers of such a run mean nothing; it only proves the
    # multi-rank path executes, and the result line says "rehearsal": true.
    rehearsal = os.environ.get("ARIA_BENCH_REHEARSAL", "0") == "1"
    if world > 1 and not rehearsal and torch.cuda.device_count() < world:
        print("bench.py: %d ranks but %d GPU(s) visible; set ARIA_BENCH_REHEARSAL=1 to rehearse the multi-rank path "
              "on one GPU (numbers invalid)" % (world, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)
    dev_index = 0 if (world == 1 or rehearsal) else local_rank
    rccl_ranks = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
            one = torch.ones(1, dtype=torch.int32, device=torch.device("cuda", dev_index))
            dist.all_reduce(one)                      # RCCL really connects world ranks
            rccl_ranks = int(one.item())
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    n_gpus = world

    A.load_library()
    W, H, NF = args.width, args.height, args.features
    B = 2 * args.pairs
    if args.chunk <= 0:
        args.chunk = min(B, 8192)

    # ---- synthetic sequence shard of this rank, resident in HBM before timing ----
    # node-wide sequence: world * B frames = pairs seeds 1 .. world*pairs; rank r owns the contiguous frame range
    # shard.frame_range gives it (B frames = pairs seeds 1 + r*pairs ..)
    lo_frame, hi_frame = shard.frame_range(world * B, rank, world)
    assert hi_frame - lo_frame == B and lo_frame % 2 == 0
    seed0 = 1 + lo_frame // 2
    host = torch.empty((B, H, W), dtype=torch.uint8, pin_memory=False)
    A.synth_sequence(seed0, args.pairs, W, H, out=host.numpy())
    # H2D staging of the shard, reported separately and never part of `value`: the upload from
    # pageable memory as a sequence reader would hold it, and from pinned memory (what a reader that decodes into registered
    # buffers gets -- aria_slam_amd/host BatchFrontEnd does). Each figure is the MEDIAN of three copies after a warm-up copy
    # (round 3 timed one cold copy per path: first-touch and registration cost, not a rate). Scaling is linear in the GPUs
    # as long as these rates exceed the compute rate; the `streamed` leg further down feeds the extractor that way.
    def timed_copy(dst, src, non_blocking):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        dst.copy_(src, non_blocking=non_blocking)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t

    images = host.to(dev)                                   # warm-up copy: the shard becomes resident
    t_page = sorted(timed_copy(images, host, False) for _ in range(3))
    staging = {"bytes": int(host.numel()), "pageable_s": round(t_page[1], 4), "pageable_GBs": round(host.numel() / t_page[1] / 1e9, 2),
               "pageable_GBs_all": [round(host.numel() / t / 1e9, 2) for t in t_page], "copies": "median of 3 after a warm-up copy"}
    host_pinned = None
    try:
        host_pinned = torch.empty((B, H, W), dtype=torch.uint8, pin_memory=True)
        host_pinned.copy_(host)
        timed_copy(images, host_pinned, True)               # warm-up
        t_pin = sorted(timed_copy(images, host_pinned, True) for _ in range(3))
        staging.update({"pinned_bytes": int(host_pinned.numel()), "pinned_GBs": round(host_pinned.numel() / t_pin[1] / 1e9, 2),
                        "pinned_GBs_all": [round(host_pinned.numel() / t / 1e9, 2) for t in t_pin]})
    except RuntimeError as ex:          # pinning can be refused (memory limits): the pageable figure stands alone
        staging["pinned_error"] = str(ex)[:120]
        host_pinned = None
    # halo frame (shard.shard_plan: the frame before this rank's range, recomputed rather than exchanged); rank 0 has
    # no predecessor and wraps around to its own last frame
    if rank > 0:
        halo_img = A.synth_frame_pair(seed0 - 1, W, H)[1]
    else:
        halo_img = host[B - 1].numpy().copy()
    del host

    # Streams. torch's default stream has handle 0, which both C-ABI handles read as "no stream given, create your
    # own": never pass it. The extractor runs on stream E, the matcher on stream M, with two sets of output buffers:
    # M matches step s (it waits for E's "set s&1 described" event) while E already extracts step s+1 into the other
    # set (it waits for M's "set matched" event of two steps ago before overwriting). k_fast_blur_band is VALU-issue
    # bound and k_knn2_mfma lives on the matrix pipe, so the two overlap well (+6 %, tools/overlap_probe.py; same
    # results). --no-pipeline puts both handles on one stream.
    se = torch.cuda.Stream(device=dev)
    sm = torch.cuda.Stream(device=dev) if args.pipeline else se
    torch.cuda.set_stream(se)
    assert se.cuda_stream != 0 and sm.cuda_stream != 0
    lanes = max(1, min(args.lanes, B)) if args.pipeline else 1
    lane_lo = [(B * k) // lanes for k in range(lanes + 1)]
    lane_streams = [se] + [torch.cuda.Stream(device=dev) for _ in range(lanes - 1)]
    if lanes > 1:
        args.chunk = min(args.chunk, max(lane_lo[k + 1] - lane_lo[k] for k in range(lanes)))
    ext = A.OrbHipExtractor(max_features=NF, stream=se.cuda_stream, device=dev.index, max_width=W, max_height=H,
                            max_batch=args.chunk)
    exts = [ext] + [A.OrbHipExtractor(max_features=NF, stream=lane_streams[k].cuda_stream, device=dev.index, max_width=W,
                                      max_height=H, max_batch=args.chunk) for k in range(1, lanes)]
    mat = A.HipMatcher(stream=sm.cuda_stream, device=dev.index, max_query=4096, max_train=4096)
    cap = ext.kp_capacity()
    dstride = cap * 32
    sets = []
    for _ in range(2 if args.pipeline else 1):
        sets.append(dict(kps=torch.empty((B, cap, 24), dtype=torch.uint8, device=dev),
                         desc=torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
                         counts=torch.zeros((B,), dtype=torch.int32, device=dev),
                         matches=torch.empty((B, cap, 12), dtype=torch.uint8, device=dev),
                         nmatches=torch.zeros((B,), dtype=torch.int32, device=dev),
                         described=[torch.cuda.Event() for _ in range(lanes)], matched=torch.cuda.Event()))
    halo_desc = torch.zeros((cap, 32), dtype=torch.uint8, device=dev)
    halo_cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    step_no = [0]
    beside_select = args.pipeline and args.match_beside == "select"
    mid_ev = torch.cuda.Event()
    pending = [None]              # --match-beside select: the set whose matching waits for the next extraction's mid event
    if beside_select:
        mid_ev.record(se)
        ext.set_stage_event("select", mid_ev.cuda_event)

    def enqueue_match(d, wait_mid=False):
        with torch.cuda.stream(sm):
            if args.pipeline:
                for ev in d["described"]:
                    sm.wait_event(ev)
                if wait_mid:
                    sm.wait_event(mid_ev)
            mat.match_batch_device(d["desc"], d["counts"], halo_desc, halo_cnt, 1, dstride, args.ratio, d["matches"],
                                   d["nmatches"], cap)
            mat.match_batch_device(d["desc"].data_ptr() + dstride, d["counts"].data_ptr() + 4, d["desc"], d["counts"], B - 1,
                                   dstride, args.ratio, d["matches"].data_ptr() + cap * 12, d["nmatches"].data_ptr() + 4, cap)
            d["matched"].record(sm)

    def flush_pending():
        if pending[0] is not None:
            enqueue_match(pending[0])
            pending[0] = None

    def step(serialise=False):
        d = sets[step_no[0] % len(sets)]
        step_no[0] += 1
        # lanes: the un-bracketed lanes are enqueued first -- the bracketed lane 0 blocks the host while its FAST/blur
        # launches run, and the other lanes' work has to be in their queues by then
        # (the extra, fully bracketed step runs the whole sequence through lane 0's handle alone)
        for k in ([0] if serialise else list(range(1, lanes)) + [0]):
            lo, n = (0, B) if serialise else (lane_lo[k], lane_lo[k + 1] - lane_lo[k])
            sk = lane_streams[k]
            with torch.cuda.stream(sk):
                if args.pipeline:
                    sk.wait_event(d["matched"])       # recorded two steps ago (a no-op before that)
                exts[k].extract_batch_device(images.data_ptr() + lo * W * H, n, W, H, d["kps"].data_ptr() + lo * cap * 24,
                                             d["desc"].data_ptr() + lo * cap * 32, d["counts"].data_ptr() + 4 * lo, cap)
                d["described"][k].record(sk)
        if serialise:
            torch.cuda.synchronize(dev)
            for ev in d["described"][1:]:
                ev.record(se)
        if beside_select and not serialise:
            # this step's extraction is queued (its mid event recorded inside the call): now the PREVIOUS step's matching,
            # held back until this extraction is past its FAST/blur launches
            if pending[0] is not None:
                enqueue_match(pending[0], wait_mid=True)
            pending[0] = d
        else:
            enqueue_match(d)
        if serialise:
            torch.cuda.synchronize(dev)
        return d

    # halo frame: extracted once, outside the timed region (1 frame per shard; shard.py)
    with torch.cuda.stream(se):
        himg = torch.from_numpy(halo_img).to(dev)
        hk = torch.empty((1, cap, 24), dtype=torch.uint8, device=dev)
        ext.extract_batch_device(himg, 1, W, H, hk, halo_desc, halo_cnt, cap)
    torch.cuda.synchronize(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    flush_pending()
    barrier()
    for e_ in exts:
        e_.check()
    mat.sync()
    # Timed region: HIP-event brackets only around what the roofline objects report (the dominant kernel and the
    # matcher's kNN-2); every bracket drains the stream twice. The full per-stage table comes from one extra,
    # untimed step afterwards.
    ext.set_profiling(True, stages=["fast_blur"])
    # the matcher is bracketed in the timed region only when it shares the extractor's stream: a bracket blocks the host
    # until its stream is drained, and the host has to be free to enqueue the next step's extraction beside the matcher
    mat.set_profiling(not args.pipeline, stages=["knn2"])
    ext.get_profile(reset=True)
    mat.get_profile(reset=True)

    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    flush_pending()               # (the last step's matching runs alone in either schedule)
    torch.cuda.synchronize(dev)
    dt_local = time.perf_counter() - t0          # this rank's own clock: start barrier -> its last kernel done
    barrier()
    dt = time.perf_counter() - t0
    for e_ in exts:
        e_.check()
    mat.sync()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=("cpu" if rehearsal else dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    slow_blocks = sum(e_.slow_path_blocks(reset=True) for e_ in exts)      # band-kernel workgroups that fell back to dense rescoring
    prof_e, prof_frames = ext.get_profile(reset=True)
    prof_m, prof_pairs = mat.get_profile(reset=True)
    ext.set_profiling(False)
    mat.set_profiling(False)
    cnt_host = last["counts"].cpu().numpy()
    nm_host = last["nmatches"].cpu().numpy()

    # ---- untimed: verify what the timed region produced (every rank verifies its own shard) ----
    verification = None
    if not args.no_verify and args.steps > 0:
        verification = verify_last_step(A, torch, dev, images, last, halo_img, B, W, H, NF, cap, args.ratio, rank, seed0)
    all_ok = True if verification is None else verification["ok"]
    if dist is not None:
        okt = torch.tensor([1 if all_ok else 0], dtype=torch.int32, device=("cpu" if rehearsal else dev))
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        all_ok = bool(int(okt.item()))

    # per-rank figures for the N > 1 line: a straggler rank must be visible when the curve is run
    per_rank = None
    if dist is not None:
        mine = torch.tensor([dt_local, 1.0 if (verification is None or verification["ok"]) else 0.0, float(B * args.steps)],
                            dtype=torch.float64, device=("cpu" if rehearsal else dev))
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(x) for x in t.cpu().tolist()] for t in allr]

    # ---- untimed: loop-closure exchange + scan (configs[4]) ----
    loop_closure = None
    if not args.no_loop_closure and args.steps > 0:
        try:
            loop_closure = loop_closure_leg(A, torch, dist, dev, mat, last, B, cap, rank, world, rehearsal, lo_frame)
        except Exception as e:          # reported, never hidden; the headline number does not depend on it
            loop_closure = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- untimed: the frame-at-a-time host path (second, clearly labelled figure) ----
    single = None
    if rank == 0 and not args.no_verify and args.steps > 0:
        try:
            single = single_frame_latency(A, dev, images, W, H, NF)
        except Exception as e:
            single = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- untimed: the shard fed from pinned host memory in chunks on a copy stream, beside the extraction ----
    streamed = None
    if host_pinned is not None and args.steps > 0 and lanes == 1:
        try:
            streamed = streamed_leg(torch, dev, ext, mat, host_pinned, B, W, H, cap, sets, last, se, sm, args, halo_desc, halo_cnt)
        except Exception as e:
            streamed = {"error": "%s: %s" % (type(e).__name__, e)}
    del host_pinned

    # extra untimed step with every stage bracketed, extractor and matcher one after the other -> stage_us_per_frame
    ext.set_profiling(True)
    mat.set_profiling(True)
    ext.get_profile(reset=True)
    mat.get_profile(reset=True)
    step(serialise=True)
    torch.cuda.synchronize(dev)
    prof_e_all, frames_all = ext.get_profile(reset=True)
    prof_m_all, pairs_all = mat.get_profile(reset=True)

    if rank == 0:
        frames_total = B * args.steps * n_gpus
        value = frames_total / dt
        # ---- roofline of the dominant kernel, from HIP events on the launch stream ----
        stage_ms = {k: v[0] for k, v in prof_e.items()}
        stage_ms.update({k: v[0] for k, v in prof_m.items()})
        launches = {k: v[1] for k, v in prof_e.items()}
        launches.update({k: v[1] for k, v in prof_m.items()})
        b_extract, b_fused = ext.algorithmic_bytes(W, H, NF)
        P = (b_fused - 56 * NF) // 2
        # algorithmic bytes per frame of each extractor kernel (DESIGN.md "Kernels")
        p0 = W * H
        lv = ext.level_info(W, H)
        p7 = lv[7][0] * lv[7][1]
        # pyramid fused into the FAST/blur launches (default): that kernel also writes the raw levels 1..7 (P - p0 bytes);
        # it reads each level once either way
        fused_pyramid = prof_e_all["resize"][1] == 0
        alg = {"resize": (P - p7) + (P - p0), "fast_blur": 2 * P + ((P - p0) if fused_pyramid else 0), "describe": 56 * NF}
        if os.environ.get("ARIA_BENCH_DEBUG"):
            print("timed:", prof_e, prof_frames, prof_m, prof_pairs, "extra:", prof_e_all, frames_all, prof_m_all, pairs_all, file=sys.stderr)
        all_ms = {k: v[0] for k, v in prof_e_all.items()}
        all_ms.update({k: v[0] for k, v in prof_m_all.items()})
        dom = max(("resize", "fast_blur", "select", "describe"), key=lambda k: all_ms[k])
        fb_ms_per_launch = stage_ms["fast_blur"] / max(launches["fast_blur"], 1)
        frames_per_launch = prof_frames / max(launches["fast_blur"], 1)
        fb_bytes_per_launch = alg["fast_blur"] * frames_per_launch
        achieved = fb_bytes_per_launch / (fb_ms_per_launch * 1e-3) / 1e9 if fb_ms_per_launch > 0 else 0.0
        fb_kernel = ext.fast_blur_kernel()
        traffic_pf, traffic_src, traffic_stale = load_traffic(args.chunk, fb_kernel)
        ext_ms = sum(all_ms[k] for k in ("resize", "fast_blur", "select", "describe")) * (prof_frames / max(frames_all, 1))
        pair_ops = 512.0 * float(cnt_host[1:].astype(np.float64) @ cnt_host[:-1].astype(np.float64))   # 2 * 256 * sum(nq * nt)
        if stage_ms.get("knn2"):
            knn_ms, knn_steps, knn_src = stage_ms["knn2"], args.steps, "timed region"
        else:
            knn_ms, knn_steps, knn_src = all_ms["knn2"], 1, "extra untimed step (the timed region does not bracket the matcher's stream)"
        # train sets up to 4096 rows (every SLAM frame) run on the FP4 matrix path: one E2M1 value per descriptor bit,
        # v_mfma_f32_32x32x64_f8f6f4; peak = the dense FP4 figure of MI355X_MICROARCH.md (10 PFLOP/s, 2 x the int8 / fp8 one)
        # which kernel did it is what the handle reports (aria_matcher_knn_kernel), not a guess from the counts: "a|b" = both
        # layouts launched behind the device-side gate, the FP4 one ran unless a train set exceeded 4096 rows
        knn_name = mat.knn_kernel()
        fp4 = knn_name == "k_knn2_fp4" or (knn_name == "k_knn2_fp4|k_knn2_mfma" and int(cnt_host.max()) <= 4096)
        mpeak = 10000.0 if fp4 else 5000.0
        matcher_roofline = {"kernel": ("k_knn2_fp4 (v_mfma_f32_32x32x64_f8f6f4 on E2M1 bit operands + per-lane top-2)" if fp4 else
                                       "k_knn2_mfma (v_mfma_i32_32x32x32_i8 + per-lane top-2)"), "bound": "mfma",
                            "achieved": round(pair_ops * knn_steps / (knn_ms * 1e-3) / 1e12, 1) if knn_ms else None,
                            "peak": mpeak, "unit": "TOP/s", "launched": knn_name,
                            "frac": round(pair_ops * knn_steps / (knn_ms * 1e-3) / 1e12 / mpeak, 4) if knn_ms else None,
                            "measured_in": knn_src,
                            # the matcher's HBM share is tiny (B_match = 32 (Nq + Nt) + 16 Nq per pair) and its
                            # work in the units of the popcount formulation is 8 dword xor+popcounts per descriptor pair
                            "hbm_frac": round((32.0 * float(cnt_host[1:].sum() + cnt_host[:-1].sum()) + 16.0 * float(cnt_host[1:].sum()))
                                              * knn_steps / (knn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if knn_ms else None,
                            "popcount_dword_equivalents_per_s": round(pair_ops / 64.0 * knn_steps / (knn_ms * 1e-3), 1) if knn_ms else None}
        # BASELINE.md section 3's figure: B_fused = 2P + 56N credited to the FAST/blur launches + k_describe together
        # (the "fused FAST+rBRIEF kernel" of the north star is these two: the per-level top-N selection sits between
        # them); times from the extra, fully bracketed step
        fused_us = 1e3 * (all_ms["fast_blur"] + all_ms["describe"]) / max(frames_all, 1)
        fused_gbs = b_fused / (fused_us * 1e-6) / 1e9 if fused_us > 0 else 0.0
        roofline = {
            "bound": "hbm", "kernel": fb_kernel + (" (FAST-9 + NMS + 7x7 Gaussian of one level + bilinear resize to the next level, "
                                                   "fused; 8 level launches per pass)" if fused_pyramid else
                                                   " (FAST-9 + NMS + 7x7 Gaussian of one level, fused; 8 level launches per pass)"),
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": (int(traffic_pf * frames_per_launch) if traffic_pf and (W, H, NF) == (640, 480, 2000) else None),
            "traffic_source": traffic_src,
            # the PMC file is a committed measurement, not a counter read of this run: stale = it was taken on another
            # build of the kernel's source file (tools/pmc_traffic.py stores the sha256 it measured)
            "traffic_stale": traffic_stale,
            "algorithmic_bytes_per_launch": int(fb_bytes_per_launch),
            "avg_launch_ms": round(fb_ms_per_launch, 4), "frames_per_launch": frames_per_launch,
            "dominant_stage": dom,
            "b_fused": {"definition": "BASELINE.md section 3: (2P + 56N) bytes per frame over the time of the FAST/blur "
                                      "launches + k_describe (extra bracketed step)",
                        "bytes_per_frame": b_fused, "us_per_frame": round(fused_us, 3),
                        "achieved_GBs": round(fused_gbs, 1), "frac": round(fused_gbs / HBM_PEAK_GBS, 4)},
            "whole_extractor": {"algorithmic_bytes_per_frame": b_extract,
                                "achieved_GBs": round(b_extract * prof_frames / (ext_ms * 1e-3) / 1e9, 1) if ext_ms else None,
                                "frac": round(b_extract * prof_frames / (ext_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ext_ms else None},
            "stage_us_per_frame": {k: round(1e3 * all_ms[k] / max(frames_all, 1), 3) for k in all_ms},
            "stage_us_per_frame_note": "every stage bracketed, one extra untimed step after the timed region; the timed "
                                       "region brackets only fast_blur (achieved / avg_launch_ms), plus knn2 with --no-pipeline",
            # kNN-2 runs on the matrix cores (knn2_mfma.hip): exact inner products of bit-widened descriptors (FP4 operands up
            # to 4096 train rows, int8 beyond). ops = 2 * 256 * sum(nq * nt) per launch
            "matcher": matcher_roofline,
        }
        default_cfg = (W, H, NF, args.pairs) == (640, 480, 2000, 4096)
        if default_cfg:
            workload = "BASELINE.json configs[2]"
        elif (W, H, NF) == (1408, 1408, 4000):
            workload = "BASELINE.json configs[3] (1408x1408 / 4000 kp), %d frames per GPU" % B
        elif (W, H, NF) == (752, 480, 1000):
            workload = "EuRoC-sized frames of BASELINE.json configs[0]/[4] (752x480 / 1000 kp), synthetic"
        else:
            workload = "custom size (not a BASELINE.json configuration)"
        out = {
            "metric": "frames/s ORB extract+BF-match, 640x480 @2000 kp" if (W, H, NF) == (640, 480, 2000)
                      else "frames/s ORB extract+BF-match, %dx%d @%d kp" % (W, H, NF),
            "value": round(value, 1), "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "verified": (bool(all_ok) if verification is not None else None),
            "config": {"workload": "%s: sequence of %d synthetic %dx%d frame pairs per GPU "
                                   "(%d frames), %d kp/frame, 1 extract + 1 kNN-2/ratio match per frame, inputs "
                                   "resident in HBM" % (workload, args.pairs, W, H, B, NF),
                       "frames_per_gpu_per_step": B, "chunk_frames": args.chunk, "ratio": args.ratio,
                       "parallelism": "frames sharded by contiguous range (aria_slam_amd/shard.py), %d rank(s), one process "
                                      "per GPU, no collective on the data path" % n_gpus,
                       "ranks_launched_by": "torch.distributed.run / external" if "TORCHELASTIC_RUN_ID" in os.environ
                                            else ("bench.py --gpus N (launch_ranks)" if world > 1 else "single process"),
                       "rccl_ranks": rccl_ranks,
                       "streams": ("extractor and matcher on two streams, double-buffered outputs: the matcher of step s "
                                   "runs beside " + ("select + describe" if beside_select else "the FAST/blur launches") +
                                   " of step s+1") if args.pipeline else "one stream",
                       "extractor_lanes": lanes,
                       "mean_keypoints_per_frame": round(float(cnt_host.mean()), 2),
                       "mean_matches_per_frame": round(float(nm_host.mean()), 2),
                       "slow_path_blocks": int(slow_blocks)},
            "roofline": roofline,
        }
        if per_rank is not None:
            # per-rank view of the same timed region: own clock (start barrier -> own last kernel done), own verification
            fps = [pr[2] / pr[0] for pr in per_rank]
            out["ranks"] = {"frames_per_s": [round(x, 1) for x in fps], "seconds": [round(pr[0], 5) for pr in per_rank],
                            "verified": [bool(pr[1] >= 0.5) for pr in per_rank],
                            "min_frames_per_s": round(min(fps), 1), "max_frames_per_s": round(max(fps), 1),
                            "slowest_rank": int(min(range(len(fps)), key=lambda i: fps[i])),
                            "note": "value = all ranks' frames / the max-over-ranks barrier-to-barrier time; these are each rank's own clock"}
            if not rehearsal:
                assert rccl_ranks == world, "RCCL all-reduce saw %s ranks, world size %d" % (rccl_ranks, world)
        if rehearsal:
            out["rehearsal"] = True
            out["config"]["rehearsal_note"] = "all ranks share cuda:0, gloo collectives: the numbers are not a measurement"
        if verification is not None:
            out["verification"] = {k: verification[k] for k in verification if k != "ok"}
            out["verification"]["all_ranks_ok"] = bool(all_ok)
        if loop_closure is not None:
            out["loop_closure"] = loop_closure
        if single is not None:
            out["single_frame_host_path"] = single
        staging["frames_per_s_at_pageable_rate"] = round(staging["pageable_GBs"] * 1e9 / (W * H), 1)
        if "pinned_GBs" in staging:
            staging["frames_per_s_at_pinned_rate"] = round(staging["pinned_GBs"] * 1e9 / (W * H), 1)
        if streamed is not None:
            staging["streamed"] = streamed
            if "frames_per_s" in streamed:
                staging["streamed_frames_per_s"] = streamed["frames_per_s"]
        staging["note"] = ("upload of this rank's %d frames before the timed region; not part of `value`. The extractor "
                           "consumes %.1f GB/s of frames at the measured rate" % (B, out["value"] / max(n_gpus, 1) * W * H / 1e9))
        out["h2d_staging"] = staging
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H, NF, args.cpu_budget)
        print(json.dumps(out), flush=True)

    for e_ in exts:
        e_.close()
    mat.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not all_ok:
        if verification is not None and verification["problems"]:
            print("bench.py rank %d: verification FAILED: %s" % (rank, verification["problems"]), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
