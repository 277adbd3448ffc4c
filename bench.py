#!/usr/bin/env python3
"""Headline benchmark: frames/s of ORB extract + brute-force Hamming match on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): a recorded sequence of 4096
synthetic 640x480 frame pairs = 8192 grayscale frames, 2000 keypoints per frame, resident in HBM before the
timed region. One "step" = one pass of the hot path over the whole sequence: every frame is extracted (pyramid,
FAST, Harris, IC angle, rBRIEF) and matched (kNN-2 + ratio 0.75) against the previous frame's descriptors
(query = current, train = previous; frame 0 matches the last frame of the previous pass). So every frame costs
1 extract + 1 match. Data: synthetic (SURVEY.md 8d generator, seeds 1..4096 per GPU shard).

Multi-GPU: one process per GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks clock);
the sequence shards by contiguous frame ranges, each rank owns 4096 pairs, no data-path collective -> weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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


def load_traffic(chunk):
    """HBM bytes per frame of k_fast_blur_band from the committed PMC passes (tools/pmc_traffic.sh), or None."""
    for name in ("pmc_traffic_%d.json" % chunk, "pmc_traffic_1024.json"):
        f = os.path.join(ROOT, "profiles", name)
        if os.path.exists(f):
            k = json.load(open(f))["kernels"].get("k_fast_blur_band")
            if k:
                return k["fetch_bytes_per_frame"] + k["write_bytes_per_frame"], name
    return None, None


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
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false",
                    help="extractor and matcher on one stream (default: matcher of step s beside the extractor of step s+1)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import aria_slam_amd as A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ARIA_BENCH_REHEARSAL=1: run the N>1 code path on a one-GPU box (every rank on cuda:0, gloo for the barrier and
    # the max-over-ranks clock). The numbers of such a run mean nothing; it only proves the multi-rank path executes.
    rehearsal = os.environ.get("ARIA_BENCH_REHEARSAL", "0") == "1"
    dev_index = 0 if (world == 1 or rehearsal) else local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
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
    host = torch.empty((B, H, W), dtype=torch.uint8, pin_memory=False)
    A.synth_sequence(1 + rank * args.pairs, args.pairs, W, H, out=host.numpy())
    images = host.to(dev)
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
    ext = A.OrbHipExtractor(max_features=NF, stream=se.cuda_stream, device=dev.index, max_width=W, max_height=H,
                            max_batch=args.chunk)
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
                         described=torch.cuda.Event(), matched=torch.cuda.Event()))
    # the frame before this rank's range (shard.py: the halo frame): its descriptors are the train set of frame 0's
    # match; extracted once, outside the timed region
    halo_desc = torch.zeros((cap, 32), dtype=torch.uint8, device=dev)
    halo_cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    step_no = [0]

    def step(serialise=False):
        d = sets[step_no[0] % len(sets)]
        step_no[0] += 1
        with torch.cuda.stream(se):
            if args.pipeline:
                se.wait_event(d["matched"])       # recorded two steps ago (a no-op before that)
            ext.extract_batch_device(images, B, W, H, d["kps"], d["desc"], d["counts"], cap)
            d["described"].record(se)
        if serialise:
            torch.cuda.synchronize(dev)
        with torch.cuda.stream(sm):
            if args.pipeline:
                sm.wait_event(d["described"])
            mat.match_batch_device(d["desc"], d["counts"], halo_desc, halo_cnt, 1, dstride, args.ratio, d["matches"],
                                   d["nmatches"], cap)
            mat.match_batch_device(d["desc"].data_ptr() + dstride, d["counts"].data_ptr() + 4, d["desc"], d["counts"], B - 1,
                                   dstride, args.ratio, d["matches"].data_ptr() + cap * 12, d["nmatches"].data_ptr() + 4, cap)
            d["matched"].record(sm)
        if serialise:
            torch.cuda.synchronize(dev)
        return d

    # halo frame = the last frame of the (identical) sequence
    with torch.cuda.stream(se):
        ext.extract_batch_device(images, B, W, H, sets[0]["kps"], sets[0]["desc"], sets[0]["counts"], cap)
        halo_desc.copy_(sets[0]["desc"][B - 1])
        halo_cnt.copy_(sets[0]["counts"][B - 1:B])
    torch.cuda.synchronize(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    ext.check()
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
    barrier()
    dt = time.perf_counter() - t0
    ext.check()
    mat.sync()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=("cpu" if rehearsal else dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    slow_blocks = ext.slow_path_blocks(reset=True)      # band-kernel workgroups that fell back to dense rescoring
    prof_e, prof_frames = ext.get_profile(reset=True)
    prof_m, prof_pairs = mat.get_profile(reset=True)
    cnt_host = last["counts"].cpu().numpy()
    nm_host = last["nmatches"].cpu().numpy()
    # extra untimed step with every stage bracketed, extractor and matcher one after the other -> stage_us_per_frame
    ext.set_profiling(True)
    mat.set_profiling(True)
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
        traffic_pf, traffic_src = load_traffic(args.chunk)
        ext_ms = sum(all_ms[k] for k in ("resize", "fast_blur", "select", "describe")) * (prof_frames / max(frames_all, 1))
        pair_ops = 512.0 * float(cnt_host[1:].astype(np.float64) @ cnt_host[:-1].astype(np.float64))   # 2 * 256 * sum(nq * nt)
        if stage_ms.get("knn2"):
            knn_ms, knn_steps, knn_src = stage_ms["knn2"], args.steps, "timed region"
        else:
            knn_ms, knn_steps, knn_src = all_ms["knn2"], 1, "extra untimed step (the timed region does not bracket the matcher's stream)"
        matcher_roofline = {"kernel": "k_knn2_mfma (v_mfma_i32_32x32x32_i8 + per-lane top-2)", "bound": "mfma",
                            "achieved": round(pair_ops * knn_steps / (knn_ms * 1e-3) / 1e12, 1) if knn_ms else None,
                            "peak": 5000.0, "unit": "TOP/s",
                            "frac": round(pair_ops * knn_steps / (knn_ms * 1e-3) / 1e12 / 5000.0, 4) if knn_ms else None,
                            "measured_in": knn_src,
                            # the matcher's HBM share is tiny (B_match = 32 (Nq + Nt) + 16 Nq per pair) and its
                            # work in the units of the popcount formulation is 8 dword xor+popcounts per descriptor pair
                            "hbm_frac": round((32.0 * float(cnt_host[1:].sum() + cnt_host[:-1].sum()) + 16.0 * float(cnt_host[1:].sum()))
                                              * knn_steps / (knn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if knn_ms else None,
                            "popcount_dword_equivalents_per_s": round(pair_ops / 64.0 * knn_steps / (knn_ms * 1e-3), 1) if knn_ms else None}
        roofline = {
            "bound": "hbm", "kernel": "k_fast_blur_band (FAST-9 + NMS + 7x7 Gaussian of one level + bilinear resize to the next level, "
                                      "fused; 8 level launches per pass)" if fused_pyramid else
                                      "k_fast_blur_band (FAST-9 + NMS + 7x7 Gaussian of one level, fused; 8 level launches per pass)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": (int(traffic_pf * frames_per_launch) if traffic_pf and (W, H, NF) == (640, 480, 2000) else None),
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": int(fb_bytes_per_launch),
            "avg_launch_ms": round(fb_ms_per_launch, 4), "frames_per_launch": frames_per_launch,
            "dominant_stage": dom,
            "whole_extractor": {"algorithmic_bytes_per_frame": b_extract,
                                "achieved_GBs": round(b_extract * prof_frames / (ext_ms * 1e-3) / 1e9, 1) if ext_ms else None,
                                "frac": round(b_extract * prof_frames / (ext_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ext_ms else None},
            "stage_us_per_frame": {k: round(1e3 * all_ms[k] / max(frames_all, 1), 3) for k in all_ms},
            "stage_us_per_frame_note": "every stage bracketed, one extra untimed step after the timed region; the timed "
                                       "region brackets only fast_blur (achieved / avg_launch_ms), plus knn2 with --no-pipeline",
            # kNN-2 runs on the matrix cores (knn2_mfma.hip): exact int8 inner products of bit-widened descriptors.
            # ops = 2 * 256 * sum(nq * nt) per launch; peak = 2 x the dense bf16 MFMA peak (MI355X_MICROARCH.md, I8 row)
            "matcher": matcher_roofline,
        }
        out = {
            "metric": "frames/s ORB extract+BF-match, 640x480 @2000 kp" if (W, H, NF) == (640, 480, 2000)
                      else "frames/s ORB extract+BF-match, %dx%d @%d kp" % (W, H, NF),
            "value": round(value, 1), "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: sequence of %d synthetic %dx%d frame pairs per GPU "
                                   "(%d frames), %d kp/frame, 1 extract + 1 kNN-2/ratio match per frame, inputs "
                                   "resident in HBM" % (args.pairs, W, H, B, NF),
                       "frames_per_gpu_per_step": B, "chunk_frames": args.chunk, "ratio": args.ratio,
                       "parallelism": "frames sharded by contiguous range, %d rank(s), no collective" % n_gpus,
                       "streams": ("extractor and matcher on two streams, double-buffered outputs: the matcher of step s "
                                   "runs beside the extractor of step s+1") if args.pipeline else "one stream",
                       "mean_keypoints_per_frame": round(float(cnt_host.mean()), 2),
                       "mean_matches_per_frame": round(float(nm_host.mean()), 2),
                       "slow_path_blocks": int(slow_blocks)},
            "roofline": roofline,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H, NF, args.cpu_budget)
        print(json.dumps(out), flush=True)

    ext.close()
    mat.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
