"""GPU tests of the C-ABI surface beyond the basic parity cases: async form, error conventions, device-resident
batch entry points, ragged/empty inputs, loop-closure DB scan, and size-independent properties at the
BASELINE.json configuration sizes. All comparisons are exact (bytes / indices / float bits)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def _ext(aria, nf=2000, w=640, h=480, **kw):
    return aria.OrbHipExtractor(max_features=nf, max_width=w, max_height=h, **kw)


# ---- IFeatureExtractor conventions --------------------------------------------------------------------------
def test_async_equals_sync_and_pending_rules(aria):
    a, b = aria.synth_frame_pair(4)
    e = _ext(aria)
    try:
        fs = e.extract(a)
        assert e.sync() is None                                  # OrbCudaExtractor.cpp:177: nothing pending -> no-op
        frame = {"id": 42}
        e.extractAsync(a, frame=frame)
        with pytest.raises(aria.AriaError) as ex:                # one pending slot (OrbCudaExtractor.hpp:47-49)
            e.extractAsync(b)
        assert ex.value.status == -7
        with pytest.raises(aria.AriaError):
            e._L.aria_orb_extract.restype = C.c_int
            from aria_slam_amd._lib import check
            check(e._L.aria_orb_extract(e._h, a.ctypes.data, 640, 480, 640, None, None, 0, C.byref(C.c_int())), "extract while pending")
        fa = e.sync()
        assert fa is frame and frame["id"] == 42 and frame["width"] == 640 and frame["height"] == 480
        assert fa["keypoints"].tobytes() == fs["keypoints"].tobytes()
        assert np.array_equal(fa["descriptors"], fs["descriptors"])
        n = C.c_int()
        assert e._L.aria_orb_sync(e._h, None, None, 0, C.byref(n)) == -8   # ARIA_E_NOT_PENDING at the C level
    finally:
        e.close()


def test_extract_error_codes(aria):
    a, _ = aria.synth_frame_pair(4)
    e = _ext(aria)
    try:
        n = C.c_int()
        kp = np.empty(10, aria.KP_DTYPE)
        ds = np.empty((10, 32), np.uint8)
        rc = e._L.aria_orb_extract(e._h, a.ctypes.data, 640, 480, 640, kp.ctypes.data, ds.ctypes.data, 10, C.byref(n))
        assert rc == -5 and n.value == 2000                       # OUTPUT_TOO_SMALL reports the required count
        big = np.zeros((481, 640), np.uint8)
        rc = e._L.aria_orb_extract(e._h, big.ctypes.data, 640, 481, 640, kp.ctypes.data, ds.ctypes.data, 10, C.byref(n))
        assert rc == -4                                           # TOO_LARGE
        rc = e._L.aria_orb_extract(e._h, None, 640, 480, 640, kp.ctypes.data, ds.ctypes.data, 10, C.byref(n))
        assert rc == -1
        rc = e._L.aria_orb_extract(e._h, a.ctypes.data, 640, 480, 600, kp.ctypes.data, ds.ctypes.data, 10, C.byref(n))
        assert rc == -1                                           # stride < width
    finally:
        e.close()


def test_strided_input_and_smaller_image_on_big_handle(aria, oracle):
    a, _ = aria.synth_frame_pair(9, 600, 400)
    padded = np.zeros((400, 700), np.uint8)
    padded[:, :600] = a
    e = _ext(aria, nf=800, w=752, h=480)
    try:
        cap = e.kp_capacity()
        kp = np.empty(cap, aria.KP_DTYPE)
        ds = np.empty((cap, 32), np.uint8)
        n = C.c_int()
        rc = e._L.aria_orb_extract(e._h, padded.ctypes.data, 600, 400, 700, kp.ctypes.data, ds.ctypes.data, cap, C.byref(n))
        assert rc == 0
        ok, od = oracle.orb_extract(a, oracle.default_params(800))
        assert n.value == len(ok) and kp[:n.value].tobytes() == ok.tobytes() and np.array_equal(ds[:n.value], od)
    finally:
        e.close()


def test_set_max_features_recreates_like_reference(aria, oracle):
    a, _ = aria.synth_frame_pair(6)
    e = _ext(aria, nf=1000)
    try:
        assert e.getMaxFeatures() == 1000
        for nf in (300, 2000, 0, 1000):
            e.setMaxFeatures(nf)
            assert e.getMaxFeatures() == nf
            f = e.extract(a)
            ok, od = oracle.orb_extract(a, oracle.default_params(nf))
            assert f["keypoints"].tobytes() == ok.tobytes() and np.array_equal(f["descriptors"], od)
    finally:
        e.close()


# ---- degenerate inputs ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["flat", "tiny", "narrow", "checker"])
def test_degenerate_images(aria, oracle, kind):
    if kind == "flat":
        img = np.full((480, 640), 77, np.uint8)
    elif kind == "tiny":
        img = aria.synth_frame_pair(2, 100, 90)[0]            # most levels are <= 62 px -> no keypoints there
    elif kind == "narrow":
        img = aria.synth_frame_pair(3, 640, 70)[0]
    else:
        yy, xx = np.mgrid[0:240, 0:320]
        img = (((xx // 8 + yy // 8) & 1) * 200 + 20).astype(np.uint8)   # exact ties everywhere
    h, w = img.shape
    e = _ext(aria, nf=500, w=w, h=h)
    try:
        f = e.extract(img)          # the checkerboard is a tie storm: every tie is kept, as retainBest does
        ok, od = oracle.orb_extract(img, oracle.default_params(500), cap=20000)

        assert len(f["keypoints"]) == len(ok)
        assert f["keypoints"].tobytes() == ok.tobytes() and np.array_equal(f["descriptors"], od)
        if kind == "flat":
            assert len(ok) == 0
    finally:
        e.close()


def test_noise_image_many_candidates(aria, oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    e = _ext(aria, nf=2000)
    try:
        f = e.extract(img)
        ok, od = oracle.orb_extract(img, oracle.default_params(2000))
        assert f["keypoints"].tobytes() == ok.tobytes() and np.array_equal(f["descriptors"], od)
    finally:
        e.close()


# ---- device-resident batch forms --------------------------------------------------------------------------------
def test_batch_device_equals_per_frame_and_chunks(aria, torch_cuda):
    torch = torch_cuda
    n_pairs, W, H, NF = 5, 640, 480, 2000
    seq = aria.synth_sequence(20, n_pairs, W, H)
    B = 2 * n_pairs
    dev = torch.device("cuda", 0)
    images = torch.from_numpy(seq).to(dev)
    work = torch.cuda.Stream(device=dev)           # a real stream to borrow (torch's default stream has handle 0 = "none")
    stream = work.cuda_stream
    torch.cuda.synchronize()
    single = _ext(aria)
    per_frame = [single.extract(seq[i]) for i in range(B)]
    single.close()
    for chunk in (1, 3, 16):
        e = aria.OrbHipExtractor(max_features=NF, stream=stream, max_width=W, max_height=H, max_batch=chunk)
        cap = e.kp_capacity()
        kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        counts = torch.zeros((B,), dtype=torch.int32, device=dev)
        e.extract_batch_device(images, B, W, H, kps, desc, counts, cap)
        e.check()
        c = counts.cpu().numpy()
        k = kps.cpu().numpy()
        d = desc.cpu().numpy()
        for i in range(B):
            assert c[i] == len(per_frame[i]["keypoints"])
            assert k[i, :c[i]].tobytes() == per_frame[i]["keypoints"].tobytes()
            assert np.array_equal(d[i, :c[i]], per_frame[i]["descriptors"])
        # device matcher on resident descriptors: pair p = (frame p+1 as query, frame p as train)
        m = aria.HipMatcher(stream=stream)
        matches = torch.zeros((B, cap, 12), dtype=torch.uint8, device=dev)
        nm = torch.zeros((B,), dtype=torch.int32, device=dev)
        m.match_batch_device(desc.data_ptr() + cap * 32, counts.data_ptr() + 4, desc, counts, B - 1, cap * 32, 0.75,
                             matches, nm, cap)
        m.sync()
        nmh = nm.cpu().numpy()
        mh = matches.cpu().numpy()
        host = aria.HipMatcher()
        for p in range(B - 1):
            want = host.match(per_frame[p + 1], per_frame[p], None, 0.75)
            assert nmh[p] == len(want) and mh[p, :nmh[p]].tobytes() == want.tobytes()
        host.close()
        m.close()
        e.close()


@pytest.mark.parametrize("off,row_stride", [(1, 643), (2, 642), (4, 644), (3, 640)])
def test_batch_device_byte_aligned_images(aria, torch_cuda, off, row_stride):
    """Resident images that are only byte-aligned (odd base address and/or odd row stride) take the per-byte load
    paths of every kernel that reads level 0; results must equal the aligned host-buffer path."""
    torch = torch_cuda
    W, H, NF, B = 640, 480, 1000, 3
    seq = aria.synth_sequence(31, 2, W, H)[:B]
    dev = torch.device("cuda", 0)
    frame_stride = row_stride * H + 5
    buf = np.zeros(off + B * frame_stride + 64, np.uint8)
    for i in range(B):
        v = buf[off + i * frame_stride: off + i * frame_stride + row_stride * H].reshape(H, row_stride)
        v[:, :W] = seq[i]
    d_buf = torch.from_numpy(buf).to(dev)
    single = _ext(aria, nf=NF)
    per_frame = [single.extract(seq[i]) for i in range(B)]
    single.close()
    e = aria.OrbHipExtractor(max_features=NF, max_width=W, max_height=H, max_batch=2)
    try:
        cap = e.kp_capacity()
        kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        counts = torch.zeros((B,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.extract_batch_device(d_buf.data_ptr() + off, B, W, H, kps, desc, counts, cap, frame_stride=frame_stride,
                               row_stride=row_stride)
        e.check()
        c, k, d = counts.cpu().numpy(), kps.cpu().numpy(), desc.cpu().numpy()
        for i in range(B):
            assert c[i] == len(per_frame[i]["keypoints"])
            assert k[i, :c[i]].tobytes() == per_frame[i]["keypoints"].tobytes()
            assert np.array_equal(d[i, :c[i]], per_frame[i]["descriptors"])
    finally:
        e.close()


def test_batch_kp_cap_too_small_is_reported(aria, torch_cuda):
    torch = torch_cuda
    seq = aria.synth_sequence(1, 1, 640, 480)
    dev = torch.device("cuda", 0)
    images = torch.from_numpy(seq).to(dev)
    e = aria.OrbHipExtractor(max_features=2000, max_width=640, max_height=480, max_batch=2)
    try:
        cap = 100
        kps = torch.zeros((2, cap, 24), dtype=torch.uint8, device=dev)
        desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device=dev)
        counts = torch.zeros((2,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.extract_batch_device(images, 2, 640, 480, kps, desc, counts, cap)
        with pytest.raises(aria.AriaError) as ex:
            e.check()
        assert ex.value.status == -5
        assert counts.cpu().numpy().tolist() == [cap, cap]       # clamped, never written past the caller's rows
        e.check()                                                # flag is cleared after being reported
    finally:
        e.close()


def test_ragged_and_empty_match_batches(aria, oracle, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(8)
    rows = 300
    nq = [0, 1, 2, 257, 300, 64, 5]
    nt = [10, 0, 1, 300, 2, 256, 300]
    P = len(nq)
    q = rng.integers(0, 256, (P, rows, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (P, rows, 32), dtype=np.uint8)
    t[3, :100] = q[3, :100]
    t[5, :64] = q[5, :64]
    t[5, 64:128] = q[5, :64]                                   # exact duplicates: ties -> lower train index
    dev = torch.device("cuda", 0)
    dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    dnq, dnt = torch.tensor(nq, dtype=torch.int32, device=dev), torch.tensor(nt, dtype=torch.int32, device=dev)
    m = aria.HipMatcher()
    try:
        for ratio in (0.75, 0.0, 1.0):
            out = torch.zeros((P, rows, 12), dtype=torch.uint8, device=dev)
            nm = torch.full((P,), -1, dtype=torch.int32, device=dev)
            m.match_batch_device(dq, dnq, dt, dnt, P, rows * 32, ratio, out, nm, rows)
            m.sync()
            nmh, oh = nm.cpu().numpy(), out.cpu().numpy()
            for p in range(P):
                want = oracle.match_ratio(q[p, :nq[p]], t[p, :nt[p]], ratio)
                assert nmh[p] == len(want), (p, ratio)
                assert oh[p, :nmh[p]].tobytes() == want.tobytes()
        # host entry points on the same ragged cases
        for p in range(P):
            got = m.match(q[p, :nq[p]], t[p, :nt[p]], None, 0.75)
            assert got.tobytes() == oracle.match_ratio(q[p, :nq[p]], t[p, :nt[p]], 0.75).tobytes()
            if nq[p]:
                idx, dist = m.knn2(q[p, :nq[p]], t[p, :nt[p]])
                oi, od = oracle.knn2(q[p, :nq[p]], t[p, :nt[p]])
                assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    finally:
        m.close()


def test_match_appends_like_cudamatcher(aria):
    a, b = aria.synth_frame_pair(1)
    e = _ext(aria)
    m = aria.HipMatcher()
    try:
        fa, fb = e.extract(a), e.extract(b)
        acc = [("sentinel",)]
        r1 = m.match(fb, fa, acc)
        assert acc[0] == ("sentinel",) and len(acc) == 1 + len(r1)       # CudaMatcher.cpp:65 push_back, no clear
        m.match({"descriptors": np.zeros((0, 32), np.uint8)}, fa, acc)   # :35-37 untouched on empty input
        assert len(acc) == 1 + len(r1)
        allm = [["old"], [], [], []]
        res = m.matchMultiple(fb, [fa, fb], allm)
        assert len(allm) == 2 and allm[0][0] == "old" and len(allm[0]) == 1 + len(res[0])   # IMatcher.hpp:33-36
    finally:
        e.close()
        m.close()


# ---- loop-closure DB scan -----------------------------------------------------------------------------------------
def test_keyframe_db_scan_equals_oracle(aria, oracle, torch_cuda):
    torch = torch_cuda
    from aria_slam_amd import loopdb
    dev = torch.device("cuda", 0)
    e = _ext(aria, nf=1000)
    m = aria.HipMatcher()
    try:
        frames, ids = [], []
        for k in range(6):
            a, b = aria.synth_frame_pair(50 + k)
            frames += [e.extract(a), e.extract(b)]
            ids += [10 * (2 * k), 10 * (2 * k + 1)]
        query = frames[3]                                           # B of seed 51: its partner A (index 2) must score
        rows = 1100
        db = loopdb.KeyframeDB(k_cap=16, rows=rows, device=dev)
        for f, kid in zip(frames, ids):
            d = torch.from_numpy(f["descriptors"]).to(dev)
            db.add(kid, d, len(f["descriptors"]))
        dq = torch.from_numpy(query["descriptors"]).to(dev)
        nq = len(query["descriptors"])
        good = torch.zeros((16,), dtype=torch.int32, device=dev)
        m.match_db_device(dq, nq, db.desc, db.counts, 16, rows * 32, 0.7, good)
        m.sync()
        want_good = [oracle.count_good_matches_f64(query["descriptors"], f["descriptors"], 0.7) for f in frames]
        assert good.cpu().numpy()[:12].tolist() == want_good and good.cpu().numpy()[12:].max() == 0
        for qid, mfb in ((1000, 30), (45, 30), (1000, 0)):
            got = db.find_candidates(m, dq, nq, qid, mfb)
            ci, cs = oracle.loop_candidates(query["descriptors"], qid, [f["descriptors"] for f in frames], ids, mfb)
            assert [i for i, _ in got] == ci.tolist() and [s for _, s in got] == cs.tolist()
        got = db.find_candidates(m, dq, nq, 1000, 30)
        assert got[0][0] == 3 and got[0][1] == 1.0 and got[1][0] == 2   # itself, then its partner frame
    finally:
        e.close()
        m.close()


# ---- BASELINE.json sizes: size-independent properties ------------------------------------------------------------
def test_config4_1408_properties(aria, torch_cuda):
    """1408x1408, 4000 kp (configs[3]): canonical order, idempotence, descriptor/keypoint consistency."""
    torch = torch_cuda
    W = H = 1408
    seq = aria.synth_sequence(300, 2, W, H)
    dev = torch.device("cuda", 0)
    images = torch.from_numpy(seq).to(dev)
    e = aria.OrbHipExtractor(max_features=4000, max_width=W, max_height=H, max_batch=3)
    try:
        cap = e.kp_capacity()
        outs = []
        for _ in range(2):
            kps = torch.zeros((4, cap, 24), dtype=torch.uint8, device=dev)
            desc = torch.zeros((4, cap, 32), dtype=torch.uint8, device=dev)
            counts = torch.zeros((4,), dtype=torch.int32, device=dev)
            e.extract_batch_device(images, 4, W, H, kps, desc, counts, cap)
            e.check()
            outs.append((kps.cpu().numpy().copy(), desc.cpu().numpy().copy(), counts.cpu().numpy().copy()))
        assert all(np.array_equal(x, y) for x, y in zip(outs[0], outs[1]))           # idempotent
        k, d, c = outs[0]
        info = e.level_info(W, H)
        for i in range(4):
            assert c[i] == 4000
            kp = k[i, :c[i]].copy().view(aria.KP_DTYPE).reshape(-1)
            assert np.all(np.diff(kp["octave"]) >= 0)                                # level ascending
            assert np.bincount(kp["octave"], minlength=8).tolist() == [q for _, _, q, _ in info]
            for l in range(8):
                s = kp[kp["octave"] == l]
                assert np.all(np.diff(s["response"]) <= 0)                           # response descending
                lw, lh, _, sc = info[l]
                xl, yl = s["x"] / np.float32(sc), s["y"] / np.float32(sc)
                assert xl.min() >= 30.5 and xl.max() <= lw - 31.5 and yl.min() >= 30.5 and yl.max() <= lh - 31.5
                assert np.all(s["size"] == np.float32(31) * np.float32(sc))
            assert np.all((kp["angle"] >= 0) & (kp["angle"] <= 360))
            assert len(np.unique(np.stack([kp["octave"], kp["x"], kp["y"]], 1), axis=0)) == c[i]   # no duplicates
        single = e.extract(seq[1])
        assert single["keypoints"].tobytes() == k[1, :c[1]].tobytes() and np.array_equal(single["descriptors"], d[1, :c[1]])
    finally:
        e.close()


def test_config3_batch_checksum_of_checksums(aria, torch_cuda):
    """A 256-frame slice of configs[2]: the batched device path equals frame-by-frame extraction (hash of hashes),
    and every frame fills its 2000-keypoint budget (SURVEY.md 8d requirement on the generator)."""
    import hashlib
    torch = torch_cuda
    W, H, NF, NP = 640, 480, 2000, 128
    seq = aria.synth_sequence(1, NP, W, H)
    dev = torch.device("cuda", 0)
    images = torch.from_numpy(seq).to(dev)
    B = 2 * NP
    e = aria.OrbHipExtractor(max_features=NF, max_width=W, max_height=H, max_batch=96)
    try:
        cap = e.kp_capacity()
        kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        counts = torch.zeros((B,), dtype=torch.int32, device=dev)
        e.extract_batch_device(images, B, W, H, kps, desc, counts, cap)
        e.check()
        c, k, d = counts.cpu().numpy(), kps.cpu().numpy(), desc.cpu().numpy()
        assert c.min() == NF and c.max() == NF
        hb = hashlib.sha256()
        hs = hashlib.sha256()
        for i in range(B):
            hb.update(hashlib.sha256(k[i, :c[i]].tobytes() + d[i, :c[i]].tobytes()).digest())
            f = e.extract(seq[i])
            hs.update(hashlib.sha256(f["keypoints"].tobytes() + f["descriptors"].tobytes()).digest())
        assert hb.hexdigest() == hs.hexdigest()
    finally:
        e.close()


def test_survivor_queue_self_tunes_on_corner_dense_images(aria, oracle):
    """A noise image overflows the band kernel's survivor queue: those workgroups take the dense-rescoring path
    (same bits), are counted, and the handle enlarges the queue so later frames avoid it."""
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    ok, od = oracle.orb_extract(img, oracle.default_params(1000))
    e = _ext(aria, nf=1000)
    try:
        slow = []
        for _ in range(5):
            f = e.extract(img)
            assert f["keypoints"].tobytes() == ok.tobytes() and np.array_equal(f["descriptors"], od)
            slow.append(e.slow_path_blocks(reset=True))
        assert slow[0] > 0, slow
        assert slow[-1] < slow[0], slow
        a, _ = aria.synth_frame_pair(1)
        e.extract(a)
        assert e.slow_path_blocks() == 0
    finally:
        e.close()


def test_profiling_stage_masks(aria, torch_cuda):
    """aria_orb_set_profiling / aria_matcher_set_profiling: 1 brackets every stage, an even value only the stages
    whose bit (s + 1) is set; unbracketed stages report 0 ms and 0 launches; results do not depend on profiling."""
    torch = torch_cuda
    W, H, NF, B = 640, 480, 1000, 4
    seq = aria.synth_sequence(77, 2, W, H)[:B]
    dev = torch.device("cuda", 0)
    images = torch.from_numpy(seq).to(dev)
    e = aria.OrbHipExtractor(max_features=NF, max_width=W, max_height=H, max_batch=B)
    m = aria.HipMatcher()
    try:
        cap = e.kp_capacity()
        kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        counts = torch.zeros((B,), dtype=torch.int32, device=dev)
        matches = torch.zeros((B, cap, 12), dtype=torch.uint8, device=dev)
        nm = torch.zeros((B,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()

        def run():
            e.extract_batch_device(images, B, W, H, kps, desc, counts, cap)
            e.check()
            m.match_batch_device(desc.data_ptr() + cap * 32, counts.data_ptr() + 4, desc, counts, B - 1, cap * 32, 0.75,
                                 matches, nm, cap)
            m.sync()
            return desc.cpu().numpy().copy(), nm.cpu().numpy().copy()

        ref = run()
        e.set_profiling(True, stages=["fast_blur"])
        m.set_profiling(True, stages=["knn2"])
        got = run()
        pe, frames = e.get_profile()
        pm, pairs = m.get_profile()
        assert frames == B and pairs == B - 1
        assert pe["fast_blur"][0] > 0 and pe["fast_blur"][1] == 8
        assert all(pe[s] == (0.0, 0) for s in ("resize", "select", "describe"))
        assert pm["knn2"][0] > 0 and pm["knn2"][1] == 1 and pm["ratio_compact"] == (0.0, 0)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        e.set_profiling(True)
        m.set_profiling(True)
        run()
        pe, _ = e.get_profile()
        pm, _ = m.get_profile()
        assert pe["select"][1] == 2 and pe["describe"][1] == 2 and pe["fast_blur"][1] == 8   # + the tie-storm fallback launches
        assert pe["resize"][1] in (0, 7)          # 0: pyramid fused into the FAST/blur launches (default)
        assert pm["ratio_compact"][1] == 1
        e.set_profiling(False)
        m.set_profiling(False)
        run()
        assert e.get_profile()[1] == 0
    finally:
        e.close()
        m.close()


def test_borrowed_stream_orders_extractor_and_matcher(aria, torch_cuda):
    """Two handles that borrow ONE real stream run in order without any host sync in between (bench.py's serial mode);
    with two streams the same holds once the matcher's stream waits for an event recorded after the extraction
    (bench.py's pipelined mode). Both must equal the host-synchronised result. (torch's default stream has handle 0,
    which means "no stream": handles given 0 create their own, unordered streams.)"""
    torch = torch_cuda
    W, H, NF, B = 640, 480, 1000, 6
    seq = aria.synth_sequence(91, 3, W, H)[:B]
    dev = torch.device("cuda", 0)
    images = torch.from_numpy(seq).to(dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    assert s1.cuda_stream != 0 and s2.cuda_stream != 0
    torch.cuda.synchronize()

    def run(mode):
        e = aria.OrbHipExtractor(max_features=NF, stream=s1.cuda_stream, max_width=W, max_height=H, max_batch=B)
        m = aria.HipMatcher(stream=(s1 if mode != "two_streams" else s2).cuda_stream)
        try:
            assert e.stream == s1.cuda_stream
            cap = e.kp_capacity()
            kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
            desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
            counts = torch.zeros((B,), dtype=torch.int32, device=dev)
            matches = torch.zeros((B, cap, 12), dtype=torch.uint8, device=dev)
            nm = torch.zeros((B,), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            for _ in range(3):                      # repeated, so a too-early matcher would see the zeroed buffers below
                with torch.cuda.stream(s1):
                    desc.zero_()
                    counts.zero_()
                    e.extract_batch_device(images, B, W, H, kps, desc, counts, cap)
                    ev = torch.cuda.Event()
                    ev.record(s1)
                if mode == "host_sync":
                    e.check()
                ms = s1 if mode != "two_streams" else s2
                with torch.cuda.stream(ms):
                    if mode == "two_streams":
                        s2.wait_event(ev)
                    m.match_batch_device(desc.data_ptr() + cap * 32, counts.data_ptr() + 4, desc, counts, B - 1, cap * 32,
                                         0.75, matches, nm, cap)
                    done = torch.cuda.Event()
                    done.record(ms)
                if mode == "two_streams":
                    s1.wait_event(done)             # the next round zeroes the buffers the matcher is reading
            torch.cuda.synchronize()
            e.check()
            m.sync()
            n = nm.cpu().numpy().copy()
            mm = matches.cpu().numpy().copy()
            return n, [mm[p, :n[p]].tobytes() for p in range(B - 1)]
        finally:
            e.close()
            m.close()

    ref = run("host_sync")
    assert ref[0][:B - 1].min() > 0
    for mode in ("one_stream", "two_streams"):
        got = run(mode)
        assert np.array_equal(got[0], ref[0]) and got[1] == ref[1], mode


# ---- handles are independent: no process-wide launch state ---------------------------------------------------
def test_two_handles_from_two_host_threads(aria, oracle):
    """include/aria_orb_hip.h: independent handles on different streams may run concurrently (the reference's adapters are
    one-instance-per-thread objects, include/adapters/gpu/OrbCudaExtractor.hpp:38-50). Two host threads each create
    their own extractor + matcher, work at the same time on different sizes, and both equal the oracle."""
    import threading
    jobs = [(11, 640, 480, 2000), (12, 752, 480, 1000)]
    out, errs = {}, []

    def work(i):
        seed, w, h, nf = jobs[i]
        try:
            a, b = aria.synth_frame_pair(seed, w, h)
            e = aria.OrbHipExtractor(max_features=nf, max_width=w, max_height=h)
            m = aria.HipMatcher(max_query=nf + 1024, max_train=nf + 1024)
            try:
                res = []
                for _ in range(6):                       # several rounds so the two threads really overlap
                    fa, fb = e.extract(a), e.extract(b)
                    res.append((fa, fb, m.match(fb, fa, None, 0.75)))
                out[i] = res
            finally:
                e.close()
                m.close()
        except Exception as ex:                          # surfaced in the main thread
            errs.append((i, ex))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for i, (seed, w, h, nf) in enumerate(jobs):
        a, b = aria.synth_frame_pair(seed, w, h)
        p = oracle.default_params(nf)
        (ka, da), (kb, db) = oracle.orb_extract(a, p), oracle.orb_extract(b, p)
        want = oracle.match_ratio(db, da, 0.75)
        for fa, fb, got in out[i]:
            assert fa["keypoints"].tobytes() == ka.tobytes() and np.array_equal(fa["descriptors"], da)
            assert fb["keypoints"].tobytes() == kb.tobytes() and np.array_equal(fb["descriptors"], db)
            assert got.tobytes() == want.tobytes()


# ---- tie storms: OpenCV's retainBest keeps every tie (reference call site src/legacy/Frame.cpp:45-49) ---------
def _checker(w, h, cell):
    yy, xx = np.mgrid[0:h, 0:w]
    return (((xx // cell + yy // cell) & 1) * 200 + 20).astype(np.uint8)


def _dots(w, h, pitch):
    """Single bright pixels on a flat background, on a regular grid: every dot is a FAST corner with the same score and
    the same Harris response at level 0 -- exact ties at both of retainBest's cuts."""
    img = np.full((h, w), 50, np.uint8)
    img[pitch // 2::pitch, pitch // 2::pitch] = 255
    return img


@pytest.mark.parametrize("kind,w,h,par,nf", [("checker", 320, 240, 8, 500), ("checker", 640, 480, 8, 2000),
                                             ("dots", 640, 480, 16, 2000), ("dots", 640, 480, 7, 2000),
                                             ("dots", 752, 480, 11, 1000)])
def test_tie_storm_single_frame_equals_oracle(aria, oracle, kind, w, h, par, nf):
    """Tie storms: checkerboards (thousands of candidates with one FAST score: more than the on-chip sort holds) and dot
    grids (every level-0 keypoint ties in FAST score AND Harris response, so retainBest keeps them all and the frame
    returns more keypoints than the plan has rows). The global-memory fallback (k_select_ovf + the arena pass of
    k_describe) must return exactly the oracle's keypoints, in canonical order, through the host entry points."""
    img = _checker(w, h, par) if kind == "checker" else _dots(w, h, par)
    e = _ext(aria, nf=nf, w=w, h=h)
    try:
        ok, od = oracle.orb_extract(img, oracle.default_params(nf), cap=400000)
        if kind == "dots" and par < 16:
            assert len(ok) > e.kp_capacity()                    # more keypoints than the plan's rows
        f = e.extract(img)
        assert len(f["keypoints"]) == len(ok)
        assert f["keypoints"].tobytes() == ok.tobytes() and np.array_equal(f["descriptors"], od)
        e.extractAsync(img)                                     # the async form takes the same route
        g = e.sync()
        assert g["keypoints"].tobytes() == ok.tobytes() and np.array_equal(g["descriptors"], od)
        a, _ = aria.synth_frame_pair(5, w, h)                   # and the handle is fine afterwards
        fa = e.extract(a)
        ka, da = oracle.orb_extract(a, oracle.default_params(nf))
        assert fa["keypoints"].tobytes() == ka.tobytes() and np.array_equal(fa["descriptors"], da)
    finally:
        e.close()


def test_tie_storm_inside_a_device_batch(aria, oracle, torch_cuda):
    """Batch entry point: a dot grid (tie storm) between two ordinary frames. With the plan's kp_cap the call reports
    ARIA_E_OUTPUT_TOO_SMALL and the rows the frame needs; with that capacity all three frames equal the oracle."""
    torch = torch_cuda
    w, h, nf = 320, 240, 300
    dev = torch.device("cuda", 0)
    imgs = np.stack([aria.synth_frame_pair(8, w, h)[0], _dots(w, h, 7), aria.synth_frame_pair(9, w, h)[1]])
    want = [oracle.orb_extract(im, oracle.default_params(nf), cap=200000) for im in imgs]
    s = torch.cuda.Stream(device=dev)
    e = _ext(aria, nf=nf, w=w, h=h, max_batch=3, stream=s.cuda_stream)
    try:
        d_img = torch.from_numpy(imgs).to(dev)
        for attempt in range(2):
            cap = e.kp_capacity() if attempt == 0 else need
            kps = torch.zeros((3, cap, 24), dtype=torch.uint8, device=dev)
            desc = torch.zeros((3, cap, 32), dtype=torch.uint8, device=dev)
            cnt = torch.zeros((3,), dtype=torch.int32, device=dev)
            with torch.cuda.stream(s):
                e.extract_batch_device(d_img, 3, w, h, kps, desc, cnt, cap)
            if attempt == 0:
                with pytest.raises(aria.AriaError) as ex:
                    e.check()
                assert ex.value.status == -5
                need = e.rows_needed()
                assert need == len(want[1][0])
            else:
                e.check()
        cnt = cnt.cpu().numpy()
        for f in range(3):
            ok, od = want[f]
            assert cnt[f] == len(ok)
            assert kps[f, :cnt[f]].cpu().numpy().tobytes() == ok.tobytes()
            assert np.array_equal(desc[f, :cnt[f]].cpu().numpy(), od)
    finally:
        e.close()


# ---- dynamic-object filter hook (src/main.cpp:42-50, 164-175) -------------------------------
@pytest.mark.parametrize("mode", [0, 1])
def test_dynamic_object_filter_on_device_equals_oracle(aria, oracle, torch_cuda, mode):
    """Boxes are an input (the detector is out of scope). Keypoints are flagged on the device between describe and match;
    the batched matcher drops ratio-test survivors with a flagged endpoint and counts them. Both box tests (the legacy
    executable's integer cv::Rect test and core::Detection::contains) against the oracle's 10-line restatement."""
    torch = torch_cuda
    dev = torch.device("cuda", 0)
    w, h, nf, B = 640, 480, 1000, 6
    imgs = aria.synth_sequence(21, B // 2, w, h)
    rng = np.random.default_rng(5 + mode)
    box_cap = 80
    boxes = np.zeros((B, box_cap, 4), np.float32)
    nbox = np.array([0, 3, 80, 1, 7, 70], np.int32)                   # none, a few, more than one LDS batch
    for f in range(B):
        for b in range(nbox[f]):
            x1, y1 = rng.uniform(0, w - 40), rng.uniform(0, h - 40)
            bw, bh = rng.uniform(4, 120), rng.uniform(4, 120)
            r = np.array([x1, y1, x1 + bw, y1 + bh], np.float32)
            boxes[f, b] = np.round(r) if (mode == 0 or b % 2 == 0) else r    # legacy boxes are integer rectangles
    s = torch.cuda.Stream(device=dev)
    e = _ext(aria, nf=nf, w=w, h=h, max_batch=B, stream=s.cuda_stream)
    m = aria.HipMatcher(stream=s.cuda_stream, max_query=4096, max_train=4096)
    try:
        cap = e.kp_capacity()
        with torch.cuda.stream(s):
            d_img = torch.from_numpy(imgs).to(dev)
            kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
            desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
            cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
            flags = torch.full((B, cap), 7, dtype=torch.uint8, device=dev)
            d_boxes, d_nbox = torch.from_numpy(boxes).to(dev), torch.from_numpy(nbox).to(dev)
            matches = torch.zeros((B, cap, 12), dtype=torch.uint8, device=dev)
            nm = torch.zeros((B,), dtype=torch.int32, device=dev)
            nfl = torch.full((B,), -1, dtype=torch.int32, device=dev)
            e.extract_batch_device(d_img, B, w, h, kps, desc, cnt, cap)
            aria.flag_keypoints_device(s.cuda_stream, kps, cnt, B, cap, d_boxes, d_nbox, box_cap, mode, flags)
            # pairs (f, f-1) for f = 1..B-1: query block f, train block f-1; flags likewise
            m.match_batch_filtered_device(desc.data_ptr() + cap * 32, cnt.data_ptr() + 4, desc, cnt, B - 1, cap * 32, 0.75,
                                          flags.data_ptr() + cap, flags, cap, matches.data_ptr() + cap * 12, nm.data_ptr() + 4,
                                          cap, nfl.data_ptr() + 4)
        e.check()
        m.sync()
        kps_h, desc_h, cnt_h = kps.cpu().numpy(), desc.cpu().numpy(), cnt.cpu().numpy()
        fl_h, mt_h, nm_h, nfl_h = flags.cpu().numpy(), matches.cpu().numpy(), nm.cpu().numpy(), nfl.cpu().numpy()
        total_filtered = 0
        for f in range(1, B):
            kq = kps_h[f, :cnt_h[f]].copy().view(oracle.KP_DTYPE).reshape(-1)
            kt = kps_h[f - 1, :cnt_h[f - 1]].copy().view(oracle.KP_DTYPE).reshape(-1)
            all_m = oracle.match_ratio(desc_h[f, :cnt_h[f]], desc_h[f - 1, :cnt_h[f - 1]], 0.75)
            # main.cpp uses ONE detection list (the current frame's) for both endpoints; the device hook takes per-frame flags,
            # so the oracle is asked with the union semantics it implies: query flags from frame f's boxes, train flags from f-1's
            want_q = np.array([oracle.filter_dynamic_matches(kq, kq, np.array([(i, i, 0.0)], oracle.MATCH_DTYPE), boxes[f, :nbox[f]], mode)[1]
                               for i in range(0, len(kq), 97)])
            assert np.array_equal(fl_h[f, :len(kq)][::97], want_q)                    # the flags themselves
            keep = np.array([not (fl_h[f, q] or fl_h[f - 1, t]) for q, t in zip(all_m["query_idx"], all_m["train_idx"])], bool)
            assert nm_h[f] == keep.sum() and nfl_h[f] == (~keep).sum()
            assert mt_h[f, :nm_h[f]].tobytes() == all_m[keep].tobytes()
            total_filtered += int(nfl_h[f])
            assert fl_h[f, cnt_h[f]:].max(initial=0) == 0                              # rows past the count are cleared
        assert total_filtered > 0
        # same-boxes case against the oracle's whole-function restatement (one detection list for both endpoints)
        f = 2
        kq = kps_h[f, :cnt_h[f]].copy().view(oracle.KP_DTYPE).reshape(-1)
        kt = kps_h[f - 1, :cnt_h[f - 1]].copy().view(oracle.KP_DTYPE).reshape(-1)
        all_m = oracle.match_ratio(desc_h[f, :cnt_h[f]], desc_h[f - 1, :cnt_h[f - 1]], 0.75)
        with torch.cuda.stream(s):
            d_boxes2 = d_boxes.clone()
            d_boxes2[f - 1] = d_boxes[f]
            d_nbox2 = d_nbox.clone()
            d_nbox2[f - 1] = d_nbox[f]
            aria.flag_keypoints_device(s.cuda_stream, kps, cnt, B, cap, d_boxes2, d_nbox2, box_cap, mode, flags)
            m.match_batch_filtered_device(desc.data_ptr() + f * cap * 32, cnt.data_ptr() + 4 * f, desc.data_ptr() + (f - 1) * cap * 32,
                                          cnt.data_ptr() + 4 * (f - 1), 1, cap * 32, 0.75, flags.data_ptr() + f * cap,
                                          flags.data_ptr() + (f - 1) * cap, cap, matches, nm, cap, nfl)
        m.sync()
        want, wf = oracle.filter_dynamic_matches(kq, kt, all_m, boxes[f, :nbox[f]], mode)
        got_n, got_f = int(nm.cpu()[0]), int(nfl.cpu()[0])
        assert got_n == len(want) and got_f == wf and matches.cpu().numpy()[0, :got_n].tobytes() == want.tobytes()
        # the reference's semantics for the whole batch in ONE call: train flags = frame f's keypoints against frame
        # f + 1's boxes (box_frame_offset +1), query flags = frame f against its own; every pair equals the oracle's
        # restatement of main.cpp:164-175 with the current frame's detection list for both endpoints
        with torch.cuda.stream(s):
            tflags = torch.full((B, cap), 9, dtype=torch.uint8, device=dev)
            aria.flag_keypoints_device(s.cuda_stream, kps, cnt, B, cap, d_boxes, d_nbox, box_cap, mode, flags)
            aria.flag_keypoints_device(s.cuda_stream, kps, cnt, B, cap, d_boxes, d_nbox, box_cap, mode, tflags, box_frame_offset=1)
            m.match_batch_filtered_device(desc.data_ptr() + cap * 32, cnt.data_ptr() + 4, desc, cnt, B - 1, cap * 32, 0.75,
                                          flags.data_ptr() + cap, tflags, cap, matches.data_ptr() + cap * 12, nm.data_ptr() + 4,
                                          cap, nfl.data_ptr() + 4)
        m.sync()
        mt_h, nm_h, nfl_h = matches.cpu().numpy(), nm.cpu().numpy(), nfl.cpu().numpy()
        assert tflags.cpu().numpy()[B - 1].max() == 0                               # no frame B: no boxes, no flags
        for f in range(1, B):
            kq = kps_h[f, :cnt_h[f]].copy().view(oracle.KP_DTYPE).reshape(-1)
            kt = kps_h[f - 1, :cnt_h[f - 1]].copy().view(oracle.KP_DTYPE).reshape(-1)
            all_m = oracle.match_ratio(desc_h[f, :cnt_h[f]], desc_h[f - 1, :cnt_h[f - 1]], 0.75)
            want, wf = oracle.filter_dynamic_matches(kq, kt, all_m, boxes[f, :nbox[f]], mode)
            assert nm_h[f] == len(want) and nfl_h[f] == wf and mt_h[f, :nm_h[f]].tobytes() == want.tobytes(), f
    finally:
        e.close()
        m.close()


def test_stage_event_is_recorded_before_select(aria, torch_cuda):
    """aria_orb_set_stage_event: the caller's event is recorded on the handle's stream inside every batch call (before the
    select stage of the last pass), a second stream waiting on it proceeds, results are unchanged; only stage 2 is accepted."""
    torch = torch_cuda
    dev = torch.device("cuda", 0)
    W, H, NF, B = 640, 480, 1000, 5
    seq = aria.synth_sequence(41, 3, W, H)[:B]
    images = torch.from_numpy(seq).to(dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    e = aria.OrbHipExtractor(max_features=NF, stream=s1.cuda_stream, max_width=W, max_height=H, max_batch=2)   # three passes
    try:
        cap = e.kp_capacity()
        out = [(torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
                torch.zeros((B,), dtype=torch.int32, device=dev)) for _ in range(2)]
        e.extract_batch_device(images, B, W, H, *out[0], cap)
        ev = torch.cuda.Event()
        ev.record(s1)
        assert e._L.aria_orb_set_stage_event(e._h, 1, ev.cuda_event) == -1       # ARIA_E_INVALID: only the select stage
        e.set_stage_event("select", ev.cuda_event)
        flag = torch.zeros((1,), dtype=torch.int32, device=dev)
        e.extract_batch_device(images, B, W, H, *out[1], cap)
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            flag += 1
        torch.cuda.synchronize(dev)
        e.check()
        assert int(flag.item()) == 1
        for a, b in zip(out[0], out[1]):
            assert torch.equal(a, b)
        e.set_stage_event("select", None)
        e.extract_batch_device(images, B, W, H, *out[1], cap)
        torch.cuda.synchronize(dev)
    finally:
        e.close()


# ---- round 3: advisor findings and the device-resident single-frame hand-off ---------------------------------
def test_match_graph_survives_key_buffer_growth(aria, oracle):
    """the captured single-pair graphs have the key buffer's address baked in. A call that makes the
    handle grow that buffer (matchMultiple with more candidates than the initial 16 slices' worth) must drop them: the
    same pair matched before and after has to give the oracle's matches, not a replay against freed memory."""
    a, b = aria.synth_frame_pair(21)
    e = _ext(aria, nf=1500)
    m = aria.HipMatcher()
    try:
        fa, fb = e.extract(a), e.extract(b)
        want = oracle.match_ratio(fb["descriptors"], fa["descriptors"], 0.75)
        for _ in range(3):                                    # third call replays the captured graph
            assert m.match(fb, fa).tobytes() == want.tobytes()
        L = m._L
        L.aria_matcher_match_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                               C.c_void_p, C.c_int, C.c_void_p]
        ncand = 20                                            # > kKnnSplitMax = 16 -> the key buffer is reallocated
        q = np.ascontiguousarray(fb["descriptors"])
        cands = [np.ascontiguousarray(fa["descriptors"] if i % 2 == 0 else fb["descriptors"]) for i in range(ncand)]
        ptrs = (C.c_void_p * ncand)(*[c.ctypes.data for c in cands])
        nts = (C.c_int * ncand)(*[len(c) for c in cands])
        out = np.empty((ncand, len(q)), aria.MATCH_DTYPE)
        n_out = (C.c_int * ncand)()
        rc = L.aria_matcher_match_multi(m._h, q.ctypes.data, len(q), ptrs, nts, ncand, C.c_float(0.75), out.ctypes.data, len(q), n_out)
        assert rc == 0
        for i in range(ncand):
            w = want if i % 2 == 0 else oracle.match_ratio(q, q, 0.75)
            assert out[i, :n_out[i]].tobytes() == w.tobytes()
        for _ in range(4):                                    # eager, eager, capture, replay -- all against the new buffer
            assert m.match(fb, fa).tobytes() == want.tobytes()
            assert m.match(fa, fb).tobytes() == oracle.match_ratio(fa["descriptors"], fb["descriptors"], 0.75).tobytes()
    finally:
        e.close()
        m.close()


def test_batch_error_survives_a_single_frame_call(aria, torch_cuda):
    """a single-frame extraction between a batch call and aria_orb_check must not wipe the batch's deferred
    error words (they used to share the single-frame result header)."""
    torch = torch_cuda
    seq = aria.synth_sequence(1, 1, 640, 480)
    dev = torch.device("cuda", 0)
    images = torch.from_numpy(seq).to(dev)
    e = aria.OrbHipExtractor(max_features=2000, max_width=640, max_height=480, max_batch=2)
    try:
        cap = 100
        kps = torch.zeros((2, cap, 24), dtype=torch.uint8, device=dev)
        desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device=dev)
        counts = torch.zeros((2,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.extract_batch_device(images, 2, 640, 480, kps, desc, counts, cap)
        f = e.extract(seq[0])                                 # zeroes ITS header, not the batch's words
        assert len(f["keypoints"]) >= 2000
        with pytest.raises(aria.AriaError) as ex:
            e.check()
        assert ex.value.status == -5 and e.rows_needed() >= 2000
        e.check()
    finally:
        e.close()


def test_device_handoff_equals_oracle(aria, oracle, torch_cuda):
    """getGpuDescriptors() / matchGpu (OrbCudaExtractor.hpp:34-35, CudaMatcher.hpp:22-28): the single-frame result is
    matched where it lies on the device; both orders, the resident-set rule, the explicit form, and the form queued behind
    extractAsync on a shared stream all give the oracle's matches for the frames' descriptors."""
    torch = torch_cuda
    W, H, NF = 640, 480, 2000
    seq = aria.synth_sequence(31, 3, W, H)                    # 6 frames
    want_f = [oracle.orb_extract(im, oracle.default_params(NF)) for im in seq]
    s = torch.cuda.Stream(device=torch.device("cuda", 0))
    for shared in (False, True):
        e = aria.OrbHipExtractor(max_features=NF, max_width=W, max_height=H, stream=s.cuda_stream if shared else None)
        m = aria.HipMatcher(stream=s.cuda_stream if shared else None)
        try:
            assert m.resident_rows() == -1
            prev = None
            for i, im in enumerate(seq):
                if shared and prev is not None:
                    # queued behind the extraction: the matcher reads the keypoint count on the device
                    fr = e.extractAsync(im)
                    kp, ds, cnt, n_host, rows = e.device_result()
                    assert n_host == -1 and rows >= e.kp_capacity()
                    m.match_device_async(ds, cnt, rows, new_is_query=(i % 2 == 0))
                    with pytest.raises(aria.AriaError) as ex:                 # one pending operation
                        m.match_device_async(ds, cnt, rows)
                    assert ex.value.status == -7
                    fr = e.sync()
                    n = len(fr["keypoints"])
                    got = m.finish(n, max(n, len(prev["keypoints"])))
                else:
                    fr = e.extract(im)
                    kp, ds, cnt, n_host, rows = e.device_result()
                    n = len(fr["keypoints"])
                    assert n_host == n
                    if prev is None:
                        m.retain_device(ds, n)
                        got = None
                    elif i % 2 == 0:
                        got = m.match_device(ds, n, None, len(prev["keypoints"]))          # query = current
                    else:
                        got = m.match_device(None, len(prev["keypoints"]), ds, n)          # query = previous (legacy order)
                assert fr["keypoints"].tobytes() == want_f[i][0].tobytes()
                assert np.array_equal(fr["descriptors"], want_f[i][1])
                if prev is not None:
                    q, t = (fr, prev) if i % 2 == 0 else (prev, fr)
                    want = oracle.match_ratio(q["descriptors"], t["descriptors"], 0.75)
                    assert got.tobytes() == want.tobytes(), (shared, i)
                    assert len(want) > 50 or i % 2 == 0         # frames 2k, 2k+1 show the same scene; 2k+1, 2k+2 do not
                assert m.resident_rows() == n
                prev = fr
            # explicit form: both sets given; the query becomes the resident one
            kp, ds, cnt, n_host, rows = e.device_result()
            d_other = torch.from_numpy(np.ascontiguousarray(want_f[0][1])).cuda()
            got = m.match_device(ds, n_host, d_other.data_ptr(), len(want_f[0][1]), 0.7)
            assert got.tobytes() == oracle.match_ratio(want_f[-1][1], want_f[0][1], 0.7).tobytes()
            # a resident-set size that does not match is refused, and the host entry point still works afterwards
            with pytest.raises(aria.AriaError) as ex:
                m.match_device(ds, n_host, None, n_host + 1)
            assert ex.value.status == -1
            assert m.match(want_f[1][1], want_f[0][1]).tobytes() == oracle.match_ratio(want_f[1][1], want_f[0][1], 0.75).tobytes()
            # ... and the host entry point's query is the resident set for a following device call
            got = m.match_device(None, len(want_f[1][1]), d_other.data_ptr(), len(want_f[0][1]))
            assert got.tobytes() == oracle.match_ratio(want_f[1][1], want_f[0][1], 0.75).tobytes()
            n_out = C.c_int()
            assert m._L.aria_matcher_finish(m._h, 0, None, 0, C.byref(n_out)) == -8      # nothing pending
        finally:
            e.close()
            m.close()


def test_kfdb_match_in_place_equals_fetch_then_match(aria, oracle):
    a, b = aria.synth_frame_pair(5)
    e = _ext(aria, nf=1000)
    m = aria.HipMatcher()
    L = m._L
    try:
        fa, fb = e.extract(a), e.extract(b)
        db = C.c_void_p()
        L.aria_kfdb_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.aria_kfdb_add.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_int]
        L.aria_kfdb_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                      C.POINTER(C.c_int)]
        L.aria_kfdb_destroy.argtypes = [C.c_void_p]
        assert L.aria_kfdb_create(0, None, 2, 2048, C.byref(db)) == 0
        for i, f in enumerate((fa, fb, fa)):                  # the third insertion drops the oldest: [fb, fa]
            d = np.ascontiguousarray(f["descriptors"])
            assert L.aria_kfdb_add(db, i, d.ctypes.data, len(d)) == 0
        q = np.ascontiguousarray(fb["descriptors"])
        out = np.empty(len(q), aria.MATCH_DTYPE)
        n = C.c_int()
        for index, kf in ((0, fb), (1, fa)):
            assert L.aria_kfdb_match(db, m._h, index, q.ctypes.data, len(q), C.c_float(0.7), out.ctypes.data, len(out), C.byref(n)) == 0
            assert out[:n.value].tobytes() == oracle.match_ratio(q, kf["descriptors"], 0.7).tobytes()
        assert L.aria_kfdb_match(db, m._h, 2, q.ctypes.data, len(q), C.c_float(0.7), out.ctypes.data, len(out), C.byref(n)) == -1
        L.aria_kfdb_destroy(db)
    finally:
        e.close()
        m.close()
