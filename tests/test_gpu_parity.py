"""GPU parity tests: HIP path (through the C-ABI) vs the CPU oracle, bit for bit.

The oracle is a restatement of OpenCV 4.9.0's CPU ORB / BFMatcher (parity unpinned: the reference holds no
golden vectors and OpenCV is not installed here), so these tests prove "HIP == restatement".
Tolerances: none. Every comparison below is exact (integer, byte, index and float-bit equality).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _kp_bits(k):
    return k.view(np.uint8).reshape(len(k), -1)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def ext2000(aria):
    e = aria.OrbHipExtractor(max_features=2000, max_width=752, max_height=480)
    yield e
    e.close()


@pytest.mark.parametrize("seed,w,h", [(1, 640, 480), (2, 640, 480), (7, 752, 480), (3, 333, 251)])
def test_pyramid_and_blur_levels_bit_exact(aria, oracle, ext2000, seed, w, h):
    a, _ = aria.synth_frame_pair(seed, w, h)
    ext2000.extract(a)
    p = oracle.default_params(2000)
    raw = oracle.build_pyramid(a, p)
    blur = oracle.blur_pyramid(raw, p, w, h)
    info = ext2000.level_info(w, h)
    for l in range(8):
        lw, lh = info[l][0], info[l][1]
        assert (lh, lw) == raw[l].shape
        g_raw = ext2000.debug_read_level(l, False, lw, lh)
        g_blur = ext2000.debug_read_level(l, True, lw, lh)
        assert np.array_equal(g_raw, raw[l]), "raw level %d differs at %d px" % (l, np.count_nonzero(g_raw != raw[l]))
        assert np.array_equal(g_blur, blur[l]), "blurred level %d differs at %d px" % (l, np.count_nonzero(g_blur != blur[l]))


@pytest.mark.parametrize("seed,w,h,nf", [(1, 640, 480, 2000), (2, 640, 480, 2000), (5, 640, 480, 500),
                                          (7, 752, 480, 1000), (3, 333, 251, 300)])
def test_extract_bit_exact(aria, oracle, seed, w, h, nf):
    a, b = aria.synth_frame_pair(seed, w, h)
    e = aria.OrbHipExtractor(max_features=nf, max_width=w, max_height=h)
    try:
        for img in (a, b):
            f = e.extract(img)
            p = oracle.default_params(nf)
            kps, desc = oracle.orb_extract(img, p)
            assert len(f["keypoints"]) == len(kps)
            assert np.array_equal(f["keypoints"]["octave"], kps["octave"])
            assert np.array_equal(_kp_bits(f["keypoints"]), _kp_bits(kps)), "keypoint records differ"
            assert np.array_equal(f["descriptors"], desc), "descriptors differ"
    finally:
        e.close()


def test_match_indices_identical(aria, oracle):
    a, b = aria.synth_frame_pair(1, 640, 480)
    e = aria.OrbHipExtractor(max_features=2000, max_width=640, max_height=480)
    m = aria.HipMatcher()
    try:
        fa, fb = e.extract(a), e.extract(b)
        idx, dist = m.knn2(fb, fa)
        oidx, odist = oracle.knn2(fb["descriptors"], fa["descriptors"])
        assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
        for ratio in (0.75, 0.7, 0.0, 1.0):
            got = m.match(fb, fa, None, ratio)
            want = oracle.match_ratio(fb["descriptors"], fa["descriptors"], ratio)
            assert got.tobytes() == want.tobytes(), "ratio %g" % ratio
        assert 200 <= len(m.match(fb, fa)) <= 2000
    finally:
        e.close()
        m.close()


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (31, 33), (64, 64), (65, 63), (257, 129), (300, 1), (2000, 1999),
                                   (513, 4095)])
def test_knn2_random_sets_with_ties(aria, oracle, nq, nt):
    """Raw descriptor sets (no extractor): sizes off every tile boundary of the matrix-core kernel (32-column /
    64-train tiles, 256-query workgroups), low-entropy descriptors so equal distances and duplicates are common
    (ties must go to the lower train index, like cv::BFMatcher's in-order scan), all-zero and all-one rows."""
    rng = np.random.default_rng(1000 * nq + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q[:, 4:] &= 0x11                      # few distinct values -> many ties
    t[:, 4:] &= 0x11
    if nt > 8:
        t[nt // 2] = t[1]                 # exact duplicates at different indices
        t[nt - 1] = t[1]
        t[3] = 0
        t[5] = 255
    if nq > 4:
        q[2] = 0
        q[3] = 255
        q[4] = t[min(1, nt - 1)]
    m = aria.HipMatcher()
    try:
        idx, dist = m.knn2({"descriptors": q}, {"descriptors": t})
        oidx, odist = oracle.knn2(q, t)
        assert np.array_equal(dist, odist)
        assert np.array_equal(idx, oidx)
        for ratio in (0.75, 0.0):
            got = m.match({"descriptors": q}, {"descriptors": t}, None, ratio)
            want = oracle.match_ratio(q, t, ratio)
            assert got.tobytes() == want.tobytes()
    finally:
        m.close()


@pytest.mark.parametrize("big", [False, True])
def test_batch_match_rows_beyond_narrow_layout(aria, oracle, torch_cuda, big):
    """Descriptor slots of 4160 rows (4000 features + slack) exceed the 12-bit train index of the narrow key layout, but
    the layout is chosen on the device from the actual counts: all pairs <= 4096 -> the 512-query narrow kernel, one pair
    above -> the wide kernel. 128 pairs (enough workgroups for the batch kernels); a few pairs against the oracle."""
    torch = torch_cuda
    dev = torch.device("cuda", 0)
    rows, n_pairs = 4160, 128
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    desc = torch.randint(0, 256, (n_pairs + 1, rows, 32), dtype=torch.uint8, device=dev, generator=g)
    desc[:, :, 8:] &= 0x0F                                   # low entropy: ties between train rows
    cnt = torch.full((n_pairs + 1,), 4000, dtype=torch.int32, device=dev)
    cnt[3] = 4096
    cnt[7] = 1
    if big:
        cnt[5] = 4100
    m = aria.HipMatcher(max_query=rows, max_train=rows)
    try:
        matches = torch.zeros((n_pairs, rows, 12), dtype=torch.uint8, device=dev)
        nm = torch.zeros((n_pairs,), dtype=torch.int32, device=dev)
        # pair p: query = slot p+1, train = slot p
        m.match_batch_device(desc.data_ptr() + rows * 32, cnt.data_ptr() + 4, desc, cnt, n_pairs, rows * 32, 0.75, matches, nm, rows)
        m.sync()
        c = cnt.cpu().numpy()
        for p in (0, 2, 3, 4, 5, 6, 7, n_pairs - 1):
            q = desc[p + 1, :c[p + 1]].cpu().numpy()
            t = desc[p, :c[p]].cpu().numpy()
            want = oracle.match_ratio(q, t, 0.75)
            n = int(nm[p].item())
            assert n == len(want), "pair %d" % p
            assert matches[p, :n].cpu().numpy().tobytes() == want.tobytes(), "pair %d" % p
    finally:
        m.close()


def test_match_train_resident_from_previous_call(aria, oracle):
    """aria_matcher_match keeps the previous call's query set on the device and skips the train upload when the train
    bytes equal it (frame i against frame i-1). Hits, misses (other set, same size with one byte changed, other size)
    and the slice merge of the latency schedule must all give the oracle's matches."""
    rng = np.random.default_rng(77)
    d = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for n in (700, 700, 1999, 2000, 2000)]
    for x in d:
        x[:, 6:] &= 0x33
    near = d[3].copy()
    near[-1, -1] ^= 1
    m = aria.HipMatcher()
    try:
        calls = [(d[1], d[0]), (d[2], d[1]), (d[3], d[2]), (d[4], near), (d[0], d[4]), (d[1], d[1]), (d[1], d[1]), (d[2], d[0])]
        # a steady stream of equal-sized frames, each matched against the one before: from the second sighting of a
        # size on, a ping-pong slot replays its captured graph (different data every time)
        chain = [d[3], d[4], near, d[3], d[4], near, d[3], d[4], near]
        calls += [(chain[i], chain[i - 1]) for i in range(1, len(chain))]
        calls += [(d[0], d[1]), (d[4], d[3]), (near, d[4])]            # size change in between, then back
        for q, t in calls:
            got = m.match({"descriptors": q}, {"descriptors": t}, None, 0.75)
            assert got.tobytes() == oracle.match_ratio(q, t, 0.75).tobytes()
    finally:
        m.close()


def _noise(seed, w, h, lo, hi):
    return np.random.default_rng(seed).integers(lo, hi, (h, w), dtype=np.uint8)


@pytest.mark.parametrize("kind,nf", [("noise", 1), ("noise", 7), ("noise", 120), ("noise", 2000), ("noise", 6000),
                                     ("narrow", 500), ("narrow", 3000), ("synth", 9), ("synth", 4000)])
def test_selection_orderings_bit_exact(aria, oracle, torch_cuda, kind, nf):
    """k_select orders the Harris keys by histogram bins + in-bin ranks (bitonic sort when a bin is too full or the kept set
    too large). Quotas from 1 to 6000, white noise (every pixel a FAST candidate: the largest key sets), low-contrast noise
    (responses within a few octaves: few, full bins) and the synthetic frames, through the single-frame entry point AND
    the batch entry point (two frames), against the oracle."""
    torch = torch_cuda
    w, h = 640, 480
    if kind == "noise":
        imgs = [_noise(100 + nf, w, h, 0, 256), _noise(200 + nf, w, h, 0, 256)]
    elif kind == "narrow":
        imgs = [_noise(300 + nf, w, h, 100, 160), _noise(400 + nf, w, h, 90, 170)]
    else:
        imgs = list(aria.synth_frame_pair(31, w, h))
    p = oracle.default_params(nf)
    want = [oracle.orb_extract(im, p) for im in imgs]
    e = aria.OrbHipExtractor(max_features=nf, max_width=w, max_height=h, max_batch=2)
    try:
        for im, (kps, desc) in zip(imgs, want):
            f = e.extract(im)
            assert f["keypoints"].tobytes() == kps.tobytes(), "single-frame keypoints differ"
            assert np.array_equal(f["descriptors"], desc)
        cap = max(e.kp_capacity(), max(len(k) for k, _ in want))
        dev = torch.device("cuda", 0)
        d_img = torch.from_numpy(np.stack(imgs)).to(dev)
        d_k = torch.zeros((2, cap, 24), dtype=torch.uint8, device=dev)
        d_d = torch.zeros((2, cap, 32), dtype=torch.uint8, device=dev)
        d_c = torch.zeros((2,), dtype=torch.int32, device=dev)
        e.extract_batch_device(d_img, 2, w, h, d_k, d_d, d_c, cap)
        e.check()
        c = d_c.cpu().numpy()
        for i, (kps, desc) in enumerate(want):
            assert c[i] == len(kps)
            assert d_k.cpu().numpy()[i, :c[i]].tobytes() == kps.tobytes(), "batch keypoints differ"
            assert np.array_equal(d_d.cpu().numpy()[i, :c[i]], desc)
    finally:
        e.close()


@pytest.mark.parametrize("w,h,nf", [(2047, 2047, 3000), (1919, 1083, 1500), (96, 2047, 200)])
def test_extract_bit_exact_large_and_limit_sizes(aria, oracle, w, h, nf):
    """The image-size limit of the candidate packing (2047 x 2047: widest staged rows, 32 x 32 selection tiles of 64 px),
    an odd HD-like size and a tall narrow strip, against the oracle."""
    a, _ = aria.synth_frame_pair(11, w, h)
    e = aria.OrbHipExtractor(max_features=nf, max_width=w, max_height=h)
    try:
        f = e.extract(a)
        kps, desc = oracle.orb_extract(a, oracle.default_params(nf))
        assert len(f["keypoints"]) == len(kps)
        assert f["keypoints"].tobytes() == kps.tobytes(), "keypoint records differ"
        assert np.array_equal(f["descriptors"], desc), "descriptors differ"
        assert e.slow_path_blocks() >= 0
    finally:
        e.close()
