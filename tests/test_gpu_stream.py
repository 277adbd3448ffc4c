"""k_fast_blur_stream (fast_blur_stream.hip), the streaming FAST/blur kernel of the batch entry point, against the oracle.

The single-frame entry points keep k_fast_blur_band, so tests that compare a batch with per-frame calls already compare
the two kernels with each other; here the batch path alone is held against the CPU restatement: raw and blurred levels
byte for byte, keypoints and descriptors of every frame, for level widths of every residue mod 4 (mirrored partial
dwords, tail columns of the column filter), frames that share waves (narrow levels), several frames per wave, a
corner-dense image (queue rounds, corner-list overflow -> dense NMS) and blur_tie_mode 0."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def _batch(aria, torch, imgs, nf, tie_mode=1, max_batch=None):
    n, h, w = imgs.shape
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    e = aria.OrbHipExtractor(max_features=nf, max_width=w, max_height=h, max_batch=max_batch or n, blur_tie_mode=tie_mode,
                             stream=s.cuda_stream)
    cap = e.kp_capacity()
    d_img = torch.from_numpy(np.ascontiguousarray(imgs)).to(dev)
    kps = torch.zeros((n, cap, 24), dtype=torch.uint8, device=dev)
    desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
    cnt = torch.zeros((n,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    e.extract_batch_device(d_img, n, w, h, kps, desc, cnt, cap)
    e.check()
    return e, cnt.cpu().numpy(), kps.cpu().numpy(), desc.cpu().numpy()


def _check_frames(oracle, imgs, nf, cnt, kps, desc, tie_mode=1):
    for f, im in enumerate(imgs):
        ok, od = oracle.orb_extract(im, oracle.default_params(nf, tie_mode), cap=200000)
        assert cnt[f] == len(ok), "frame %d: %d keypoints, oracle %d" % (f, cnt[f], len(ok))
        assert kps[f, :cnt[f]].tobytes() == ok.tobytes(), "frame %d keypoints" % f
        assert np.array_equal(desc[f, :cnt[f]], od), "frame %d descriptors" % f


@pytest.mark.parametrize("w,h,nf,nframes", [(640, 480, 2000, 3), (752, 480, 1000, 2), (333, 251, 300, 5), (258, 200, 300, 7),
                                            (131, 97, 100, 9), (1408, 1408, 4000, 1), (643, 211, 500, 2)])
def test_stream_levels_and_frames_equal_oracle(aria, oracle, torch_cuda, w, h, nf, nframes):
    imgs = np.stack([aria.synth_frame_pair(40 + i, w, h)[i & 1] for i in range(nframes)])
    e, cnt, kps, desc = _batch(aria, torch_cuda, imgs, nf)
    try:
        p = oracle.default_params(nf)
        raw = oracle.build_pyramid(imgs[0], p)
        blur = oracle.blur_pyramid(raw, p, w, h)
        info = e.level_info(w, h)
        for l in range(8):
            lw, lh = info[l][0], info[l][1]
            g_raw = e.debug_read_level(l, False, lw, lh)
            g_blur = e.debug_read_level(l, True, lw, lh)
            assert np.array_equal(g_raw, raw[l]), "raw level %d (%dx%d) differs at %d px" % (l, lw, lh, np.count_nonzero(g_raw != raw[l]))
            assert np.array_equal(g_blur, blur[l]), "blurred level %d (%dx%d) differs at %d px" % (l, lw, lh, np.count_nonzero(g_blur != blur[l]))
        _check_frames(oracle, imgs, nf, cnt, kps, desc)
    finally:
        e.close()


def test_stream_chunks_and_last_frame(aria, oracle, torch_cuda):
    """11 frames in passes of 4 (the last pass holds 3): the wave that holds the end of the virtual row is partly empty."""
    imgs = np.stack([aria.synth_frame_pair(60 + i, 320, 240)[0] for i in range(11)])
    e, cnt, kps, desc = _batch(aria, torch_cuda, imgs, 400, max_batch=4)
    try:
        _check_frames(oracle, imgs, 400, cnt, kps, desc)
    finally:
        e.close()


def test_stream_corner_dense_images(aria, oracle, torch_cuda):
    """Noise: a third of the pixels pass the compass test and thousands are corners -- the survivor queue goes through
    several rounds per group and the corner list overflows into the dense NMS scan. A dot grid is a tie storm on top."""
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (2, 240, 320), dtype=np.uint8)
    yy, xx = np.mgrid[0:240, 0:320]
    dots = np.where(((xx % 7) == 3) & ((yy % 7) == 3), 255, 0).astype(np.uint8)
    imgs = np.concatenate([noise, dots[None]])
    e, cnt, kps, desc = _batch(aria, torch_cuda, imgs[:2], 500)
    try:
        _check_frames(oracle, imgs[:2], 500, cnt, kps, desc)
    finally:
        e.close()


@pytest.mark.parametrize("w,h", [(640, 480), (333, 251)])
def test_stream_tie_mode_0(aria, oracle, torch_cuda, w, h):
    imgs = np.stack([aria.synth_frame_pair(70 + i, w, h)[0] for i in range(2)])
    e, cnt, kps, desc = _batch(aria, torch_cuda, imgs, 600, tie_mode=0)
    try:
        p = oracle.default_params(600, 0)
        raw = oracle.build_pyramid(imgs[0], p)
        blur = oracle.blur_pyramid(raw, p, w, h)
        info = e.level_info(w, h)
        for l in range(8):
            g_blur = e.debug_read_level(l, True, info[l][0], info[l][1])
            assert np.array_equal(g_blur, blur[l]), "blurred level %d differs at %d px" % (l, np.count_nonzero(g_blur != blur[l]))
        _check_frames(oracle, imgs, 600, cnt, kps, desc, tie_mode=0)
    finally:
        e.close()


# ---- oracle exposure switches: the kernels follow the parameters ------------------------------------
def _tie_image(aria, w, h, seed):
    """Upper half: whole columns of exact blur ties (tests/test_oracle_known_answers.py tie_stripes); lower half: a synthetic
    frame, so that keypoints exist too."""
    pat = np.array([127, 128, 128, 126, 128, 128, 128], np.uint8)
    img = aria.synth_frame_pair(seed, w, h)[0].copy()
    img[: h // 2] = pat[np.arange(w) % 7][None, :]
    return img


@pytest.mark.parametrize("tie_mode", [0, 1, 2, 3])
@pytest.mark.parametrize("w,h", [(645, 300), (333, 251)])
def test_blur_tie_modes_follow_the_parameter(aria, oracle, torch_cuda, tie_mode, w, h):
    """blur_tie_mode 0..3 on an image with many exact ties, through the batch entry point (k_fast_blur_stream) and the
    single-frame one (k_fast_blur_band): blurred level 0 and the extraction equal the oracle run with the same mode, and the
    modes really differ on this image."""
    imgs = np.stack([_tie_image(aria, w, h, 80), _tie_image(aria, w, h, 81)])
    p = oracle.default_params(400, tie_mode)
    want_blur = oracle.blur_pyramid(oracle.build_pyramid(imgs[0], p), p, w, h)
    if tie_mode in (1, 2, 3):
        other = oracle.default_params(400, 0)
        assert not np.array_equal(oracle.blur_pyramid(oracle.build_pyramid(imgs[0], other), other, w, h)[0], want_blur[0])
    e, cnt, kps, desc = _batch(aria, torch_cuda, imgs, 400, tie_mode=tie_mode)
    try:
        assert e.fast_blur_kernel() == "k_fast_blur_stream"
        info = e.level_info(w, h)
        for l in range(8):
            got = e.debug_read_level(l, True, info[l][0], info[l][1])
            assert np.array_equal(got, want_blur[l]), "batch, level %d: %d px differ" % (l, np.count_nonzero(got != want_blur[l]))
        _check_frames(oracle, imgs, 400, cnt, kps, desc, tie_mode=tie_mode)
        f = e.extract(imgs[1])                                   # single-frame schedule on the same handle
        assert e.fast_blur_kernel() == "k_fast_blur_band"
        ok, od = oracle.orb_extract(imgs[1], p, cap=200000)
        assert f["keypoints"].tobytes() == ok.tobytes() and np.array_equal(f["descriptors"], od)
        want1 = oracle.blur_pyramid(oracle.build_pyramid(imgs[1], p), p, w, h)
        for l in range(8):
            got = e.debug_read_level(l, True, info[l][0], info[l][1])
            assert np.array_equal(got, want1[l]), "single frame, level %d" % l
    finally:
        e.close()


@pytest.mark.parametrize("w,h", [(558, 300), (640, 117)])
def test_level_size_mode_follows_the_parameter(aria, oracle, torch_cuda, w, h):
    """Sizes from profiles/level_size_sweep.txt where cvRound(dim * (1.0f / scale)) != cvRound(dim / scale) (558 at level 2,
    117 at level 1): with level_size_mode 1 the library's levels and keypoints equal the oracle's in that mode."""
    imgs = np.stack([aria.synth_frame_pair(90 + i, w, h)[0] for i in range(2)])
    p0, p1 = oracle.default_params(300, 1, 0), oracle.default_params(300, 1, 1)
    assert oracle.level_sizes(p0, w, h) != oracle.level_sizes(p1, w, h)
    n, dev = len(imgs), torch_cuda.device("cuda", 0)
    s = torch_cuda.cuda.Stream(device=dev)
    e = aria.OrbHipExtractor(max_features=300, max_width=w, max_height=h, max_batch=n, level_size_mode=1, stream=s.cuda_stream)
    try:
        cap = e.kp_capacity()
        d_img = torch_cuda.from_numpy(imgs).to(dev)
        kps = torch_cuda.zeros((n, cap, 24), dtype=torch_cuda.uint8, device=dev)
        desc = torch_cuda.zeros((n, cap, 32), dtype=torch_cuda.uint8, device=dev)
        cnt = torch_cuda.zeros((n,), dtype=torch_cuda.int32, device=dev)
        torch_cuda.cuda.synchronize()
        e.extract_batch_device(d_img, n, w, h, kps, desc, cnt, cap)
        e.check()
        c, k, d = cnt.cpu().numpy(), kps.cpu().numpy(), desc.cpu().numpy()
        for f in range(n):
            ok, od = oracle.orb_extract(imgs[f], p1)
            assert c[f] == len(ok) and k[f, :c[f]].tobytes() == ok.tobytes() and np.array_equal(d[f, :c[f]], od)
        f1 = e.extract(imgs[0])
        ok, od = oracle.orb_extract(imgs[0], p1)
        assert f1["keypoints"].tobytes() == ok.tobytes() and np.array_equal(f1["descriptors"], od)
        ok0, _ = oracle.orb_extract(imgs[0], p0)
        assert ok0.tobytes() != ok.tobytes()                      # the switch matters at this size
    finally:
        e.close()


def test_extraction_beside_the_matcher_equals_extraction_alone(aria, torch_cuda):
    """bench.py's two-stream schedule: the FAST/blur launches of one step run while the matcher's MFMA waves of the previous
    step share the SIMDs. A batch extracted that way must equal the same batch extracted alone, byte for byte (round 3: a
    float2 formulation of the blur's column pass -- v_pk_add_f32 with op_sel-swizzled register pairs -- produced a few
    wrong blurred pixels per ~1000 frames ONLY in this situation; tools/race_probe.py prints what differs)."""
    torch = torch_cuda
    W, H, NF, B = 640, 480, 2000, 2048
    dev = torch.device("cuda", 0)
    host = torch.empty((B, H, W), dtype=torch.uint8)
    aria.synth_sequence(1, B // 2, W, H, out=host.numpy())
    img = host.to(dev)
    se, sm = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    e = aria.OrbHipExtractor(max_features=NF, max_width=W, max_height=H, max_batch=B, stream=se.cuda_stream)
    m = aria.HipMatcher(stream=sm.cuda_stream, max_query=4096, max_train=4096)
    try:
        cap = e.kp_capacity()

        def bufs():
            return (torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev),
                    torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
        k0, d0, c0 = bufs()
        with torch.cuda.stream(se):
            e.extract_batch_device(img, B, W, H, k0, d0, c0, cap)
        torch.cuda.synchronize()
        assert e.fast_blur_kernel() == "k_fast_blur_stream"
        mt = torch.empty((B, cap, 12), dtype=torch.uint8, device=dev)
        nm = torch.zeros(B, dtype=torch.int32, device=dev)
        k1, d1, c1 = bufs()
        for rep in range(3):
            with torch.cuda.stream(sm):
                for _ in range(4):          # long enough to run beside select and describe too
                    m.match_batch_device(d0.data_ptr() + cap * 32, c0.data_ptr() + 4, d0, c0, B - 1, cap * 32, 0.75,
                                         mt.data_ptr() + cap * 12, nm.data_ptr() + 4, cap)
            with torch.cuda.stream(se):
                e.extract_batch_device(img, B, W, H, k1, d1, c1, cap)
            torch.cuda.synchronize()
            e.check()
            assert torch.equal(c0, c1), rep
            assert torch.equal(k0, k1), rep
            bad = (d0 != d1).flatten(1).any(dim=1).nonzero().flatten().tolist()
            assert not bad, (rep, bad[:8])
    finally:
        e.close()
        m.close()
