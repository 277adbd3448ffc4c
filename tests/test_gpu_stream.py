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
