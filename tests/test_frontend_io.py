"""the SlamPipeline front half over the ports (FrontEnd) and the ASL/EuRoC input format
(CSV + PNG reader, euroc_eval-style driver)."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "aria_slam_amd")


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def write_png(img, filters="mixed", level=6):
    """Minimal PNG encoder (8-bit gray HxW or RGB/RGBA HxWxC) that exercises every scanline filter type."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    rows = img.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        cur = rows[y]
        ft = (y % 5) if filters == "mixed" else int(filters)
        a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - a
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((a + prev) >> 1)
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            f = cur - pred
        raw.append(ft)
        raw += (f & 0xFF).astype(np.uint8).tobytes()
        prev = cur
    comp = zlib.compress(bytes(raw), level)
    idat = b"".join(_chunk(b"IDAT", comp[i:i + 8000]) for i in range(0, len(comp), 8000))     # several IDAT chunks
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
            _chunk(b"tEXt", b"Comment\x00synthetic") + idat + _chunk(b"IEND", b""))


@pytest.fixture(scope="module")
def hostlib(aria):
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host"), "-s"])
    aria.load_library()
    L = C.CDLL(os.path.join(PKG, "libaria_hip_adapters.so"))
    L.aria_asl_decode_png_gray.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.aria_asl_list.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    return L


def _decode(L, png):
    out = np.zeros(2048 * 2048, np.uint8)
    w, h = C.c_int(), C.c_int()
    buf = np.frombuffer(png, np.uint8)
    rc = L.aria_asl_decode_png_gray(buf.ctypes.data, len(png), out.ctypes.data, out.size, C.byref(w), C.byref(h))
    return rc, (out[:w.value * h.value].reshape(h.value, w.value).copy() if rc == 0 else None)


def test_png_decoder_all_filters_and_colour_types(aria, hostlib):
    a, _ = aria.synth_frame_pair(5, 752, 480)
    for filt in ("mixed", 0, 1, 2, 3, 4):
        rc, g = _decode(hostlib, write_png(a, filt))
        assert rc == 0 and np.array_equal(g, a), filt
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    want = ((rgb[..., 0].astype(int) * 4899 + rgb[..., 1].astype(int) * 9617 + rgb[..., 2].astype(int) * 1868 + 8192) >> 14).astype(np.uint8)
    rc, g = _decode(hostlib, write_png(rgb))
    assert rc == 0 and np.array_equal(g, want)
    rgba = np.concatenate([rgb, rng.integers(0, 256, (37, 53, 1), dtype=np.uint8)], 2)
    rc, g = _decode(hostlib, write_png(rgba))
    assert rc == 0 and np.array_equal(g, want)
    ga = np.stack([a[:20, :30], a[20:40, :30]], 2)
    rc, g = _decode(hostlib, write_png(ga))
    assert rc == 0 and np.array_equal(g, a[:20, :30])
    # malformed inputs fail cleanly
    good = write_png(a[:16, :16])
    for bad in (good[:20], b"\x00" * 64, good[:-20], good.replace(b"IHDR", b"IHDX")):
        assert _decode(hostlib, bad)[0] != 0


def _make_dataset(aria, root, n_pairs, w=320, h=240, shuffle=True, revisit=0):
    cam = os.path.join(root, "mav0", "cam0", "data")
    os.makedirs(cam, exist_ok=True)
    seq = aria.synth_sequence(40, n_pairs, w, h)
    if revisit:                                                     # the sequence comes back to its first frames: a loop
        seq = np.concatenate([seq, seq[:revisit]])
    t0 = 1403636579763555584
    rows = []
    for i in range(len(seq)):
        ts = t0 + i * 50_000_000                                   # 20 Hz (H07_EUROC_DATASET_AUDIT.md:16)
        name = "%d.png" % ts
        open(os.path.join(cam, name), "wb").write(write_png(seq[i]))
        rows.append("%d,%s" % (ts, name))
    if shuffle:
        rows = rows[3:] + rows[:3]                                 # the reader must sort by timestamp
    with open(os.path.join(root, "mav0", "cam0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],filename\n" + "\n".join(rows) + "\n\n# trailing comment\n")
    return seq, t0


def test_asl_sequence_listing(aria, hostlib, tmp_path):
    seq, t0 = _make_dataset(aria, str(tmp_path), 4)
    ts = np.zeros(64, np.float64)
    first = C.create_string_buffer(512)
    n = hostlib.aria_asl_list(str(tmp_path).encode(), ts.ctypes.data, 64, first, 512)
    assert n == 8
    assert np.all(np.diff(ts[:n]) > 0) and abs(ts[0] - t0 * 1e-9) < 1e-3 and abs((ts[1] - ts[0]) - 0.05) < 1e-6
    assert first.value.decode().endswith("%d.png" % t0)
    assert hostlib.aria_asl_list(os.path.join(str(tmp_path), "mav0").encode(), ts.ctypes.data, 64, first, 512) == 8
    assert hostlib.aria_asl_list(b"/nonexistent", ts.ctypes.data, 64, first, 512) == -1


def test_frontend_over_mock_ports():
    exe = os.path.join(ROOT, "tests", "cpp", "adapter_selftest")
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host"), "-s"])
    src = os.path.join(ROOT, "tests", "cpp", "adapter_selftest.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(PKG, "host", "include"), src, "-o", exe, "-L" + PKG,
                           "-laria_hip_adapters", "-laria_orb_hip", "-lz", "-Wl,-rpath," + PKG])
    out = subprocess.run([exe, "frontend_mock"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "OK frontend_mock" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("legacy", [False, True])
def test_euroc_frontend_driver_on_synthetic_asl(aria, hostlib, tmp_path, legacy):
    seq, _ = _make_dataset(aria, str(tmp_path), 6)
    csv = os.path.join(str(tmp_path), "out.csv")
    exe = os.path.join(PKG, "euroc_frontend")
    cmd = [exe, str(tmp_path), "800", "--csv", csv] + (["--legacy-order"] if legacy else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l.split(",") for l in open(csv).read().strip().split("\n")[1:]]
    assert len(rows) == len(seq)
    e = aria.OrbHipExtractor(max_features=800, max_width=320, max_height=240)
    m = aria.HipMatcher()
    try:
        prev = None
        for i, img in enumerate(seq):
            f = e.extract(img)
            nm = 0
            if prev is not None:
                nm = len(m.match(prev, f) if legacy else m.match(f, prev))
            assert int(rows[i][2]) == len(f["keypoints"]) and int(rows[i][3]) == nm, i
            prev = f
    finally:
        e.close()
        m.close()
    assert "mean_matches" in out.stdout


def _fnv1a(parts):
    h = 14695981039346656037
    for p in parts:
        for b in np.ascontiguousarray(p).view(np.uint8).reshape(-1).tolist():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
def test_euroc_frontend_hashes_equal_the_oracle_at_euroc_size(aria, hostlib, oracle, tmp_path):
    """BASELINE.json configs[0]: 752x480 mono, 1000 kp/frame through the euroc_eval-equivalent driver (PipelineFactory's
    HIP mode -> FrontEnd -> C++ adapters -> C-ABI). Every frame's keypoint records, descriptor rows and match records
    (query = current, train = previous) hash to what the ORACLE produces for the same PNG files."""
    seq, _ = _make_dataset(aria, str(tmp_path), 5, w=752, h=480)
    csv = os.path.join(str(tmp_path), "out.csv")
    out = subprocess.run([os.path.join(PKG, "euroc_frontend"), str(tmp_path), "1000", "--csv", csv], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l.split(",") for l in open(csv).read().strip().split("\n")[1:]]
    assert len(rows) == len(seq)
    p = oracle.default_params(1000)
    prev = None
    for i, img in enumerate(seq):
        k, d = oracle.orb_extract(img, p)
        m = oracle.match_ratio(d, prev, 0.75) if prev is not None else np.zeros(0, oracle.MATCH_DTYPE)
        assert int(rows[i][2]) == len(k) and int(rows[i][3]) == len(m), i
        assert int(rows[i][4]) == _fnv1a([k, d, m]), i
        prev = d


@pytest.mark.gpu
def test_euroc_frontend_loop_closure_step(aria, hostlib, oracle, tmp_path):
    """--loop: keyframe insertion and the loop query of src/euroc_eval.cpp:103, 230-247 over ILoopDetector, database in
    HBM. The sequence returns to its first frames after 460 frames (about half of them become keyframes); the detector (min_frames_between 200, min_score 0.4,
    min_matches 50) must report exactly the loops the oracle's restatement of LoopClosureDetector::findCandidates +
    the same acceptance rule gives."""
    n_pairs, revisit = 230, 6
    seq, _ = _make_dataset(aria, str(tmp_path), n_pairs, w=320, h=240, revisit=revisit)
    csv = os.path.join(str(tmp_path), "out.csv")
    out = subprocess.run([os.path.join(PKG, "euroc_frontend"), str(tmp_path), "500", "--csv", csv, "--loop"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l.split(",") for l in open(csv).read().strip().split("\n")[1:]]
    assert len(rows) == len(seq) == 2 * n_pairs + revisit
    p = oracle.default_params(500)
    descs, kf = [], []                       # kf: (frame id, descriptors)
    want_loops = {}
    for i, img in enumerate(seq):
        k, d = oracle.orb_extract(img, p)
        nm = len(oracle.match_ratio(d, descs[-1], 0.75)) if descs else 0
        descs.append(d)
        is_kf = i > 0 and nm >= 8
        assert int(rows[i][5]) == int(is_kf), i
        if not is_kf:
            continue
        if len(kf) >= 200:                                           # LoopClosure.cpp:34-36
            ci, cs = oracle.loop_candidates(d, i, [x[1] for x in kf], [x[0] for x in kf], 200)
            for idx, score in zip(ci.tolist(), cs.tolist()):
                if score < 0.4:
                    continue
                if len(oracle.match_ratio(d, kf[idx][1], 0.7)) < 50:
                    continue
                want_loops[i] = (kf[idx][0], score)
                break
        kf.append((i, d))
        kf = kf[-500:]
    got_loops = {i: (int(r[6]), float(r[7])) for i, r in enumerate(rows) if int(r[6]) >= 0}
    assert want_loops, "the test sequence must contain a loop"
    assert set(got_loops) == set(want_loops)
    for i in want_loops:
        assert got_loops[i][0] == want_loops[i][0] and abs(got_loops[i][1] - want_loops[i][1]) < 1e-6


@pytest.mark.gpu
def test_euroc_frontend_shards_equal_the_single_shard_run(aria, hostlib, tmp_path):
    """euroc_frontend --devices N --shards K: K host threads, each with its own extractor + matcher
    handles over a contiguous frame range with a one-frame halo, results merged in frame order; the loop-closure step
    consumes the merged stream. Every CSV row (counts, per-frame hash over keypoints + descriptors + matches, keyframe flag,
    loop id and score) must equal the single-shard run's. One GPU here: the K shards are logical shards on device 0, which
    also exercises K handles from K host threads at once."""
    n_pairs, revisit = 230, 6                                      # as in the loop-closure test: the sequence contains a loop
    seq, _ = _make_dataset(aria, str(tmp_path), n_pairs, w=320, h=240, revisit=revisit)
    exe = os.path.join(PKG, "euroc_frontend")
    runs = {}
    for name, extra in (("one", []), ("k3", ["--devices", "1", "--shards", "3"]), ("k5", ["--shards", "5"])):
        csv = os.path.join(str(tmp_path), name + ".csv")
        out = subprocess.run([exe, str(tmp_path), "500", "--csv", csv, "--loop"] + extra, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout + out.stderr
        runs[name] = open(csv).read().strip().split("\n")
        if extra:
            assert "%s shards on 1 device" % extra[-1] in out.stdout
    assert len(runs["one"]) == len(seq) + 1
    assert runs["k3"] == runs["one"]
    assert runs["k5"] == runs["one"]
    assert sum(int(r.split(",")[5]) for r in runs["one"][1:]) > 50          # keyframes exist, so the loop step did run
    assert any(int(r.split(",")[6]) >= 0 for r in runs["one"][1:])          # ... and found the loop


@pytest.mark.gpu
def test_euroc_frontend_batch_mode_equals_the_frame_at_a_time_run(aria, hostlib, tmp_path):
    """euroc_frontend --batch B: chunks of B frames through aria_orb_extract_batch_device +
    aria_matcher_match_batch_device (decode workers -> pinned ring -> copy stream -> batch kernels -> result stream), the
    descriptors of a chunk's last frame carried to the next chunk on the device. BASELINE.json configs[0] size: 752x480,
    1000 kp. Every CSV row (counts, hash over keypoint records + descriptor rows + match records, keyframe flag, loop id and
    score) must equal the frame-at-a-time run's -- for one shard and three, for a chunk size that does not divide the
    sequence, for chunks of one frame, and in the legacy query/train order."""
    seq, _ = _make_dataset(aria, str(tmp_path), 19, w=752, h=480, revisit=3)          # 41 frames
    exe = os.path.join(PKG, "euroc_frontend")

    def run(name, extra):
        csv = os.path.join(str(tmp_path), name + ".csv")
        out = subprocess.run([exe, str(tmp_path), "1000", "--csv", csv, "--loop"] + extra, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout + out.stderr
        return open(csv).read().strip().split("\n"), out.stdout

    one, _ = run("one", [])
    assert len(one) == len(seq) + 1
    for name, extra in (("b16", ["--batch", "16"]), ("b16k3", ["--batch", "16", "--shards", "3", "--decode-threads", "2"]),
                        ("b7", ["--batch", "7"]), ("b1", ["--batch", "1"]), ("b64", ["--batch", "64"])):
        got, stdout = run(name, extra)
        assert got == one, name
        assert "extract+match kernels" in stdout and "decode" in stdout and "H2D" in stdout, stdout
    legacy, _ = run("legacy", ["--legacy-order"])
    got, _ = run("legacy_b16", ["--legacy-order", "--batch", "16", "--shards", "2"])
    assert got == legacy and legacy != one
    assert sum(int(r.split(",")[3]) for r in one[1:]) > 1000                       # the pairs do match


def test_runtime_helpers_fail_loudly_without_a_device(aria):
    """ABI 4 (device memory, staging copies, events): exported, and on a machine without a GPU they report
    ARIA_E_NO_DEVICE / ARIA_E_HIP instead of pretending -- the batched C++ driver has no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = aria.load_library()
    n = C.c_int(-1)
    assert L.aria_device_count(C.byref(n)) == 0 and n.value == 0
    p = C.c_void_p()
    assert L.aria_device_alloc(0, C.c_size_t(1024), C.byref(p)) != 0 and not p.value
    assert L.aria_host_alloc_pinned(C.c_size_t(1024), C.byref(p)) != 0 and not p.value
    assert L.aria_event_create(0, C.byref(p)) != 0 and not p.value
    exe = os.path.join(PKG, "euroc_frontend")
    if os.path.exists(exe):
        out = subprocess.run([exe, "/nonexistent", "--batch", "8"], capture_output=True, text=True, timeout=60)
        assert out.returncode != 0 and "HIP device" in out.stderr


@pytest.mark.gpu
def test_euroc_frontend_batch_mode_reports_a_bad_sequence(aria, hostlib, tmp_path):
    """Error behaviour of the chunked pipeline: an image of another size in the middle of the sequence (the batch entry point
    takes one size per call) stops the run with a message naming the file and a non-zero exit code -- from the producer
    thread, through the consumer, without hanging; the frame-at-a-time mode accepts such a sequence (the adapter re-creates
    its handle, as OrbCudaExtractor would re-allocate its GpuMat)."""
    seq, t0 = _make_dataset(aria, str(tmp_path), 6, w=320, h=240, shuffle=False)
    cam = os.path.join(str(tmp_path), "mav0", "cam0", "data")
    odd = "%d.png" % (t0 + 7 * 50_000_000)
    open(os.path.join(cam, odd), "wb").write(write_png(np.zeros((200, 300), np.uint8) + 7))
    exe = os.path.join(PKG, "euroc_frontend")
    out = subprocess.run([exe, str(tmp_path), "500", "--batch", "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and odd in out.stderr and "another size" in out.stderr, out.stdout + out.stderr
    out = subprocess.run([exe, str(tmp_path), "500"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
