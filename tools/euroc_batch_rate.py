#!/usr/bin/env python3
"""euroc_frontend frame-at-a-time against --batch on a synthetic ASL tree of EuRoC-sized frames (752x480, 1000 kp):
the summary lines of both (frames/s; for the batch run decode, staging and kernel time apart). GPU box only.
usage: euroc_batch_rate.py [frames=1024] [batch=256] [decode_threads=8] [shards=1]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np   # noqa: E402
import aria_slam_amd as A   # noqa: E402
from test_frontend_io import write_png   # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
shards = int(sys.argv[4]) if len(sys.argv) > 4 else 1
W, H = 752, 480
A.load_library()
subprocess.check_call(["make", "-C", os.path.join(ROOT, "aria_slam_amd", "host"), "-s"])
root = tempfile.mkdtemp(prefix="asl_")
cam = os.path.join(root, "mav0", "cam0", "data")
os.makedirs(cam)
t_enc = time.time()
# 64 distinct frames, cycled: the encoder (python) is the slow part here, the decoder sees every file
seq = A.synth_sequence(40, 32, W, H)
pngs = [write_png(seq[i], filters="1", level=1) for i in range(len(seq))]
rows = []
for i in range(frames):
    ts = 1403636579763555584 + i * 50_000_000
    open(os.path.join(cam, "%d.png" % ts), "wb").write(pngs[i % len(pngs)])
    rows.append("%d,%d.png" % (ts, ts))
open(os.path.join(root, "mav0", "cam0", "data.csv"), "w").write("#timestamp [ns],filename\n" + "\n".join(rows) + "\n")
print("dataset: %d frames of %dx%d, %.1f KB per PNG, written in %.1f s" % (frames, W, H, np.mean([len(p) for p in pngs]) / 1e3, time.time() - t_enc))
exe = os.path.join(ROOT, "aria_slam_amd", "euroc_frontend")
runs = [("frame-at-a-time", ["--shards", str(shards)]),
        ("batch %d, %d decode threads" % (batch, threads), ["--batch", str(batch), "--decode-threads", str(threads), "--shards", str(shards)]),
        ("batch %d, %d decode threads" % (batch, 2 * threads), ["--batch", str(batch), "--decode-threads", str(2 * threads), "--shards", str(shards)])]
csvs = []
for name, extra in runs:
    csv = os.path.join(root, "out%d.csv" % len(csvs))
    t0 = time.time()
    out = subprocess.run([exe, root, "1000", "--csv", csv] + extra, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    csvs.append(open(csv).read())
    print("== %s (%.2f s)" % (name, time.time() - t0))
    for line in out.stdout.strip().split("\n"):
        if line.startswith("frames ") or line.startswith("batch "):
            print("   " + line)
assert csvs[1] == csvs[0] and csvs[2] == csvs[0], "batch CSV differs from the frame-at-a-time CSV"
print("per-frame CSVs identical")
