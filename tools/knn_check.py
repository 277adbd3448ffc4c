#!/usr/bin/env python3
"""kNN-2 of raw descriptor sets on the GPU against a numpy brute force (no oracle needed): the host entry point, the
batched device entry point (which selects the 512-query workgroups once there are enough pairs), the keyframe-DB scan,
train sets on both sides of 4096 descriptors (12-bit / 16-bit index key layouts) and tie-heavy data. Exit status 1 on any mismatch.
Used by tests/test_gpu_variants.py under ARIA_KNN_NC=2/4 and ARIA_KNN_IMPL=valu."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aria_slam_amd as A  # noqa: E402

PC = np.array([bin(i).count("1") for i in range(256)], np.int64)


def brute(q, t):
    key = np.empty((len(q), len(t)), np.int64)
    for i0 in range(0, len(q), 64):
        d = PC[q[i0:i0 + 64, None, :] ^ t[None, :, :]].sum(-1)
        key[i0:i0 + 64] = d * 65536 + np.arange(len(t))[None, :]
    key.sort(axis=1)
    k = np.full((len(q), 2), -1, np.int64)
    k[:, :min(2, len(t))] = key[:, :2]
    idx = np.where(k >= 0, k & 0xFFFF, -1).astype(np.int32)
    dist = np.where(k >= 0, k >> 16, np.iinfo(np.int32).max).astype(np.int32)
    return idx, dist


def sets(rng, nq, nt, ties):
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    if ties:
        q[:, 3:] &= 0x11
        t[:, 3:] &= 0x11
        if nt > 8:
            t[nt // 2] = t[1]
            t[nt - 1] = t[1]
            t[3] = 0
            t[5] = 255
        if nq > 4:
            q[2] = 0
            q[3] = 255
            q[4] = t[min(1, nt - 1)]
    return q, t


def main():
    bad = 0
    # host entry point
    for nq, nt, ties, mt in [(64, 64, 0, 4096), (300, 200, 1, 4096), (500, 700, 0, 4096), (1000, 1, 1, 4096),
                             (70, 4096, 1, 8192), (70, 4097, 1, 8192), (100, 17000, 1, 20000), (40, 65535, 0, 65535)]:
        rng = np.random.default_rng(nq * 7 + nt)
        q, t = sets(rng, nq, nt, ties)
        m = A.HipMatcher(max_query=4096, max_train=mt)
        idx, dist = m.knn2({"descriptors": q}, {"descriptors": t})
        bi, bd = brute(q, t)
        nbad = int(((idx != bi).any(1) | (dist != bd).any(1)).sum())
        print("host knn2 %5d x %5d: %s" % (nq, nt, "OK" if nbad == 0 else "%d bad rows" % nbad))
        bad += nbad
        m.close()
    # batched device entry point + DB scan (needs torch for device memory)
    import torch
    dev = torch.device("cuda:0")
    for n_pairs, cap in [(3, 640), (1100, 640)]:
        rng = np.random.default_rng(n_pairs)
        nq = rng.integers(0, cap + 1, n_pairs).astype(np.int32)
        nt = rng.integers(0, cap + 1, n_pairs).astype(np.int32)
        nq[0], nt[0] = cap, cap
        if n_pairs > 2:
            nq[1], nt[2] = 0, 0
        Q = rng.integers(0, 256, (n_pairs, cap, 32), dtype=np.uint8)
        T = rng.integers(0, 256, (n_pairs, cap, 32), dtype=np.uint8)
        Q[:, :, 2:] &= 0x11
        T[:, :, 2:] &= 0x11
        m = A.HipMatcher()
        dQ, dT = torch.from_numpy(Q).to(dev), torch.from_numpy(T).to(dev)
        dnq, dnt = torch.from_numpy(nq).to(dev), torch.from_numpy(nt).to(dev)
        dM = torch.zeros((n_pairs, cap, 3), dtype=torch.int32, device=dev)
        dN = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
        m.match_batch_device(dQ, dnq, dT, dnt, n_pairs, cap * 32, 0.75, dM, dN, cap)
        m.sync()
        M = dM.cpu().numpy().view(A.MATCH_DTYPE).reshape(n_pairs, cap)
        N = dN.cpu().numpy()
        nbad = 0
        check = range(n_pairs) if n_pairs <= 8 else list(range(0, n_pairs, 97)) + [n_pairs - 1]
        for p in check:
            want = []
            if nq[p] and nt[p]:
                bi, bd = brute(Q[p, :nq[p]], T[p, :nt[p]])
                for i in range(nq[p]):
                    if bi[i, 1] >= 0 and np.float32(bd[i, 0]) < np.float32(0.75) * np.float32(bd[i, 1]):
                        want.append((i, bi[i, 0], float(bd[i, 0])))
            got = [(int(r["query_idx"]), int(r["train_idx"]), float(r["distance"])) for r in M[p, :N[p]]]
            nbad += 0 if got == want else 1
        print("match_batch_device %4d pairs: %s" % (n_pairs, "OK" if nbad == 0 else "%d bad pairs" % nbad))
        bad += nbad
        # keyframe-DB scan: query = pair 0's queries, DB = all train sets
        dG = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
        m.match_db_device(dQ, int(nq[0]), dT, dnt, n_pairs, cap * 32, 0.7, dG)
        m.sync()
        G = dG.cpu().numpy()
        nbad = 0
        for p in (range(n_pairs) if n_pairs <= 8 else range(0, n_pairs, 211)):
            g = 0
            if nt[p]:
                bi, bd = brute(Q[0, :nq[0]], T[p, :nt[p]])
                g = int(sum(1 for i in range(nq[0]) if bi[i, 1] >= 0 and
                            float(np.float32(bd[i, 0])) < 0.7 * float(np.float32(bd[i, 1]))))
            nbad += 0 if g == G[p] else 1
        print("match_db_device    %4d keyframes: %s" % (n_pairs, "OK" if nbad == 0 else "%d bad" % nbad))
        bad += nbad
        m.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
