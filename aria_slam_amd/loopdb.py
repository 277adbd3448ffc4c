"""Keyframe-descriptor database for loop-closure candidate search (BASELINE.json configs[4]).

Replaces the CPU loop of the reference's LoopClosureDetector::findCandidates (src/legacy/LoopClosure.cpp:72-114):
each rank keeps its shard's keyframe descriptors resident in HBM in fixed-capacity slots; an optional RCCL
all-gather over xGMI (torch.distributed, backend "nccl") gives every rank the whole database, after which
candidates are scored locally with the device matcher (aria_matcher_match_db_device) and the reference's host rules.
"""
import numpy as np

MAX_KEYFRAMES = 500          # LoopClosure.cpp:28-30 database cap


def score_candidates(good, nq, query_id, kf_ids, kf_counts, min_frames_between):
    """LoopClosure.cpp:79-111: skip recent/empty keyframes, score = good / max(1, |query kps|), keep > 0.1,
    sort by score descending (ties keep database order), top 5. Returns [(db_index, score)]."""
    out = []
    if nq <= 0:
        return out
    for i, (g, kid, cnt) in enumerate(zip(good, kf_ids, kf_counts)):
        if int(query_id) - int(kid) < min_frames_between:
            continue
        if int(cnt) <= 0:
            continue
        score = float(int(g)) / max(1, int(nq))
        if score > 0.1:
            out.append((i, score))
    out.sort(key=lambda t: (-t[1], t[0]))
    return out[:5]


class KeyframeDB:
    """Fixed-capacity, device-resident descriptor slots: desc [K_cap, rows, 32] u8, counts [K_cap], ids [K_cap]."""

    def __init__(self, k_cap, rows, device):
        import torch
        self.k_cap, self.rows = int(k_cap), int(rows)
        self.desc = torch.zeros((self.k_cap, self.rows, 32), dtype=torch.uint8, device=device)
        self.counts = torch.zeros((self.k_cap,), dtype=torch.int32, device=device)
        self.ids = torch.full((self.k_cap,), -1, dtype=torch.int64, device=device)
        self.n = 0

    def add(self, kf_id, desc, count):
        """desc: [>=count, 32] u8 tensor on the DB's device. Oldest keyframe is dropped beyond capacity (:28-30)."""
        import torch
        if self.n == self.k_cap:
            self.desc = torch.roll(self.desc, -1, 0)
            self.counts = torch.roll(self.counts, -1, 0)
            self.ids = torch.roll(self.ids, -1, 0)
            self.n -= 1
        c = min(int(count), self.rows)
        self.desc[self.n].zero_()
        self.desc[self.n, :c] = desc[:c]
        self.counts[self.n] = c
        self.ids[self.n] = int(kf_id)
        self.n += 1

    def all_gather(self, group=None, via_host=False):
        """One all-gather per tensor of the padded slots (uniform message sizes). Returns a KeyframeDB holding
        world*k_cap slots; empty slots keep count 0 and are skipped by the scorer. With backend "nccl" (RCCL) the
        device tensors are gathered in place over xGMI; via_host=True stages through host memory for a CPU-only
        backend (gloo rehearsal of a device-resident DB)."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(group)
        dev = self.desc.device
        g = KeyframeDB.__new__(KeyframeDB)
        g.k_cap, g.rows = self.k_cap * world, self.rows
        gdev = torch.device("cpu") if via_host else dev
        g.desc = torch.empty((g.k_cap, self.rows, 32), dtype=torch.uint8, device=gdev)
        g.counts = torch.empty((g.k_cap,), dtype=torch.int32, device=gdev)
        g.ids = torch.empty((g.k_cap,), dtype=torch.int64, device=gdev)
        for dst, src in ((g.desc, self.desc), (g.counts, self.counts), (g.ids, self.ids)):
            src = src.contiguous().view(-1)
            dist.all_gather_into_tensor(dst.view(-1), src.cpu() if via_host else src, group=group)
        if via_host:
            g.desc, g.counts, g.ids = g.desc.to(dev), g.counts.to(dev), g.ids.to(dev)
        g.n = g.k_cap
        return g

    def find_candidates(self, matcher, d_query, nq, query_id, min_frames_between, ratio=0.7):
        """Device scan + host scoring. d_query: [>=nq, 32] u8 on the same device."""
        import torch
        good = torch.zeros((self.k_cap,), dtype=torch.int32, device=self.desc.device)
        matcher.match_db_device(d_query, int(nq), self.desc, self.counts, self.k_cap, self.rows * 32, ratio, good)
        matcher.sync()
        return score_candidates(good.cpu().numpy(), nq, query_id, self.ids.cpu().numpy(), self.counts.cpu().numpy(),
                                min_frames_between)
