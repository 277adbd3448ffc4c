"""Frame sharding of a recorded sequence across the GPUs of one node.

Extraction of frame i depends on nothing else and the temporal match (i, i-1) on one neighbour, so the sequence
shards embarrassingly: rank g of G owns the contiguous range [g*F/G, (g+1)*F/G) and additionally extracts the
last frame of range g-1 as a one-frame halo (recompute is cheaper than exchanging 64 KB of descriptors), so every
consecutive pair is matched exactly once and no collective is needed on the data path.
"""


def frame_range(n_frames, rank, world):
    """Contiguous range [lo, hi) owned by `rank`."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard request")
    return (n_frames * rank) // world, (n_frames * (rank + 1)) // world


def shard_plan(n_frames, rank, world):
    """What a rank extracts and which (query, train) frame pairs it matches.

    Returns dict(lo, hi, extract=[first, last) incl. halo, pairs=[(q, t), ...]) with query = current frame and
    train = previous frame (the SlamPipeline sketch's order; reference docs/milestones/H12_CLEAN_ARCHITECTURE.md:595-605).
    """
    lo, hi = frame_range(n_frames, rank, world)
    first = lo - 1 if (lo > 0 and hi > lo) else lo
    pairs = [(i, i - 1) for i in range(max(lo, 1), hi)]
    return {"lo": lo, "hi": hi, "extract": (first, hi), "pairs": pairs}
