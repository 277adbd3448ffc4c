// Frame sharding of a recorded sequence (BASELINE.json north_star: "frames from a recorded sequence shard
// embarrassingly across the 8 GPUs of one node"). Extraction of frame i depends on nothing else and the temporal match
// (i, i-1) on one neighbour, so shard g of G owns the contiguous range [g F / G, (g+1) F / G) and additionally extracts the
// last frame of range g-1 as a one-frame halo: every consecutive pair is matched exactly once, by the shard that owns its
// later frame, and no exchange is needed. Same rule as aria_slam_amd/shard.py (which bench.py uses).
#pragma once
#include <cstddef>

namespace aria::pipeline {

struct ShardPlan {
    std::size_t lo = 0, hi = 0;    // owned frames [lo, hi)
    std::size_t first = 0;         // first frame the shard extracts: lo - 1 (the halo) when lo > 0 and the range is not empty
};

inline ShardPlan shardPlan(std::size_t n_frames, int shard, int n_shards) {
    ShardPlan p;
    if (n_shards < 1 || shard < 0 || shard >= n_shards) return p;
    p.lo = n_frames * (std::size_t)shard / (std::size_t)n_shards;
    p.hi = n_frames * (std::size_t)(shard + 1) / (std::size_t)n_shards;
    p.first = (p.lo > 0 && p.hi > p.lo) ? p.lo - 1 : p.lo;
    return p;
}

}  // namespace aria::pipeline
