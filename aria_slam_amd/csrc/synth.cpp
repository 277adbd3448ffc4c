// Synthetic grayscale frame-pair generator (workload tooling for bench.py and the tests).
// Integer arithmetic only, so every consumer (C++, tests, CPU baseline, GPU bench) sees identical bytes:
//   scene  = mid-grey 128 canvas of (w+8) x (h+8), painted with R axis-aligned filled rectangles
//            (R = 400 per 640x480 of area; size 8..64 px, grey 16..239), parameters drawn from one
//            splitmix64 stream seeded with `seed`;
//   frame A = scene cropped at (4,4)     + per-pixel noise in [-8, 8]  (stateless hash of seed, pixel)
//   frame B = scene cropped at (1,2)     + per-pixel noise in [-4, 4]  -> B(x,y) shows A's content moved by (+3,+2)
// both clamped to [0,255].
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "aria_orb_hip.h"

namespace {

inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void synth_pair(uint64_t seed, int w, int h, uint8_t* a, uint8_t* b) {
    const int M = 4, cw = w + 2 * M, ch = h + 2 * M;
    std::vector<uint8_t> scene((size_t)cw * ch, 128);
    uint64_t s = seed * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull;
    int64_t nrect = (400ll * w * h + (640ll * 480) / 2) / (640ll * 480);
    if (nrect < 16) nrect = 16;
    for (int64_t r = 0; r < nrect; r++) {
        int rw = 8 + (int)(splitmix64(s) % 57);
        int rh = 8 + (int)(splitmix64(s) % 57);
        int x0 = (int)(splitmix64(s) % (uint64_t)(cw + 32)) - 32;
        int y0 = (int)(splitmix64(s) % (uint64_t)(ch + 32)) - 32;
        uint8_t g = (uint8_t)(16 + splitmix64(s) % 224);
        int xa = x0 < 0 ? 0 : x0, ya = y0 < 0 ? 0 : y0;
        int xb = x0 + rw > cw ? cw : x0 + rw, yb = y0 + rh > ch ? ch : y0 + rh;
        for (int y = ya; y < yb; y++)
            if (xb > xa) std::memset(&scene[(size_t)y * cw + xa], g, (size_t)(xb - xa));
    }
    const uint64_t ka = mix64(seed ^ 0xA5A5A5A5A5A5A5A5ull), kb = mix64(seed ^ 0x5A5A5A5A5A5A5A5Aull);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint64_t idx = (uint64_t)y * (uint64_t)w + (uint64_t)x;
            if (a) {
                int n = (int)(mix64(ka + idx * 0x9E3779B97F4A7C15ull) % 17) - 8;
                int v = scene[(size_t)(y + M) * cw + (x + M)] + n;
                a[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
            if (b) {
                int n = (int)(mix64(kb + idx * 0x9E3779B97F4A7C15ull) % 9) - 4;
                int v = scene[(size_t)(y + M - 2) * cw + (x + M - 3)] + n;
                b[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        }
}

}  // namespace

extern "C" {

int aria_synth_frame_pair(uint64_t seed, int width, int height, uint8_t* frame_a, uint8_t* frame_b) {
    if (width < 16 || height < 16) return ARIA_E_INVALID;
    synth_pair(seed, width, height, frame_a, frame_b);
    return ARIA_OK;
}

// out holds 2*n_pairs frames: A(seed0), B(seed0), A(seed0+1), B(seed0+1), ...
int aria_synth_sequence(uint64_t seed0, int n_pairs, int width, int height, uint8_t* out, int n_threads) {
    if (width < 16 || height < 16 || n_pairs < 0 || !out) return ARIA_E_INVALID;
    if (n_threads < 1) n_threads = 1;
    const size_t fb = (size_t)width * height;
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([=]() {
            for (int p = t; p < n_pairs; p += n_threads)
                synth_pair(seed0 + (uint64_t)p, width, height, out + (size_t)(2 * p) * fb, out + (size_t)(2 * p + 1) * fb);
        });
    for (auto& x : th) x.join();
    return ARIA_OK;
}

}  // extern "C"
