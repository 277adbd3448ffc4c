// Probe for the matrix-core 7x7 Gaussian of the band kernel (round 2): checks on the real chip, against an exact
// integer reference, that
//   pass 1  H = P x T_h      v_mfma_i32_32x32x32_i8 (pixels XOR 0x80, C-in = 128*257)             -> exact u16 sums
//   pass 2  out = H^T x T_v  v_mfma_f32_32x32x16_bf16 on the hi/lo BYTES of H zero-extended to 16 bits, read as bf16
//                            subnormals/normals (0x00XX == XX * 2^-133 for all XX in 0..255)       -> acc * 2^-24, exact
//   round   v_mul_f32 x256 + v_cvt_pk_u8_f32 (needs round-to-nearest-even + saturation)
// and prints what v_cvt_pk_u8_f32 does on ties, and whether unaligned ds_read_b64 works.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_blur_probe mfma_blur_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ int k7(int t) { return t == 0 || t == 6 ? 18 : (t == 1 || t == 5 ? 34 : (t == 2 || t == 4 ? 49 : (t == 3 ? 55 : 0))); }

__global__ void k_cvt_probe(const float* in, uint32_t* out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}

__global__ void k_lds_unaligned(uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint8_t s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = (uint8_t)i;
    __syncthreads();
    const int off = threadIdx.x;      // byte offsets 0..63
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
    out[2 * threadIdx.x] = (uint32_t)v;
    out[2 * threadIdx.x + 1] = (uint32_t)(v >> 32);
}

// One wave per (column block of 32 output columns, strip of 26 output rows). src: rows y0-3 .. y0+28 (32 rows) of a
// plane with pitch `pitch`, column 0 of the buffer <-> x = -16. dst: [rows][w] blurred (ties to even).
__global__ __launch_bounds__(64) void k_blur_probe(const uint8_t* __restrict__ src, int pitch, int nblk, uint8_t* __restrict__ dst,
                                                   int dpitch, int out_rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x, r = lane & 31, g = lane >> 5;
    const int cb = blockIdx.x, strip = blockIdx.y;
    const uint8_t* sp = src + (size_t)strip * 26 * pitch;       // strip's first input row (= output row - 3)
    // stage 32 rows x 64 columns [x0-16, x0+48) into LDS (pitch 64)
    for (int i = lane; i < 32 * 4; i += 64) {
        const int row = i >> 2, c = i & 3;
        *reinterpret_cast<uint4*>(smem + row * 64 + 16 * c) = *reinterpret_cast<const uint4*>(sp + (size_t)row * pitch + 32 * cb + 16 * c);
    }
    __syncthreads();
    // band matrices of pass 1 (B operand: lane (n = r, g) holds k = 16g .. 16g+15 as bytes)
    v4i B1, B2;
    {
        uint32_t b1[4], b2[4];
        for (int d = 0; d < 4; d++) {
            uint32_t w1 = 0, w2 = 0;
            for (int t = 0; t < 4; t++) {
                const int k = 16 * g + 4 * d + t;
                const int t1 = k - r - 13, t2 = k - r + 19;
                w1 |= (uint32_t)((t1 >= 0 && t1 <= 6) ? k7(t1) : 0) << (8 * t);
                w2 |= (uint32_t)((t2 >= 0 && t2 <= 6) ? k7(t2) : 0) << (8 * t);
            }
            b1[d] = w1; b2[d] = w2;
        }
        B1 = v4i{(int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
        B2 = v4i{(int)b2[0], (int)b2[1], (int)b2[2], (int)b2[3]};
    }
    // weights of pass 2 (B operand, bf16: lane (n = r = output row, g), k-step s, element j <-> input row i(s,g,j))
    v4i Wlo[2], Whi[2];
    for (int s = 0; s < 2; s++) {
        uint32_t lo[4], hi[4];
        for (int d = 0; d < 4; d++) {
            uint32_t l2 = 0, h2 = 0;
            for (int e = 0; e < 2; e++) {
                const int j = 2 * d + e;
                const int i = (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * g;
                const int tap = i - r;
                const float w = (tap >= 0 && tap <= 6) ? (float)k7(tap) : 0.f;
                // result scale 2^-24 (x256 in the epilogue): lo plane w * 2^(133-24), hi plane w * 2^(133-16)
                const uint32_t bl = __float_as_uint(ldexpf(w, 109)) >> 16, bh = __float_as_uint(ldexpf(w, 117)) >> 16;
                l2 |= bl << (16 * e); h2 |= bh << (16 * e);
            }
            lo[d] = l2; hi[d] = h2;
        }
        Wlo[s] = v4i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3]};
        Whi[s] = v4i{(int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
    v16i C0;
    for (int i = 0; i < 16; i++) C0[i] = 128 * 257;

    // ---- pass 1 ----
    v4i a0 = *reinterpret_cast<const v4i*>(smem + r * 64 + 16 * g);
    v4i a1 = *reinterpret_cast<const v4i*>(smem + r * 64 + 32 + 16 * g);
    for (int i = 0; i < 4; i++) { a0[i] ^= 0x80808080; a1[i] ^= 0x80808080; }
    v16i H = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, B1, C0, 0, 0, 0);
    H = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, B2, H, 0, 0, 0);
    // ---- repack: registers 8s .. 8s+7 -> bf16 pairs of the lo / hi byte planes ----
    v4i Alo[2], Ahi[2];
    for (int s = 0; s < 2; s++)
        for (int d = 0; d < 4; d++) {
            const uint32_t ha = (uint32_t)H[8 * s + 2 * d], hb = (uint32_t)H[8 * s + 2 * d + 1];
            Alo[s][d] = (int)__builtin_amdgcn_perm(hb, ha, 0x0c040c00u);
            Ahi[s][d] = (int)__builtin_amdgcn_perm(hb, ha, 0x0c050c01u);
        }
    // ---- pass 2 ----
    v16f O;
    for (int i = 0; i < 16; i++) O[i] = 0.f;
    for (int s = 0; s < 2; s++) {
        O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, Ahi[s]), __builtin_bit_cast(v8bf, Whi[s]), O, 0, 0, 0);
        O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, Alo[s]), __builtin_bit_cast(v8bf, Wlo[s]), O, 0, 0, 0);
    }
    // ---- epilogue: lane (n = r = output row, g): registers 4q .. 4q+3 = columns 8q + 4g + (0..3) ----
    if (r < 26) {
        const int orow = strip * 26 + r;
        if (orow < out_rows)
            for (int q = 0; q < 4; q++) {
                uint32_t o = 0;
                for (int e = 0; e < 4; e++) o = __builtin_amdgcn_cvt_pk_u8_f32(O[4 * q + e] * 256.0f, e, o);
                *reinterpret_cast<uint32_t*>(dst + (size_t)orow * dpitch + 32 * cb + 8 * q + 4 * g) = o;
            }
    }
}

int main() {
    // ---- v_cvt_pk_u8_f32 ----
    {
        const float hin[] = {0.5f, 1.5f, 2.5f, 3.5f, 254.5f, 255.5f, 256.7f, -3.0f, 0.49999997f, 1e9f, 127.5f, 128.5f, 0.75f, 1.25f};
        const int n = sizeof(hin) / sizeof(float);
        float* din; uint32_t* dout;
        CK(hipMalloc(&din, sizeof(hin))); CK(hipMalloc(&dout, 4 * n));
        CK(hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_cvt_probe, dim3(1), dim3(64), 0, 0, din, dout, n);
        uint32_t ho[32];
        CK(hipMemcpy(ho, dout, 4 * n, hipMemcpyDeviceToHost));
        printf("v_cvt_pk_u8_f32:");
        for (int i = 0; i < n; i++) printf(" %g->%u", hin[i], ho[i]);
        printf("\n");
        const bool rne = ho[0] == 0 && ho[1] == 2 && ho[2] == 2 && ho[3] == 4 && ho[4] == 254 && ho[5] == 255 && ho[6] == 255 && ho[7] == 0;
        printf("cvt_pk_u8_f32 rounds to nearest even and saturates: %s\n", rne ? "YES" : "NO");
    }
    // ---- unaligned ds_read_b64 ----
    {
        uint32_t* dout; CK(hipMalloc(&dout, 4 * 128));
        hipLaunchKernelGGL(k_lds_unaligned, dim3(1), dim3(64), 0, 0, dout);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) printf("unaligned ds_read_b64: FAULT (%s)\n", hipGetErrorString(e));
        else {
            uint32_t ho[128];
            CK(hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost));
            int bad = 0;
            for (int t = 0; t < 64; t++) {
                uint64_t v = ((uint64_t)ho[2 * t + 1] << 32) | ho[2 * t];
                for (int b = 0; b < 8; b++) if (((v >> (8 * b)) & 0xFF) != (uint64_t)((t + b) & 0xFF)) bad++;
            }
            printf("unaligned ds_read_b64: %s (%d wrong bytes)\n", bad ? "WRONG DATA" : "OK", bad);
        }
    }
    // ---- the blur ----
    const int nblk = 128, w = 32 * nblk, strips = 10, out_rows = 26 * strips, in_rows = out_rows + 6;
    const int pitch = w + 32;    // column 0 <-> x = -16
    std::vector<uint8_t> src((size_t)(in_rows + 32) * pitch);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t i = 0; i < src.size(); i++) {
        const size_t row = i / pitch, col = i % pitch;
        uint8_t v = (uint8_t)(rnd() >> 24);
        if (row >= 60 && row < 100 && col > 500 && col < 900) v = 255;             // saturating plateau
        if (row >= 120 && row < 160 && col > 1500 && col < 1900) v = 0;
        if (row >= 180 && row < 220) v = (uint8_t)(128 + ((rnd() >> 20) & 1));      // around the i8 sign flip
        if (col > 3000 && col < 3400) v = (uint8_t)(250 + (rnd() >> 60) % 6);       // bright noise
        src[i] = v;
    }
    uint8_t *dsrc, *ddst;
    CK(hipMalloc(&dsrc, src.size())); CK(hipMalloc(&ddst, (size_t)out_rows * w));
    CK(hipMemcpy(dsrc, src.data(), src.size(), hipMemcpyHostToDevice));
    CK(hipMemset(ddst, 0xEE, (size_t)out_rows * w));
    hipLaunchKernelGGL(k_blur_probe, dim3(nblk, strips), dim3(64), 32 * 64, 0, dsrc, pitch, nblk, ddst, w, out_rows);
    CK(hipDeviceSynchronize());
    std::vector<uint8_t> got((size_t)out_rows * w);
    CK(hipMemcpy(got.data(), ddst, got.size(), hipMemcpyDeviceToHost));
    const int K[7] = {18, 34, 49, 55, 49, 34, 18};
    long bad = 0, ties = 0, sat = 0;
    for (int y = 0; y < out_rows; y++)
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int dy = 0; dy < 7; dy++) {
                uint32_t hs = 0;
                for (int dx = 0; dx < 7; dx++) hs += K[dx] * src[(size_t)(y + dy) * pitch + (x + 16 - 3 + dx)];
                acc += K[dy] * hs;
            }
            if ((acc & 0xFFFF) == 0x8000) ties++;
            uint32_t rr = (acc + 0x7FFF + ((acc >> 16) & 1)) >> 16;
            if (rr > 255) { rr = 255; sat++; }
            if (got[(size_t)y * w + x] != rr) {
                if (bad < 10) printf("mismatch y %d x %d: got %u want %u (acc %u)\n", y, x, got[(size_t)y * w + x], rr, acc);
                bad++;
            }
        }
    printf("blur probe: %ld mismatches of %ld pixels (%ld exact ties, %ld saturated)\n", bad, (long)out_rows * w, ties, sat);
    return bad ? 1 : 0;
}
