// Shared host-side helpers for the C-ABI implementation (not part of the public interface).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "aria_orb_hip.h"

namespace aria {

// The product library (libaria_orb_hip.so) reads NO environment variable: every variant kernel and every tuning or
// experiment switch exists only in the variants build (-DARIA_VARIANTS: libaria_orb_hip_variants.so, built by
// tests/test_gpu_variants.py and the profiling tools under tools/), where this is getenv.
#ifdef ARIA_VARIANTS
inline const char* aria_getenv(const char* name) { return std::getenv(name); }
#else
inline const char* aria_getenv(const char*) { return nullptr; }
#endif

// thread-local text of the last HIP failure, surfaced through aria_last_hip_error()
char* last_hip_error_buf();

inline int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    std::snprintf(last_hip_error_buf(), 256, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    switch (e) {
        case hipErrorOutOfMemory: return ARIA_E_OOM;
        case hipErrorNoDevice: case hipErrorInvalidDevice: case hipErrorInsufficientDriver: case hipErrorNotInitialized:
            return ARIA_E_NO_DEVICE;
        case hipErrorLaunchFailure: case hipErrorIllegalAddress: case hipErrorLaunchTimeOut: case hipErrorAssert:
        case hipErrorLaunchOutOfResources: case hipErrorSharedObjectInitFailed:
            return ARIA_E_KERNEL;
        default: return ARIA_E_HIP;
    }
}

#define ARIA_HIP(call)                                                          \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) return aria::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

// Host-blocking fill / copy ORDERED ON A HANDLE'S STREAM instead of the legacy default stream (plain hipMemset / hipMemcpy).
// A legacy-stream operation implicitly depends on every blocking stream of the device and fails with "operation would make
// the legacy stream depend on a capturing blocking stream" while ANOTHER host thread is capturing a hipGraph on one -- the
// single-frame entry points capture their schedule, and euroc_frontend --shards runs several handles from several threads
// (seen once in round 4: a matcher being created while a neighbour shard captured). The handles' own streams are also
// created non-blocking (aria::create_stream) so that nobody else's legacy-stream call can depend on them.
inline hipError_t memset_on(hipStream_t st, void* p, int v, size_t n) {
    const hipError_t e = hipMemsetAsync(p, v, n, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}
inline hipError_t memcpy_on(hipStream_t st, void* dst, const void* src, size_t n, hipMemcpyKind kind) {
    const hipError_t e = hipMemcpyAsync(dst, src, n, kind, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}
inline hipError_t create_stream(hipStream_t* st) { return hipStreamCreateWithFlags(st, hipStreamNonBlocking); }

// deferred error bits written by kernels
enum : int {
    ERRBIT_CAND_OVERFLOW = 1,   // FAST candidate list of some (frame, level) exceeded cand_cap
    ERRBIT_SORT_OVERFLOW = 2,   // tie-storm fallback: key arena exhausted (more tied levels in one pass than provisioned)
    ERRBIT_SEL_OVERFLOW = 4,    // tie-storm fallback: keypoint arena exhausted
    ERRBIT_KPCAP = 8,           // caller's kp_cap smaller than a frame's result
    ERRBIT_MATCHCAP = 16        // caller's match_cap smaller than a pair's result
};

inline int errbits_to_status(int bits) {
    if (bits & (ERRBIT_CAND_OVERFLOW | ERRBIT_SORT_OVERFLOW | ERRBIT_SEL_OVERFLOW)) return ARIA_E_OVERFLOW;
    if (bits & (ERRBIT_KPCAP | ERRBIT_MATCHCAP)) return ARIA_E_OUTPUT_TOO_SMALL;
    return ARIA_OK;
}

}  // namespace aria
