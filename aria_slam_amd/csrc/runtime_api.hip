// Device memory, staging copies and events of the C-ABI (include/aria_orb_hip.h, ABI 4): what a host in the reference's
// language needs to feed the BATCH entry points without linking the HIP runtime itself -- the role cv::cuda::GpuMat::upload /
// download and the CUDA runtime play in the reference (src/legacy/Frame.cpp:19, src/adapters/gpu/OrbCudaExtractor.cpp:83,
// 102-103, 158, 186-187; cudaStreamSynchronize src/euroc_eval.cpp:153-154). No kernels here.
#include <hip/hip_runtime.h>

#include "aria_orb_hip.h"
#include "common.h"

using namespace aria;

namespace {
int set_device(int device) {
    int ndev = 0;
    ARIA_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return ARIA_E_NO_DEVICE;
    ARIA_HIP(hipSetDevice(device));
    return ARIA_OK;
}
}  // namespace

extern "C" {

int aria_device_count(int* n) {
    if (!n) return ARIA_E_INVALID;
    *n = 0;
    int ndev = 0;
    const hipError_t e = hipGetDeviceCount(&ndev);
    if (e == hipErrorNoDevice) return ARIA_OK;         // a machine without a GPU has zero devices: not a failure of the query
    ARIA_HIP(e);
    *n = ndev;
    return ARIA_OK;
}

int aria_device_alloc(int device, size_t bytes, void** d_ptr) {
    if (!d_ptr) return ARIA_E_INVALID;
    *d_ptr = nullptr;
    const int rc = set_device(device);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return ARIA_OK;
}

int aria_device_free(int device, void* d_ptr) {
    if (!d_ptr) return ARIA_OK;
    const int rc = set_device(device);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipFree(d_ptr));
    return ARIA_OK;
}

int aria_host_alloc_pinned(size_t bytes, void** h_ptr) {
    if (!h_ptr) return ARIA_E_INVALID;
    *h_ptr = nullptr;
    int ndev = 0;
    ARIA_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return ARIA_E_NO_DEVICE;
    ARIA_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return ARIA_OK;
}

int aria_host_free_pinned(void* h_ptr) {
    if (!h_ptr) return ARIA_OK;
    ARIA_HIP(hipHostFree(h_ptr));
    return ARIA_OK;
}

static int copy_async(int device, void* stream, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (bytes == 0) return ARIA_OK;
    if (!dst || !src || !stream) return ARIA_E_INVALID;      // stream 0 would be the legacy default stream: never implicit here
    const int rc = set_device(device);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipMemcpyAsync(dst, src, bytes, kind, (hipStream_t)stream));
    return ARIA_OK;
}
int aria_copy_h2d_async(int device, void* stream, void* d_dst, const void* h_src, size_t bytes) {
    return copy_async(device, stream, d_dst, h_src, bytes, hipMemcpyHostToDevice);
}
int aria_copy_d2h_async(int device, void* stream, void* h_dst, const void* d_src, size_t bytes) {
    return copy_async(device, stream, h_dst, d_src, bytes, hipMemcpyDeviceToHost);
}
int aria_copy_d2d_async(int device, void* stream, void* d_dst, const void* d_src, size_t bytes) {
    return copy_async(device, stream, d_dst, d_src, bytes, hipMemcpyDeviceToDevice);
}
int aria_fill_async(int device, void* stream, void* d_dst, int byte_value, size_t bytes) {
    if (bytes == 0) return ARIA_OK;
    if (!d_dst || !stream) return ARIA_E_INVALID;
    const int rc = set_device(device);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipMemsetAsync(d_dst, byte_value, bytes, (hipStream_t)stream));
    return ARIA_OK;
}

int aria_stream_synchronize(int device, void* stream) {
    if (!stream) return ARIA_E_INVALID;
    const int rc = set_device(device);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipStreamSynchronize((hipStream_t)stream));
    return ARIA_OK;
}

int aria_event_create(int device, void** event) {
    if (!event) return ARIA_E_INVALID;
    *event = nullptr;
    const int rc = set_device(device);
    if (rc != ARIA_OK) return rc;
    hipEvent_t e = nullptr;
    ARIA_HIP(hipEventCreate(&e));
    *event = (void*)e;
    return ARIA_OK;
}
int aria_event_destroy(int device, void* event) {
    if (!event) return ARIA_OK;
    const int rc = set_device(device);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipEventDestroy((hipEvent_t)event));
    return ARIA_OK;
}
int aria_event_record(int device, void* event, void* stream) {
    if (!event || !stream) return ARIA_E_INVALID;
    const int rc = set_device(device);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return ARIA_OK;
}
int aria_stream_wait_event(int device, void* stream, void* event) {
    if (!event || !stream) return ARIA_E_INVALID;
    const int rc = set_device(device);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return ARIA_OK;
}
int aria_event_synchronize(int device, void* event) {
    if (!event) return ARIA_E_INVALID;
    const int rc = set_device(device);
    if (rc != ARIA_OK) return rc;
    ARIA_HIP(hipEventSynchronize((hipEvent_t)event));
    return ARIA_OK;
}
int aria_event_elapsed_ms(void* start_event, void* stop_event, float* ms) {
    if (!start_event || !stop_event || !ms) return ARIA_E_INVALID;
    ARIA_HIP(hipEventElapsedTime(ms, (hipEvent_t)start_event, (hipEvent_t)stop_event));
    return ARIA_OK;
}

}  // extern "C"
