// runtime.hip -- device, memory, stream and event entry points of the C-ABI (include/bcnn_hip.h).
// Replaces the bcnn_cuda_* helper family (reference src/bcnn_utils.c:101-201). No process-global
// library handles: the only state is one current stream per host thread.
#include "common.h"

#include <cstring>

namespace bcnn_hip {
static thread_local hipStream_t g_stream = nullptr;  // nullptr = null stream (PyTorch-ROCm default)
hipStream_t current_stream() { return g_stream; }

__global__ void fill_f32_kernel(float* __restrict__ x, size_t n, float v) {
    // scalar head up to 16-byte alignment, 16-byte stores on the body, scalar tail
    size_t head = ((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / 4;
    if (head > n) head = n;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (i < head) x[i] = v;
    float* body = x + head;
    const size_t nb = n - head, n4 = nb / 4;
    float4* b4 = reinterpret_cast<float4*>(body);
    const float4 v4 = make_float4(v, v, v, v);
    for (size_t j = i; j < n4; j += stride) b4[j] = v4;
    for (size_t j = n4 * 4 + i; j < nb; j += stride) body[j] = v;
}
}  // namespace bcnn_hip

using namespace bcnn_hip;

extern "C" {

int bcnn_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void bcnn_hip_set_device(int id) { HIP_CHECK(hipSetDevice(id)); }

int bcnn_hip_get_device(void) {
    int id = 0;
    HIP_CHECK(hipGetDevice(&id));
    return id;
}

const char* bcnn_hip_device_name(void) {
    static thread_local char name[256];
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, bcnn_hip_get_device()));
    snprintf(name, sizeof(name), "%s", prop.gcnArchName);
    return name;
}

float* bcnn_hip_malloc_f32(size_t n) {
    float* p = nullptr;
    if (n == 0) return nullptr;
    HIP_CHECK(hipMalloc((void**)&p, n * sizeof(float)));
    HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(float), current_stream()));
    return p;
}

int* bcnn_hip_malloc_i32(size_t n) {
    int* p = nullptr;
    if (n == 0) return nullptr;
    HIP_CHECK(hipMalloc((void**)&p, n * sizeof(int)));
    HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(int), current_stream()));
    return p;
}

void bcnn_hip_free(void* p) {
    if (p) HIP_CHECK(hipFree(p));
}

void bcnn_hip_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, current_stream()));
    HIP_CHECK(hipStreamSynchronize(current_stream()));
}

void bcnn_hip_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, current_stream()));
    HIP_CHECK(hipStreamSynchronize(current_stream()));
}

void bcnn_hip_memcpy_d2d(void* dst, const void* src, size_t bytes) {
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, current_stream()));
}

void bcnn_hip_fill_f32(float* x, size_t n, float value) {
    if (n == 0) return;
    if (value == 0.0f) {
        HIP_CHECK(hipMemsetAsync(x, 0, n * sizeof(float), current_stream()));
        return;
    }
    fill_f32_kernel<<<stream_grid(n / 4 + 1, 256), 256, 0, current_stream()>>>(x, n, value);
    KERNEL_CHECK();
}

void bcnn_hip_sync(void) { HIP_CHECK(hipStreamSynchronize(current_stream())); }

void* bcnn_hip_stream_create(void) {
    hipStream_t s;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return (void*)s;
}

void bcnn_hip_stream_destroy(void* stream) {
    if (stream) HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
}

void bcnn_hip_set_stream(void* stream) { g_stream = (hipStream_t)stream; }
void* bcnn_hip_get_stream(void) { return (void*)g_stream; }

void* bcnn_hip_event_create(void) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    return (void*)e;
}

void bcnn_hip_event_destroy(void* ev) {
    if (ev) HIP_CHECK(hipEventDestroy((hipEvent_t)ev));
}

void bcnn_hip_event_record(void* ev) { HIP_CHECK(hipEventRecord((hipEvent_t)ev, current_stream())); }
void bcnn_hip_event_sync(void* ev) { HIP_CHECK(hipEventSynchronize((hipEvent_t)ev)); }

float bcnn_hip_event_elapsed_ms(void* start, void* stop) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
    return ms;
}

}  // extern "C"
