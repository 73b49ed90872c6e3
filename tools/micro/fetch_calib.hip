// Calibration of rocprofv3's FETCH_SIZE on gfx950 against kernels that read a known number of bytes from HBM exactly once:
// the guide's "x 2" correction is stated for wide coalesced reads; the Winograd kernels gather 8- and 4-byte pieces per lane
// and the LDS-DMA GEMMs dword rows, so the factor is measured here per access width (a 1 GiB buffer, larger than the 256 MB
// Infinity Cache, read once per kernel). build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib     (FETCH_SIZE is in KiB)
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T>
__global__ __launch_bounds__(256) void read_w(const T* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = p[i];
        s += reinterpret_cast<const float*>(&v)[0];
    }
    if (s == 123.456f) out[0] = s;
}
// 8 bytes per lane with a 16-byte lane stride (every other pair: the pairs of a Winograd tile row), two passes cover the buffer
__global__ __launch_bounds__(256) void read_pairs_strided(const float2* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (int half = 0; half < 2; ++half)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; 2 * i + half < n; i += (size_t)gridDim.x * 256) s += p[2 * i + half].x;
    if (s == 123.456f) out[0] = s;
}
int main() {
    const size_t bytes = 1ull << 30;
    void* p; float* out;
    hipMalloc(&p, bytes); hipMalloc(&out, 4); hipMemset(p, 0, bytes);
    hipDeviceSynchronize();
    read_w<float4><<<8192, 256>>>((const float4*)p, bytes / 16, out);
    read_w<float2><<<8192, 256>>>((const float2*)p, bytes / 8, out);
    read_w<float><<<8192, 256>>>((const float*)p, bytes / 4, out);
    read_pairs_strided<<<8192, 256>>>((const float2*)p, bytes / 8, out);
    hipDeviceSynchronize();
    printf("each kernel read %zu bytes once\n", bytes);
    return 0;
}
