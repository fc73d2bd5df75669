// fetch_calib.hip — what does rocprofv3's FETCH_SIZE report for THIS code's load patterns on a known byte count?
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies 128-B requests at 64 B on gfx950 for wide streaming reads; other
// widths are uncalibrated.)  Each kernel reads a 2 GiB buffer exactly once — 8x the 256 MiB Infinity Cache, so nothing is
// served on-die — with one of the chain's access shapes:
//   k_px12   12 bytes per lane (float3 pixels), consecutive lanes consecutive pixels      = k_warp's taps, k_commit
//   k_byte3  three single-byte loads per lane at a 3-byte lane stride                       = k_phosphor's frame bytes
//   k_dword  4 bytes per lane                                                                = the reference shape
// Run:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib   and compare with the bytes printed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
__global__ __launch_bounds__(256) void k_px12(const F3* __restrict__ p, size_t n, float* sink) {
    float a = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { F3 v = p[i]; a += v.x + v.y + v.z; }
    if (a == -1.f) sink[0] = a;
}
__global__ __launch_bounds__(256) void k_byte3(const uint8_t* __restrict__ p, size_t npx, float* sink) {
    uint32_t a = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256) a += p[3 * i] + p[3 * i + 1] + p[3 * i + 2];
    if (a == 0xFFFFFFFFu) sink[0] = (float)a;
}
__global__ __launch_bounds__(256) void k_dword(const float* __restrict__ p, size_t n, float* sink) {
    float a = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += p[i];
    if (a == -1.f) sink[0] = a;
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    void* buf; float* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    const size_t npx12 = bytes / 12, npx3 = bytes / 3, n4 = bytes / 4;
    hipLaunchKernelGGL(k_px12, dim3(8192), dim3(256), 0, 0, (const F3*)buf, npx12, sink);
    hipLaunchKernelGGL(k_byte3, dim3(8192), dim3(256), 0, 0, (const uint8_t*)buf, npx3, sink);
    hipLaunchKernelGGL(k_dword, dim3(8192), dim3(256), 0, 0, (const float*)buf, n4, sink);
    (void)hipDeviceSynchronize();
    printf("bytes read once by each kernel: k_px12 %zu  k_byte3 %zu  k_dword %zu\n", npx12 * 12, npx3 * 3, n4 * 4);
    return 0;
}
