// cvt_pk_u8_test.hip — is v_cvt_pk_u8_f32 the same function as cv2.convertScaleAbs' saturate_cast<uchar>(cvRound(|x|))?
// Exhaustive over every float32 in [0, 1.25] (the image scale; x * 255 up to 318) and a sweep of larger / special values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t quant_ref(float v) {
    const float s = fabsf(v * 255.0f);
    const int r = (int)rintf(s);
    return (uint32_t)min(max(r, 0), 255);
}
__device__ __forceinline__ uint32_t quant_pk(float v) {
    uint32_t d = 0;
    const float s = fabsf(v * 255.0f);
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(d) : "v"(s));
    return d & 255u;
}
__global__ void k(uint32_t lo, uint32_t n, unsigned long long* bad, uint32_t* first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = __uint_as_float(lo + i);
    if (quant_ref(v) != quant_pk(v)) { if (atomicAdd(bad, 1ull) == 0) *first = lo + i; }
}
int main() {
    unsigned long long* bad; uint32_t* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
    const uint32_t hi = 0x3FA00000u;            // 1.25f
    for (uint32_t lo = 0; lo < hi; lo += 1u << 28) {
        const uint32_t n = (hi - lo) < (1u << 28) ? hi - lo : (1u << 28);
        hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, lo, n, bad, first);
    }
    hipLaunchKernelGGL(k, dim3((1u << 24) / 256), dim3(256), 0, 0, 0x7F000000u, 1u << 24, bad, first);   // huge, inf, nan
    hipLaunchKernelGGL(k, dim3((1u << 24) / 256), dim3(256), 0, 0, 0xBF000000u, 1u << 24, bad, first);   // negatives around -0.5 .. -2
    unsigned long long b; uint32_t f;
    hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
    printf("mismatches %llu (first bits 0x%08x)\n", b, f);
    return 0;
}
