// write_bw.hip — how fast can the chip take a float32 image being written (and read back)?  The pre-warp intermediate of
// the CRT chain is 99.5 MB per 4K frame; k_phosphor's cost with and without its stores differs by ~48 us per 2 frames.
//   hipcc --offload-arch=gfx950 -O3 -o write_bw tools/ubench/write_bw.hip && ./write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

// every thread writes `per` dwords, a wave writes 256 contiguous bytes per store instruction
template <int NT>
__global__ __launch_bounds__(256) void k_write(float* __restrict__ out, size_t n, int per) {
    size_t base = ((size_t)blockIdx.x * 256 * per) + threadIdx.x;
    float v = (float)threadIdx.x;
    for (int i = 0; i < per; ++i) {
        size_t idx = base + (size_t)i * 256;
        if (idx < n) { if (NT) __builtin_nontemporal_store(v, out + idx); else out[idx] = v; }
    }
}
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(256) void k_write_x4(f32x4v* __restrict__ out, size_t n4, int per) {
    size_t base = ((size_t)blockIdx.x * 256 * per) + threadIdx.x;
    f32x4v v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int i = 0; i < per; ++i) {
        size_t idx = base + (size_t)i * 256;
        if (idx < n4) { if (NT) __builtin_nontemporal_store(v, out + idx); else out[idx] = v; }
    }
}
__global__ __launch_bounds__(256) void k_read_x3(const float* __restrict__ in, size_t npx, float* sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0;
    for (; i < npx; i += (size_t)gridDim.x * 256) { acc += in[i * 3] + in[i * 3 + 1] + in[i * 3 + 2]; }
    if (acc == -1.f) sink[0] = acc;
}

int main() {
    const size_t frame = (size_t)3840 * 2160 * 3;          // floats per 4K pre-warp frame
    for (int frames : {1, 2, 4}) {
        const size_t n = frame * frames;
        float* buf; float* sink;
        CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&sink, 64));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto time = [&](auto launch, const char* name, double bytes) {
            for (int i = 0; i < 3; ++i) launch();
            (void)hipEventRecord(e0);
            const int reps = 20;
            for (int i = 0; i < reps; ++i) launch();
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%d frame(s) %-28s %8.1f us  %7.0f GB/s\n", frames, name, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
        };
        const int per = 8;
        const unsigned blocks = (unsigned)((n + 256 * per - 1) / (256 * per));
        time([&] { hipLaunchKernelGGL(k_write<0>, dim3(blocks), dim3(256), 0, 0, buf, n, per); }, "store dword", n * 4.0);
        time([&] { hipLaunchKernelGGL(k_write<1>, dim3(blocks), dim3(256), 0, 0, buf, n, per); }, "store dword nt", n * 4.0);
        const unsigned blocks4 = (unsigned)((n / 4 + 256 * per - 1) / (256 * per));
        time([&] { hipLaunchKernelGGL(k_write_x4<0>, dim3(blocks4), dim3(256), 0, 0, (f32x4v*)buf, n / 4, per); }, "store dwordx4", n * 4.0);
        time([&] { hipLaunchKernelGGL(k_write_x4<1>, dim3(blocks4), dim3(256), 0, 0, (f32x4v*)buf, n / 4, per); }, "store dwordx4 nt", n * 4.0);
        time([&] { hipLaunchKernelGGL(k_read_x3, dim3(4096), dim3(256), 0, 0, buf, n / 3, sink); }, "read 3 x dword / px", n * 4.0);
        time([&] { (void)hipMemsetAsync(buf, 0, n * 4, 0); }, "hipMemsetAsync", n * 4.0);
        // write then read back, alternating (the phosphor -> warp pattern): is the image served from the Infinity Cache?
        time([&] { hipLaunchKernelGGL(k_write_x4<0>, dim3(blocks4), dim3(256), 0, 0, (f32x4v*)buf, n / 4, per);
                   hipLaunchKernelGGL(k_read_x3, dim3(4096), dim3(256), 0, 0, buf, n / 3, sink); }, "write x4 + read back", n * 8.0);
        CK(hipFree(buf)); CK(hipFree(sink));
    }
    return 0;
}
