// mall_copy.hip — the ceiling k_warp_lean should be held against: its float32 source (the pre-warp images of one launch group, 199 MB at 4K)
// was WRITTEN by the kernel just before it and is kept under the 256 MB Infinity Cache on purpose, so an HBM-sized device copy is the wrong yardstick.
// Here a writer kernel fills n frames of 3840 x 2160 x 3 float32 and a reader kernel, launched right behind it on the same stream, streams them
// back and stores a quarter of the bytes (the uint8 frame); only the reader is timed (events on either side of it).  Reader shapes:
//   x4      16 B per lane, consecutive lanes consecutive 16-B pieces (the widest coalesced stream)
//   x3      12 B per lane = one pixel per lane (buffer_load_b96), lanes consecutive pixels
//   tap4    four 12-B taps per pixel at (x, y) (x+1, y) (x, y+1) (x+1, y+1), 4 rows per thread, all 16 loads issued first: k_warp_lean's access
//           shape with the identity map (no coordinates, no weights), i.e. the gathers' cost without the barrel's footprint
//   row2    each thread loads its pixel of rows y and y + 1 only (2 x 12 B) and gets the right-hand neighbour by a lane shift (DPP): half the loads
//   hipcc --offload-arch=gfx950 -O3 -o mall_copy tools/ubench/mall_copy.hip && ./mall_copy
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
static int W = 3840, H = 2160;      // argv: W H frames...   (default: 4K, 1 - 4 frames)

__global__ __launch_bounds__(256) void k_fill(f32x4* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u, i += 256) if (i < n4) out[i] = f32x4{1.f, 2.f, 3.f, (float)threadIdx.x};
}
__device__ __forceinline__ uint32_t q8(float a, float b, float c, float d) {
    return (uint32_t)(a * 255.f) | ((uint32_t)(b * 255.f) << 8) | ((uint32_t)(c * 255.f) << 16) | ((uint32_t)(d * 255.f) << 24);
}
// 16 B per lane: 4 loads in flight per thread, one dword of output per load
__global__ __launch_bounds__(256) void k_read_x4(const f32x4* __restrict__ in, size_t n4, uint32_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = in[i + 256 * u < n4 ? i + 256 * u : 0];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + 256 * u < n4) out[i + 256 * u] = q8(v[u][0], v[u][1], v[u][2], v[u][3]);
}
__device__ __forceinline__ u32x3 ld3(__amdgpu_buffer_rsrc_t rs, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b96(rs, off, 0, 0); }
// one pixel per lane and row, 4 rows per thread (128 x 8 tiles as k_warp_lean)
template <int MODE>
__global__ __launch_bounds__(256) void k_read_px(const float* __restrict__ in, uint8_t* __restrict__ out, int W, int H) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 2 + (wv & 1)) * 64 + lane;
    const int yb = blockIdx.y * 8 + (wv >> 1);
    const float* pre = in + (size_t)blockIdx.z * W * H * 3;
    uint8_t* o = out + (size_t)blockIdx.z * W * H * 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pre), 0, (int)((uint32_t)W * H * 12u), 0x00020000);
    const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(o, 0, W * H * 3, 0x00020000);
    float acc[4][3];
    if (MODE == 0) {                // x3
        u32x3 a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = ld3(rs, ((uint32_t)(yb + 2 * r) * W + x) * 12u);
#pragma unroll
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) acc[r][c] = __uint_as_float(a[r][c]);
    } else if (MODE == 1) {         // tap4
        u32x3 a[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t off = ((uint32_t)(yb + 2 * r) * W + x) * 12u;
            a[r][0] = ld3(rs, off); a[r][1] = ld3(rs, off + 12u); a[r][2] = ld3(rs, off + W * 12u); a[r][3] = ld3(rs, off + W * 12u + 12u);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c)
            acc[r][c] = ((__uint_as_float(a[r][0][c]) * 0.4f + __uint_as_float(a[r][1][c]) * 0.1f) + __uint_as_float(a[r][2][c]) * 0.4f) + __uint_as_float(a[r][3][c]) * 0.1f;
    } else {                        // row2
        u32x3 a[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t off = ((uint32_t)(yb + 2 * r) * W + x) * 12u;
            a[r][0] = ld3(rs, off); a[r][1] = ld3(rs, off + W * 12u);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) {
            const float p = __uint_as_float(a[r][0][c]), q = __uint_as_float(a[r][1][c]);
            const float pr = __uint_as_float(__builtin_amdgcn_update_dpp(0u, a[r][0][c], 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
            const float qr = __uint_as_float(__builtin_amdgcn_update_dpp(0u, a[r][1][c], 0x130, 0xf, 0xf, false));
            acc[r][c] = ((p * 0.4f + pr * 0.1f) + q * 0.4f) + qr * 0.1f;
        }
    }
    // uint8 row out: three bytes per lane, packed across lanes as dwords is the real kernel's job; here 3 byte stores would distort, so pack 4 lanes' worth
    // with a DPP-free trick: each lane stores one dword made of its own three bytes + zero at a 3-byte stride through the buffer (unaligned dword store)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t v = (uint32_t)(acc[r][0] * 255.f) | ((uint32_t)(acc[r][1] * 255.f) << 8) | ((uint32_t)(acc[r][2] * 255.f) << 16);
        // 48 lanes store the row's 192 bytes as dwords (values approximate: timing only)
        const uint32_t v1 = __builtin_amdgcn_ds_bpermute(((lane * 4 / 3) & 63) << 2, v);
        if (lane < 48) __builtin_amdgcn_raw_buffer_store_b32(v1, os, ((uint32_t)(yb + 2 * r) * W + (x - lane)) * 3u + lane * 4u, 0, 0);
    }
}

#include <vector>
#include <cstdlib>
int main(int argc, char** argv) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<int> counts = {1, 2, 3, 4};
    if (argc >= 3) { W = atoi(argv[1]); H = atoi(argv[2]); }
    if (argc >= 4) { counts.clear(); for (int i = 3; i < argc; ++i) counts.push_back(atoi(argv[i])); }
    printf("%d x %d float32 frames\n", W, H);
    for (int frames : counts) {
        const size_t nfl = (size_t)W * H * 3 * frames, n4 = nfl / 4;
        float* buf; uint8_t* out;
        CK(hipMalloc(&buf, nfl * 4 + 65536)); CK(hipMalloc(&out, nfl + 65536));
        const unsigned fb = (unsigned)((n4 + 2047) / 2048);
        auto timed = [&](auto reader, const char* name) -> int {
            float tot = 0; const int reps = 20;
            for (int i = 0; i < reps + 3; ++i) {
                hipLaunchKernelGGL(k_fill, dim3(fb), dim3(256), 0, 0, (f32x4*)buf, n4);
                CK(hipEventRecord(e0));
                reader();
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (i >= 3) tot += ms;
            }
            const double us = tot * 1e3 / reps, bytes = nfl * 4.0 + nfl;
            printf("%d frame(s) (%.0f MB float32 just written) %-8s %7.1f us  %6.0f GB/s (source + uint8 out)\n", frames, nfl * 4.0 / 1e6, name, us, bytes / us / 1e3);
            return 0;
        };
        timed([&] { hipLaunchKernelGGL(k_read_x4, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, (const f32x4*)buf, n4, (uint32_t*)out); }, "x4");
        timed([&] { hipLaunchKernelGGL(k_read_px<0>, dim3(W / 128, H / 8, frames), dim3(256), 0, 0, buf, out, W, H); }, "x3");
        timed([&] { hipLaunchKernelGGL(k_read_px<1>, dim3(W / 128, H / 8, frames), dim3(256), 0, 0, buf, out, W, H); }, "tap4");
        timed([&] { hipLaunchKernelGGL(k_read_px<2>, dim3(W / 128, H / 8, frames), dim3(256), 0, 0, buf, out, W, H); }, "row2");
        CK(hipFree(buf)); CK(hipFree(out));
    }
    return 0;
}
