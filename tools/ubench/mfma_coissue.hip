// mfma_coissue.hip — does a SIMD's matrix pipe run v_mfma_f32_4x4x1_16b_f32 BESIDE the vector instructions of its other waves?
// (round 4: the review's premise for moving the blur's FMAs to MFMA is "a different issue port".)
// One 1024-thread workgroup per CU (96 KB of LDS keeps a second one away): 16 waves, wave w on SIMD w % 4 (checked below from HW_ID).
// Every wave runs one ROLE for a fixed number of instructions and reports its own elapsed time (s_memrealtime, 100 MHz):
//   S  v_mfma_f32_16x16x4_f32, two accumulators (the 8-pass f32 form)
//   M  v_mfma_f32_4x4x1_16b_f32, two accumulators        B  v_mfma_f32_16x16x16_bf16, two accumulators (reference: a "real" matrix-pipe op)
//   P  v_pk_fma_f32 x 8 independent                       F  v_fma_f32 x 8 independent
//   I  v_add_u32 / v_xor_b32 x 8 independent              C  v_cvt_f32_ubyte0 + v_cvt_u32_f32 chains
//   D  v_fma_f64 x 8 independent                          L  ds_read_b32 x 8 (linear)
//   .  idle
// A pattern names the role of waves 0..15.  "MMMM...." = one M wave per SIMD alone; "MMMMPPPP" = one M + one P wave per SIMD, etc.
// If M and P overlap, each finishes in about the time it needs alone; if they share the vector ALU, each takes about the SUM.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue tools/ubench/mfma_coissue.hip && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
struct Roles { char r[16]; int iters[16]; };

__global__ __launch_bounds__(1024) void k_roles(Roles R, float* sink, unsigned long long* t_out, unsigned* hwid_out) {
    extern __shared__ float lds[];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const char role = R.r[w];
    const int iters = R.iters[w];
    lds[threadIdx.x] = (float)threadIdx.x;
    float a = (float)l * 1e-3f, b = 1.0f + (float)l * 1e-4f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1};
    f32x2 p[8]; float f[8]; uint32_t u[8]; double d[8];
    for (int i = 0; i < 8; ++i) { p[i] = f32x2{(float)i, (float)l}; f[i] = (float)(i + l); u[i] = (uint32_t)(i * 77 + l); d[i] = (double)(i + l); }
    const f32x2 mul = {1.0000001f, 0.9999999f}, add = {1e-6f, -1e-6f};
    bf16x4 ba = {(short)(0x3f80 + l), 0x3f80, 0x3f00, 0x3e80}, bb = {0x3f80, (short)(0x3f00 + l), 0x3f80, 0x3f80};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (role == 'M') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 3, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 5, 0); }
        }
    } else if (role == 'S') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0); }
        }
    } else if (role == 'B') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ba, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ba, bb, c1, 0, 0, 0); }
        }
    } else if (role == 'P') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], mul, add);
        }
    } else if (role == 'F') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(b), "v"(a));
        }
    } else if (role == 'I') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(l));
        }
    } else if (role == 'C') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[i]) : "v"(u[i]));
        }
    } else if (role == 'D') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        }
    } else if (role == 'L') {
        const float* lp = lds + l;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f[i]) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lp), "n"(i * 256));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + f[i] + (float)u[i] + (float)d[i];
    if (s == 12345.678f) sink[0] = s;
    if (l == 0) {
        t_out[blockIdx.x * 16 + w] = t1 - t0;
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        hwid_out[blockIdx.x * 16 + w] = hw;
    }
}

int main() {
    float* sink; unsigned long long* dt; unsigned* dh;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&dt, cus * 16 * 8)); CK(hipMalloc(&dh, cus * 16 * 4));
    std::vector<unsigned long long> ht(cus * 16); std::vector<unsigned> hh(cus * 16);
    CK(hipFuncSetAttribute((const void*)k_roles, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    // instructions per role tuned so that every role alone (one wave per SIMD) takes a similar time
    auto iters_of = [](char r) { switch (r) { case 'M': return 2000; case 'B': return 2000; case 'S': return 1000; case 'P': return 4000; case 'F': return 6000; case 'I': return 6000;
                                              case 'C': return 4000; case 'D': return 4000; case 'L': return 2000; default: return 0; } };
    bool printed_map = false;
    auto run = [&](const char* pat) -> int {
        Roles R; memset(&R, 0, sizeof R);
        for (int w = 0; w < 16; ++w) { R.r[w] = pat[w]; R.iters[w] = iters_of(pat[w]); }
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_roles, dim3(cus), dim3(1024), 96 * 1024, 0, R, sink, dt, dh);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(ht.data(), dt, cus * 16 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hh.data(), dh, cus * 16 * 4, hipMemcpyDeviceToHost));
        if (!printed_map) {
            printed_map = true;
            printf("wave -> SIMD of workgroup 0 (HW_ID bits 5:4):"); for (int w = 0; w < 16; ++w) printf(" %u", (hh[w] >> 4) & 3); printf("\n");
        }
        // mean time of each role over all CUs, in ns per wave-instruction of that role
        printf("%-18s", pat);
        std::string seen;
        for (int w = 0; w < 16; ++w) {
            const char r = pat[w];
            if (r == '.' || seen.find(r) != std::string::npos) continue;
            seen += r;
            double tot = 0; int n = 0;
            for (int c = 0; c < cus; ++c) for (int v = 0; v < 16; ++v) if (pat[v] == r) { tot += (double)ht[c * 16 + v]; ++n; }
            const double ns = tot / n * 10.0;       // 100 MHz ticks
            printf("  %c: %8.1f us = %6.2f ns / instr / wave", r, ns * 1e-3, ns / (iters_of(r) * 16.0));
        }
        printf("\n");
        return 0;
    };
    printf("one role alone, 1 / 2 / 4 waves per SIMD\n");
    for (const char* p : {"MMMM............", "MMMMMMMM........", "MMMMMMMMMMMMMMMM", "BBBB............", "BBBBBBBBBBBBBBBB", "SSSS............", "SSSSSSSSSSSSSSSS", "PPPP............", "PPPPPPPP........", "PPPPPPPPPPPPPPPP",
                          "FFFF............", "FFFFFFFFFFFFFFFF", "IIII............", "IIIIIIIIIIIIIIII", "CCCC............", "CCCCCCCCCCCCCCCC", "DDDD............", "DDDDDDDDDDDDDDDD",
                          "LLLL............", "LLLLLLLLLLLLLLLL"}) run(p);
    printf("mixes on the SAME SIMD (waves w and w + 4 share SIMD w %% 4)\n");
    for (const char* p : {"MMMMPPPP........", "MMMMFFFF........", "MMMMIIII........", "MMMMCCCC........", "MMMMDDDD........", "MMMMLLLL........",
                          "SSSSPPPP........", "SSSSFFFF........", "SSSSIIII........", "SSSSCCCC........", "SSSSDDDD........", "SSSSLLLL........", "SSSSSSSSPPPPPPPP", "SSSSSSSSIIIIIIII",
                          "BBBBPPPP........", "BBBBFFFF........", "BBBBIIII........", "BBBBCCCC........",
                          "MMMMMMMMPPPPPPPP", "MMMMMMMMIIIIIIII", "MMMMMMMMCCCCCCCC", "MMMMPPPPIIIICCCC", "BBBBBBBBPPPPPPPP",
                          "PPPPIIII........", "PPPPCCCC........"}) run(p);
    printf("mixes on DIFFERENT SIMDs (SIMD 0,2 one role, SIMD 1,3 the other)\n");
    for (const char* p : {"MPMPMPMP........", "BPBPBPBP........"}) run(p);
    return 0;
}
