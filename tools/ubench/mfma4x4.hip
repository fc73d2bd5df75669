// mfma4x4.hip — what v_mfma_f32_4x4x1_16b_f32 computes, lane by lane, and what it costs next to the VALU.
// The blur passes of k_phosphor_* are 1-D filters over INDEPENDENT signals held one per lane (V pass: lane = column, the row window in
// registers; H pass: lane = an 8-output segment of a staged row).  The 16-block 4x4x1 MFMA is an outer product per group of four lanes:
//   D_b[i][j] += A_b[i] * B_b[j]        block b = lane / 4;  A_b[i] read from lane 4b + i,  B_b[j] from lane 4b + j;  D_b[i][j] in register i of lane 4b + j
// so with B = the lane's own signal sample k and A_b[i] = tap[k - i], register i of a lane accumulates output i of ITS OWN signal:
// no cross-lane data layout at all.  cbsz:4 abid:m broadcasts block m's four A values to all sixteen blocks — one VGPR holds the tap
// vectors of sixteen filter steps.  This program checks, on the device:
//   1. the lane / register map of A, B and D with cbsz 0, and the broadcast with cbsz 4 and every abid;
//   2. that a chain of such MFMAs is bit for bit the k-ordered fmaf chain (zero taps included), on random data;
//   3. cycles per instruction with one and four waves per SIMD, dependent and independent accumulators, and whether a SIMD's other waves
//      keep issuing v_pk_fma_f32 at full rate meanwhile.
//   hipcc --offload-arch=gfx950 -O3 -o mfma4x4 tools/ubench/mfma4x4.hip && ./mfma4x4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CBSZ, int ABID>
__device__ f32x4 mm(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CBSZ, ABID, 0); }

__global__ void k_map(const float* a, const float* b, float* d) {      // d[17][4][64]: cbsz 0, then cbsz 4 with abid 0..15
    const int l = threadIdx.x;
    const f32x4 z = {0, 0, 0, 0};
    f32x4 r[17];
    r[0] = mm<0, 0>(a[l], b[l], z);
#define AB(m) r[1 + m] = mm<4, m>(a[l], b[l], z);
    AB(0) AB(1) AB(2) AB(3) AB(4) AB(5) AB(6) AB(7) AB(8) AB(9) AB(10) AB(11) AB(12) AB(13) AB(14) AB(15)
#undef AB
    for (int t = 0; t < 17; ++t) for (int i = 0; i < 4; ++i) d[(t * 4 + i) * 64 + l] = r[t][i];
}

// a 19-tap filter over a 22-sample window per lane, 4 outputs per lane: the MFMA chain against the fmaf chain
constexpr int R = 9, NT = 2 * R + 1, NW = NT + 3;
__global__ void k_chain(const float* taps, const float* win, float* out_mfma, float* out_fma) {
    const int l = threadIdx.x, gl = blockIdx.x * 64 + l;
    // tap vectors: step m (0 .. NW-1) lives in lanes 4 (m & 15) .. + 3 of register m >> 4: value tap[m - i] or 0
    float av[2];
    for (int q = 0; q < 2; ++q) { const int m = (l >> 2) + 16 * q, t = m - (l & 3); av[q] = (t >= 0 && t < NT) ? taps[t] : 0.0f; }
    float w[NW];
    for (int k = 0; k < NW; ++k) w[k] = win[(size_t)gl * NW + k];
    f32x4 acc = {0, 0, 0, 0};
#define ST(m) acc = mm<4, (m) & 15>(av[(m) >> 4], w[m], acc);
    ST(0) ST(1) ST(2) ST(3) ST(4) ST(5) ST(6) ST(7) ST(8) ST(9) ST(10) ST(11) ST(12) ST(13) ST(14) ST(15) ST(16) ST(17) ST(18) ST(19) ST(20) ST(21)
#undef ST
    for (int i = 0; i < 4; ++i) {
        out_mfma[(size_t)gl * 4 + i] = acc[i];
        float s = 0.0f;
        for (int t = 0; t < NT; ++t) s = fmaf(w[i + t], taps[t], s);
        out_fma[(size_t)gl * 4 + i] = s;
    }
}

// timing: mode 0 = MFMA chain on ONE accumulator, 1 = two accumulators alternating, 2 = v_pk_fma_f32 only (8 independent), 3 = waves 0,1 of a
// SIMD run MFMA (two accumulators) and waves 2,3 run pk_fma.  blocks of 256 threads = one wave per SIMD; gridDim.x = CUs * waves per SIMD
__global__ __launch_bounds__(256) void k_time(int mode, int iters, float* sink, unsigned long long* cyc) {
    const int l = threadIdx.x & 63;
    float a = (float)l * 1e-3f, b = 1.0f + (float)l * 1e-4f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1};
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f32x2{(float)i, (float)l};
    const f32x2 mul = {1.0000001f, 0.9999999f}, add = {1e-6f, -1e-6f};
    int m = mode;
    if (mode == 3) m = ((blockIdx.x >> 8) & 2) ? 2 : 1;       // blocks 0..511 -> MFMA, 512..1023 -> pk_fma (dispatch order fills CUs round robin: 4 blocks per CU)
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (m == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c0 = mm<4, 3>(a, b, c0);
        }
    } else if (m == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { c0 = mm<4, 3>(a, b, c0); c1 = mm<4, 5>(a, b, c1); }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], mul, add);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    // ---- 1. lane / register map ----
    std::vector<float> a(64), b(64), d(17 * 4 * 64);
    for (int l = 0; l < 64; ++l) { a[l] = 1.0f + l; b[l] = 100.0f + l; }
    float *da, *db, *dd;
    CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dd, d.size() * 4));
    CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
    int bad0 = 0, bad4 = 0;
    for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) if (d[i * 64 + l] != a[4 * (l / 4) + i] * b[l]) ++bad0;
    for (int m = 0; m < 16; ++m) for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) if (d[((1 + m) * 4 + i) * 64 + l] != a[4 * m + i] * b[l]) ++bad4;
    printf("map: D[reg i][lane l] = A[lane 4(l/4)+i] * B[lane l] (cbsz 0): %d mismatches of 256;  = A[lane 4 abid + i] * B[lane l] (cbsz 4): %d of 4096\n", bad0, bad4);
    if (bad0 || bad4) {
        printf("  cbsz 0 reg 0 lanes 0..7:"); for (int l = 0; l < 8; ++l) printf(" %g", d[l]); printf("\n  cbsz 4 abid 1 reg 0 lanes 0..7:");
        for (int l = 0; l < 8; ++l) printf(" %g", d[(2 * 4) * 64 + l]); printf("\n");
    }
    // ---- 2. chain against fmaf ----
    const int NBLK = 4096, NL = NBLK * 64;
    std::vector<float> taps(NT), win((size_t)NL * NW), om((size_t)NL * 4), of((size_t)NL * 4);
    double ssum = 0; for (int t = 0; t < NT; ++t) { taps[t] = (float)std::exp(-0.5 * (t - R) * (t - R) / 9.0); ssum += taps[t]; }
    for (int t = 0; t < NT; ++t) taps[t] = (float)(taps[t] / ssum);
    uint64_t s = 88172645463325252ull;
    for (auto& v : win) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const uint32_t r = (uint32_t)(s >> 32); v = (r & 7) == 0 ? 0.0f : (float)(r >> 8) * (1.0f / 16777216.0f) * ((r & 0x70) == 0 ? 1e-30f : 1.0f); }
    float *dt, *dw, *dom, *dof;
    CK(hipMalloc(&dt, NT * 4)); CK(hipMalloc(&dw, win.size() * 4)); CK(hipMalloc(&dom, om.size() * 4)); CK(hipMalloc(&dof, of.size() * 4));
    CK(hipMemcpy(dt, taps.data(), NT * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, win.data(), win.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_chain, dim3(NBLK), dim3(64), 0, 0, dt, dw, dom, dof);
    CK(hipMemcpy(om.data(), dom, om.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(of.data(), dof, of.size() * 4, hipMemcpyDeviceToHost));
    size_t badc = 0, badh = 0; for (size_t i = 0; i < om.size(); ++i) if (memcmp(&om[i], &of[i], 4)) ++badc;
    for (int gl = 0; gl < NL; gl += 97) for (int i = 0; i < 4; ++i) { float acc = 0; for (int t = 0; t < NT; ++t) acc = fmaf(win[(size_t)gl * NW + i + t], taps[t], acc); if (memcmp(&acc, &of[(size_t)gl * 4 + i], 4)) ++badh; }
    printf("chain: 19-tap filter, 4 outputs per lane, %d lanes (denormal-range and zero samples included): MFMA chain vs device fmaf chain %zu bit mismatches of %zu; device fmaf vs host fmaf %zu\n", NL, badc, om.size(), badh);
    // ---- 3. timing ----
    float* sink; unsigned long long* dc; CK(hipMalloc(&sink, 64)); CK(hipMalloc(&dc, 4096 * 8));
    std::vector<unsigned long long> hc(4096);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 4000;
    auto run = [&](int mode, int wps, const char* name, double insts_per_iter) -> int {
        const int blocks = cus * wps;
        hipLaunchKernelGGL(k_time, dim3(blocks), dim3(256), 0, 0, mode, 100, sink, dc);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_time, dim3(blocks), dim3(256), 0, 0, mode, iters, sink, dc);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hc.data(), dc, blocks * 8, hipMemcpyDeviceToHost));
        double lo = 1e30, hi = 0; for (int i = 0; i < blocks; ++i) { lo = std::fmin(lo, (double)hc[i]); hi = std::fmax(hi, (double)hc[i]); }
        // s_memtime / readcyclecounter ticks at 100 MHz on this part: report wall time per instruction and derive cycles from the event time at an assumed 2.4 GHz
        printf("%-58s %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD-wave = %.1f cycles at 2.4 GHz per instruction per SIMD\n", name, wps, ms,
               ms * 1e6 / (iters * insts_per_iter), ms * 1e6 / (iters * insts_per_iter) * 2.4 / wps);
        return 0;
    };
    for (int wps : {1, 2, 4}) {
        run(0, wps, "mfma 4x4x1, one accumulator (dependent chain)", 16);
        run(1, wps, "mfma 4x4x1, two accumulators alternating", 16);
        run(2, wps, "v_pk_fma_f32, 8 independent", 16);
    }
    run(3, 4, "2 waves mfma (2 acc) + 2 waves v_pk_fma per SIMD", 16);
    return 0;
}
