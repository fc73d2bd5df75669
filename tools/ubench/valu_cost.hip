// valu_cost.hip — gfx950 instruction-cost probe behind DESIGN.md §4's VALU floor for k_phosphor_rr.
//
// For each instruction class the kernels below run a long unrolled stream of INDEPENDENT instances
// (8 accumulators, 32 instances per loop trip) and report shader cycles per wave-instruction as seen
// by ONE SIMD:  cycles = (s_memtime delta of a wave) * (waves per SIMD) / instructions issued by that wave
// ... divided again by waves per SIMD gives the per-wave cadence.  Run at 1, 2 and 4 waves per SIMD
// (256-thread blocks = one wave per SIMD; 1, 2, 4 blocks per CU) so that both the single-wave issue
// cadence and the SIMD's saturated throughput are visible.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_cost tools/ubench/valu_cost.hip && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint64_t memtime() {
    uint64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// 32 instances of INSN over 8 accumulators; X is a macro taking the accumulator index
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define KERNEL_HEAD(name)                                                                    \
    __global__ __launch_bounds__(256) void name(uint64_t* cyc, float* sink, int trips, float sv, const float* tab) { \
        __shared__ float lds[4096];                                                          \
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;                      \
        __syncthreads();                                                                     \
        float a[8]; double d[8]; f32x2 p[8]; uint32_t u[8];                                  \
        for (int i = 0; i < 8; ++i) { a[i] = sv + i + threadIdx.x; d[i] = a[i]; p[i] = f32x2{a[i], a[i] + 1.f}; u[i] = (uint32_t)(a[i] * 977.f); } \
        float s0 = sv * 0.5f, s1 = sv * 0.25f;                                               \
        f32x2 sp = f32x2{s0, s1};                                                            \
        double sd = (double)sv * 0.999;                                                      \
        uint32_t la = (threadIdx.x * 4u) & 16380u;                                           \
        uint32_t lg = ((threadIdx.x * 2654435761u) >> 20) & 4092u;   /* scattered LUT-style address */ \
        (void)la; (void)lg; (void)sp; (void)sd; (void)s0; (void)s1;                          \
        const uint64_t t0 = memtime();                                                       \
        _Pragma("unroll 1") for (int t = 0; t < trips; ++t) {

#define KERNEL_TAIL                                                                          \
        }                                                                                    \
        const uint64_t t1 = memtime();                                                       \
        float acc = 0.f;                                                                     \
        for (int i = 0; i < 8; ++i) acc += a[i] + (float)d[i] + p[i].x + p[i].y + (float)u[i]; \
        if (acc == 12345.678f) sink[0] = acc + lds[threadIdx.x];                             \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;     \
    }

// --- float32 ---
#define I_FMAC_S(i) asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s0));
KERNEL_HEAD(k_fmac_sgpr) REP32(I_FMAC_S) KERNEL_TAIL
#define I_FMA_V(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
KERNEL_HEAD(k_fma_vvv) REP32(I_FMA_V) KERNEL_TAIL
#define I_MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s0));
KERNEL_HEAD(k_mul_f32) REP32(I_MUL) KERNEL_TAIL
#define I_MED3(i) asm volatile("v_med3_f32 %0, %0, 0, 1.0" : "+v"(a[i]));
KERNEL_HEAD(k_med3_f32) REP32(I_MED3) KERNEL_TAIL
#define I_MAXF(i) asm volatile("v_max_f32 %0, 0, %0" : "+v"(a[i]));
KERNEL_HEAD(k_max_f32) REP32(I_MAXF) KERNEL_TAIL
// --- packed float32 ---
#define I_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
KERNEL_HEAD(k_pk_fma_vvv) REP32(I_PKFMA) KERNEL_TAIL
#define I_PKFMA_S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "s"(sp));
KERNEL_HEAD(k_pk_fma_bcast_sgprpair) REP32(I_PKFMA_S) KERNEL_TAIL
#define I_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
KERNEL_HEAD(k_pk_mul) REP32(I_PKMUL) KERNEL_TAIL
#define I_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
KERNEL_HEAD(k_pk_add) REP32(I_PKADD) KERNEL_TAIL
// --- float64 ---
#define I_MULD(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "s"(sd));
KERNEL_HEAD(k_mul_f64) REP32(I_MULD) KERNEL_TAIL
#define I_ADDD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
KERNEL_HEAD(k_add_f64) REP32(I_ADDD) KERNEL_TAIL
#define I_FMAD(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
KERNEL_HEAD(k_fma_f64) REP32(I_FMAD) KERNEL_TAIL
#define I_MAXD(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
KERNEL_HEAD(k_max_f64) REP32(I_MAXD) KERNEL_TAIL
// --- conversions ---
#define I_CVTDF(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
KERNEL_HEAD(k_cvt_f64_f32) REP32(I_CVTDF) KERNEL_TAIL
#define I_CVTFD(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
KERNEL_HEAD(k_cvt_f32_f64) REP32(I_CVTFD) KERNEL_TAIL
#define I_CVTIF(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
KERNEL_HEAD(k_cvt_i32_f32) REP32(I_CVTIF) KERNEL_TAIL
#define I_CVTUB(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));
KERNEL_HEAD(k_cvt_f32_ubyte1) REP32(I_CVTUB) KERNEL_TAIL
#define I_RNDNE(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
KERNEL_HEAD(k_rndne_f32) REP32(I_RNDNE) KERNEL_TAIL
// --- transcendental ---
#define I_LOG(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
KERNEL_HEAD(k_log_f32) REP32(I_LOG) KERNEL_TAIL
#define I_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
KERNEL_HEAD(k_sqrt_f32) REP32(I_SQRT) KERNEL_TAIL
#define I_COS(i) asm volatile("v_cos_f32 %0, %0" : "+v"(a[i]));
KERNEL_HEAD(k_cos_f32) REP32(I_COS) KERNEL_TAIL
// --- integer ---
#define I_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
KERNEL_HEAD(k_add_u32) REP32(I_ADDU) KERNEL_TAIL
#define I_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
KERNEL_HEAD(k_lshl_add_u32) REP32(I_LSHLADD) KERNEL_TAIL
#define I_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
KERNEL_HEAD(k_mul_lo_u32) REP32(I_MULLO) KERNEL_TAIL
#define I_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
KERNEL_HEAD(k_mad_u32_u24) REP32(I_MAD24) KERNEL_TAIL
#define I_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
KERNEL_HEAD(k_xor_b32) REP32(I_XOR) KERNEL_TAIL
#define I_BFE(i) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u[i]));
KERNEL_HEAD(k_bfe_u32) REP32(I_BFE) KERNEL_TAIL
#define I_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : );
KERNEL_HEAD(k_cndmask) REP32(I_CNDMASK) KERNEL_TAIL
// --- cross-lane ---
#define I_DPP_WSHR(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
KERNEL_HEAD(k_mov_dpp_wave_shr) REP32(I_DPP_WSHR) KERNEL_TAIL
#define I_DPP_FMAC(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
KERNEL_HEAD(k_fmac_dpp_row_shr) REP32(I_DPP_FMAC) KERNEL_TAIL
#define I_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(la));
KERNEL_HEAD(k_ds_bpermute_waited) REP32(I_BPERM) KERNEL_TAIL
// --- LDS (8 reads in flight, then one wait) ---
#define I_LDS_B32(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*256" : "=v"(a[i]) : "v"(la));
KERNEL_HEAD(k_ds_read_b32_linear) REP32(I_LDS_B32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); KERNEL_TAIL
#define I_LDS_G32(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*4" : "=v"(a[i]) : "v"(lg));
KERNEL_HEAD(k_ds_read_b32_scattered) REP32(I_LDS_G32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); KERNEL_TAIL
// dependent LDS gather chain: address of the next read comes from the previous value (LUT -> LUT)
#define I_LDS_DEP(i) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %1, 0xffc, %0" : "=&v"(u[i]), "+v"(lg));
KERNEL_HEAD(k_ds_read_b32_dependent_chain) REP32(I_LDS_DEP) KERNEL_TAIL
// --- dependent-issue latency: one accumulator, 32 back-to-back dependent fmac ---
#define I_FMAC_DEP(i) asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a[0]) : "s"(s0));
KERNEL_HEAD(k_fmac_dependent_chain) REP32(I_FMAC_DEP) KERNEL_TAIL
#define I_FMAC_DEP2(i) asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a[i & 1]) : "s"(s0));
KERNEL_HEAD(k_fmac_two_chains) REP32(I_FMAC_DEP2) KERNEL_TAIL
#define I_PKFMA_DEP(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(p[i & 1]) : "v"(p[2 + (i & 3)]), "s"(sp));
KERNEL_HEAD(k_pk_fma_two_chains) REP32(I_PKFMA_DEP) KERNEL_TAIL

typedef void (*kern_t)(uint64_t*, float*, int, float, const float*);
struct Case { const char* name; kern_t fn; };

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    uint64_t* cyc; float* sink; float* tab;
    CK(hipMalloc(&cyc, sizeof(uint64_t) * cus * 8 * 4));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&tab, 4096 * 4));
    std::vector<Case> cases = {
#define C(k) {#k, k}
        C(k_fmac_sgpr), C(k_fma_vvv), C(k_mul_f32), C(k_med3_f32), C(k_max_f32),
        C(k_pk_fma_vvv), C(k_pk_fma_bcast_sgprpair), C(k_pk_mul), C(k_pk_add),
        C(k_mul_f64), C(k_add_f64), C(k_fma_f64), C(k_max_f64),
        C(k_cvt_f64_f32), C(k_cvt_f32_f64), C(k_cvt_i32_f32), C(k_cvt_f32_ubyte1), C(k_rndne_f32),
        C(k_log_f32), C(k_sqrt_f32), C(k_cos_f32),
        C(k_add_u32), C(k_lshl_add_u32), C(k_mul_lo_u32), C(k_mad_u32_u24), C(k_xor_b32), C(k_bfe_u32), C(k_cndmask),
        C(k_mov_dpp_wave_shr), C(k_fmac_dpp_row_shr), C(k_ds_bpermute_waited),
        C(k_ds_read_b32_linear), C(k_ds_read_b32_scattered), C(k_ds_read_b32_dependent_chain),
        C(k_fmac_dependent_chain), C(k_fmac_two_chains), C(k_pk_fma_two_chains),
    };
    const int trips = 2000;
    printf("%-34s %12s %12s %12s   (shader cycles per wave-instruction at the SIMD: wave cycles / instrs / waves-per-SIMD)\n",
           "instruction stream", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    for (auto& c : cases) {
        printf("%-34s", c.name);
        for (int wps : {1, 2, 4}) {
            const int blocks = cus * wps;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, cyc, sink, 50, 1.0f, tab);   // warm-up
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, cyc, sink, trips, 1.0f, tab);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint64_t> h(blocks * 4);
            CK(hipMemcpy(h.data(), cyc, sizeof(uint64_t) * blocks * 4, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double med = (double)h[h.size() / 2];
            const double per = med / ((double)trips * 32.0) / wps;
            // wall-clock cross-check: ns per instruction per SIMD
            const double ns = (double)ms * 1e6 / ((double)trips * 32.0) / wps;
            printf(" %6.2f/%5.2fns", per, ns);
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        printf("\n");
    }
    return 0;
}
