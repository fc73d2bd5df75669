// crtfx_blur.hip.h — the split Gaussian bloom: k_sb_src / k_sb_rows / k_sb_cols / k_sb_cols_lds (any radius)
// (one of the parts of crtfx_kernels.hip.h; see the chain overview there and DESIGN.md §3)
#pragma once
#include "crtfx_common.hip.h"

namespace crtfx {

#ifdef CRTFX_MAIN_TU
// ---------------------------------------------------------------------------------------
// Split Gaussian bloom (ref:609-610 for ANY sigma): the blur as its own three kernels, the rest of the chain in the
// pointwise kernels below (which add strength * blur from the full-resolution plane these leave in P.ds).
//   k_sb_src   plane A = bloom source (a1..a4 + threshold) of every pixel
//   k_sb_rows  plane B = row pass of A     (taps left to right, fmaf, BORDER_REPLICATE: oracle/crt_oracle.c orc_sepblur_f32)
//   k_sb_cols  plane A = column pass of B  (taps top to bottom, fmaf)
// The fused register-window kernels keep 2R + 1 rows in registers and redo 2R halo columns per 64-px strip, which is
// right for the GUI's radii (<= 30) and hopeless far beyond them; these are output-stationary instead: a thread owns
// SB_N neighbouring outputs ALONG the pass direction and walks the 2R + SB_N source samples they touch once, each
// sample feeding all SB_N accumulators (24 / 32 FMAs per sample loaded), so the work per output is the 2R + 1 FMAs of
// the definition (+ SB_N - 1 with a zero tap) at any radius, and the taps come from a zero-padded device array
// (wave-uniform loads), not from the kernel arguments: no radius limit, no per-radius build.
// Accumulation order per output = the oracle's (k = 0 .. 2R), so the planes are bit-exact; a zero tap adds +0.
// ---------------------------------------------------------------------------------------
constexpr int SB_N = 8;                      // outputs per thread along the pass direction
constexpr int SB_SPAN = 64 * SB_N;           // row pass: pixels per wavefront
#ifndef SB_CH_STEPS
#define SB_CH_STEPS 264
#endif
#ifndef SB_RW
#define SB_RW 1      // rows of a row-pass block = its wavefronts (1: no barrier partner, finer tail; 4K R = 32: 80.5 vs 84 us)
#endif
#ifndef SBC_W
#define SBC_W 4
#endif
constexpr int SB_CH = SB_CH_STEPS;           // row pass: source steps staged per LDS chunk (264: radii <= 128 in one chunk)
constexpr int SB_TILE = SB_SPAN + SB_CH;
constexpr int SB_PLANE = SB_TILE + SB_TILE / 32 + 8;   // one channel of a tile; a pad word per 32 px keeps the 8-px lane stride off the same banks
                                                       // (a pad word per 8 px — no conflict at all — costs a resident block per CU: 92 vs 86 us at R = 32)
constexpr int sb_steps(int R) { return (2 * R + SB_N + 7) & ~7; }      // source steps per output run, rounded up to the unroll
constexpr int sb_tpad_len(int R) { return sb_steps(R) + 16; }          // tpad[i] = taps[i - (SB_N - 1)], zero elsewhere; the device array is [tpad | tpadB], tpadB[i] = tpad[i + 1]

template <int PIX>
__global__ __launch_bounds__(256) void k_sb_src(KParams Pin, KFrame F) {
    KParams P = Pin;
    P.pix = PIX;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.W || y >= P.H) return;
    float r, g, b;
    fetch_graded(P, F, y, x, r, g, b);
    *reinterpret_cast<F3*>(P.ds + ((size_t)y * P.W + x) * 3) = F3{bloom_src(P, r), bloom_src(P, g), bloom_src(P, b)};
}

// The SB_N accumulators of a thread as SB_N / 2 packed pairs (outputs 2p, 2p + 1), one v_pk_fma_f32 per pair and sample:
// sample step m is tap m - j of output j, so a pair wants (tap[i], tap[i - 1]) with i = m + SB_N - 1 - 2p in the padded
// array — an aligned 64-bit scalar pair of tpad for odd i, of the one-float-shifted copy tpadB for even i (both swapped).
// U = the step inside the unrolled group of 8; TA / TB = the 16-float windows of tpad / tpadB at the group's first step.
template <int U>
__device__ __forceinline__ void sb_fma(f32x2 (&acc)[SB_N / 2], f32x2 w, bool whigh, const unsigned long long (&TA)[8], const unsigned long long (&TB)[8]) {
#pragma unroll
    for (int p = 0; p < SB_N / 2; ++p) {
        constexpr int base = U + SB_N - 1;
        const int i = base - 2 * p;
#ifdef SB_SCALAR_FMA      // A/B: two v_fmac_f32 with an SGPR tap instead of one packed FMA
        const unsigned long long tpair = (i & 1) ? TA[(i - 1) / 2] : TB[(i - 2) / 2];
        const float t_hi = __builtin_bit_cast(float, (uint32_t)(tpair >> 32)), t_lo = __builtin_bit_cast(float, (uint32_t)tpair);
        const float wv = whigh ? w.y : w.x;
        asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[p].x) : "s"(t_hi), "v"(wv));
        asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[p].y) : "s"(t_lo), "v"(wv));
#else
        pk_fma_bcast(acc[p], w, whigh, (i & 1) ? TA[(i - 1) / 2] : TB[(i - 2) / 2], true);
#endif
    }
}

// one wavefront = SB_SPAN pixels of one row; a block = 4 rows.  Lane L owns pixels 8L .. 8L+7 of the span.
// tp64: [tpad | tpadB] as 64-bit pairs, npairs each.
// PIX >= 0: the bloom source is computed from the frame while the tile is staged (k_sb_src folded in: saves writing and
// re-reading a float32 plane; the halo pixels are graded (SB_SPAN + 2R + 8) / SB_SPAN times); PIX = -1: src is plane A.
template <int PIX>
__global__ __launch_bounds__(64 * SB_RW) void k_sb_rows(KParams Pin, KFrame F, const float* __restrict__ src, float* __restrict__ dst, int R,
                                                 const unsigned long long* __restrict__ tp64, int npairs) {
    __shared__ float tile[SB_RW][3][SB_PLANE];
    KParams P = Pin;
    if constexpr (PIX >= 0) P.pix = PIX;
    const int H = P.H, W = P.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y = blockIdx.y * SB_RW + wave;
    const int yc = min(y, H - 1);                   // rows past the bottom redo the last row without storing
    const int wx0 = blockIdx.x * SB_SPAN;
    const float* __restrict__ srow = src + (size_t)yc * W * 3;
    f32x2 acc[3][SB_N / 2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int p = 0; p < SB_N / 2; ++p) acc[c][p] = f32x2{0.0f, 0.0f};
    const int S8 = sb_steps(R);
    float (*tl)[SB_PLANE] = tile[wave];
    for (int c0 = 0; c0 < S8; c0 += SB_CH) {
        const int nsteps = min(SB_CH, S8 - c0);
        const int gx0 = wx0 - R + c0;               // image column of tile pixel 0
        __syncthreads();
        // four tile pixels per lane and round, every stage's loads issued together (the stage-by-stage fetch_graded waits
        // for memory twice per pixel: ten dependent round trips per tile made this kernel latency-bound, 95 us at R = 32
        // against 19 us of packed FMAs); slots past the tile's end redo its last pixel
        const int n_t = SB_SPAN + nsteps;
        for (int tb = 0; tb < n_t; tb += 256) {
            int tt[4], px[4];
            float v[4][3];
#pragma unroll
            for (int i = 0; i < 4; ++i) { tt[i] = min(tb + 64 * i + lane, n_t - 1); px[i] = min(max(gx0 + tt[i], 0), W - 1); }      // BORDER_REPLICATE
            if constexpr (PIX >= 0) {
                int xs[4], ys = yc;
                if (P.flags & CRTFX_F_PIXELATE) {
                    ys = P.ymap[yc];
#pragma unroll
                    for (int i = 0; i < 4; ++i) xs[i] = P.xmap[px[i]];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) xs[i] = px[i];
                }
                const uint32_t row = (uint32_t)ys * (uint32_t)W * 3u;
                RawRGB raw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int xr = xs[i], xb = xs[i];
                    if (P.ab != 0) { xr = wrap(xs[i] - P.ab, W); xb = wrap(xs[i] + P.ab, W); }      // ref:573-575
                    raw[i] = load_raw(PIX, F.in, row + (uint32_t)xr * 3u, row + (uint32_t)xs[i] * 3u + 1u, row + (uint32_t)xb * 3u + 2u);
                }
                if (P.grade_lut && (P.flags & CRTFX_F_GAMMA)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v[i][0] = P.grade_lut[raw[i].r]; v[i][1] = P.grade_lut[256 + raw[i].g]; v[i][2] = P.grade_lut[512 + raw[i].b]; }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i][0] = norm_px(PIX, raw[i].r); v[i][1] = norm_px(PIX, raw[i].g); v[i][2] = norm_px(PIX, raw[i].b);
                        grade(P, v[i][0], v[i][1], v[i][2]);
                    }
                }
                if (F.overlay_before) {
                    uint32_t ov[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) ov[i] = reinterpret_cast<const uint32_t*>(F.overlay_before)[(uint32_t)yc * (uint32_t)W + (uint32_t)px[i]];
#pragma unroll
                    for (int i = 0; i < 4; ++i) overlay_blend_px<float>(ov[i], v[i][0], v[i][1], v[i][2]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i][0] = bloom_src(P, v[i][0]); v[i][1] = bloom_src(P, v[i][1]); v[i][2] = bloom_src(P, v[i][2]); }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const F3 t3 = *reinterpret_cast<const F3*>(srow + (size_t)px[i] * 3); v[i][0] = t3.x; v[i][1] = t3.y; v[i][2] = t3.z; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int a = tt[i] + (tt[i] >> 5);
                tl[0][a] = v[i][0]; tl[1][a] = v[i][1]; tl[2][a] = v[i][2];
            }
        }
        __syncthreads();
        const unsigned long long* __restrict__ pa = tp64 + (c0 >> 1);
        const unsigned long long* __restrict__ pb = pa + npairs;
        unsigned long long TA[8], TB[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { TA[i] = pa[i]; TB[i] = pb[i]; }
        for (int m8 = 0; m8 < nsteps; m8 += 8) {
            const int q = lane + (m8 >> 3);          // tile pixel 8q + u: the pad term (8q + u) >> 5 = q >> 2 for every u < 8
            const int base = 8 * q + (q >> 2);
            f32x2 s[3][4];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int h = 0; h < 4; ++h) s[c][h] = f32x2{tl[c][base + 2 * h], tl[c][base + 2 * h + 1]};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sb_fma<0>(acc[c], s[c][0], false, TA, TB); sb_fma<1>(acc[c], s[c][0], true, TA, TB);
                sb_fma<2>(acc[c], s[c][1], false, TA, TB); sb_fma<3>(acc[c], s[c][1], true, TA, TB);
                sb_fma<4>(acc[c], s[c][2], false, TA, TB); sb_fma<5>(acc[c], s[c][2], true, TA, TB);
                sb_fma<6>(acc[c], s[c][3], false, TA, TB); sb_fma<7>(acc[c], s[c][3], true, TA, TB);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { TA[i] = TA[i + 4]; TB[i] = TB[i + 4]; TA[i + 4] = pa[(m8 >> 1) + 8 + i]; TB[i + 4] = pb[(m8 >> 1) + 8 + i]; }
        }
    }
    if (y >= H) return;
    float* __restrict__ drow = dst + (size_t)y * W * 3;
    const int X = wx0 + SB_N * lane;
#define SB_A(j, c) acc[c][(j) >> 1][(j) & 1]
    if (X + SB_N <= W && ((W & 3) == 0)) {          // 24 floats from a 16-byte aligned address
        float4* d4 = reinterpret_cast<float4*>(drow + (size_t)X * 3);
        d4[0] = make_float4(SB_A(0, 0), SB_A(0, 1), SB_A(0, 2), SB_A(1, 0));
        d4[1] = make_float4(SB_A(1, 1), SB_A(1, 2), SB_A(2, 0), SB_A(2, 1));
        d4[2] = make_float4(SB_A(2, 2), SB_A(3, 0), SB_A(3, 1), SB_A(3, 2));
        d4[3] = make_float4(SB_A(4, 0), SB_A(4, 1), SB_A(4, 2), SB_A(5, 0));
        d4[4] = make_float4(SB_A(5, 1), SB_A(5, 2), SB_A(6, 0), SB_A(6, 1));
        d4[5] = make_float4(SB_A(6, 2), SB_A(7, 0), SB_A(7, 1), SB_A(7, 2));
    } else {
#pragma unroll
        for (int j = 0; j < SB_N; ++j)
            if (X + j < W) *reinterpret_cast<F3*>(drow + (size_t)(X + j) * 3) = F3{SB_A(j, 0), SB_A(j, 1), SB_A(j, 2)};
    }
#undef SB_A
}

// one wavefront = 64 * VEC neighbouring floats of SB_N output rows (a row = 3W floats; channels do not matter here);
// a block = 4 such bands one below the other.  Source rows come straight from global memory, one coalesced load per step.
template <int VEC>
__global__ __launch_bounds__(256) void k_sb_cols(const float* __restrict__ src, float* __restrict__ dst, int H, int rowlen, int R,
                                                 const unsigned long long* __restrict__ tp64, int npairs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Y = (blockIdx.y * 4 + wave) * SB_N;
    if (Y >= H) return;                              // whole wavefront
    const int i0 = (blockIdx.x * 64 + lane) * VEC;
    const bool live = i0 < rowlen;                   // VEC = 4 only when rowlen % 4 == 0
    const int ic = live ? i0 : 0;
    constexpr int NV = VEC == 4 ? 4 : 2;             // VEC = 1: the sample sits in the low half of a pair
    f32x2 acc[NV][SB_N / 2];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int p = 0; p < SB_N / 2; ++p) acc[v][p] = f32x2{0.0f, 0.0f};
    const int S8 = sb_steps(R);
    const unsigned long long* __restrict__ pa = tp64;
    const unsigned long long* __restrict__ pb = pa + npairs;
    unsigned long long TA[8], TB[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { TA[i] = pa[i]; TB[i] = pb[i]; }
    for (int m8 = 0; m8 < S8; m8 += 8) {
        f32x2 s[8][2];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sy = min(max(Y - R + m8 + u, 0), H - 1);     // BORDER_REPLICATE
            const float* __restrict__ p = src + (size_t)sy * rowlen + ic;
            if constexpr (VEC == 4) { const float4 t4 = *reinterpret_cast<const float4*>(p); s[u][0] = f32x2{t4.x, t4.y}; s[u][1] = f32x2{t4.z, t4.w}; }
            else { s[u][0] = f32x2{*p, 0.0f}; s[u][1] = s[u][0]; }
        }
#define SB_STEP(u)                                                                                  \
        sb_fma<u>(acc[0], s[u][0], false, TA, TB);                                                  \
        if constexpr (VEC == 4) { sb_fma<u>(acc[1], s[u][0], true, TA, TB); sb_fma<u>(acc[2], s[u][1], false, TA, TB); sb_fma<u>(acc[3], s[u][1], true, TA, TB); }
        SB_STEP(0) SB_STEP(1) SB_STEP(2) SB_STEP(3) SB_STEP(4) SB_STEP(5) SB_STEP(6) SB_STEP(7)
#undef SB_STEP
#pragma unroll
        for (int i = 0; i < 4; ++i) { TA[i] = TA[i + 4]; TB[i] = TB[i + 4]; TA[i + 4] = pa[(m8 >> 1) + 8 + i]; TB[i + 4] = pb[(m8 >> 1) + 8 + i]; }
    }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < SB_N; ++j) {
        if (Y + j < H) {
            float* d = dst + (size_t)(Y + j) * rowlen + i0;
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(d) = make_float4(acc[0][j >> 1][j & 1], acc[1][j >> 1][j & 1], acc[2][j >> 1][j & 1], acc[3][j >> 1][j & 1]);
            else *d = acc[0][j >> 1][j & 1];
        }
    }
}

// k_sb_cols<4> re-reads every source row once per 8-row band ((8 + 2R) / 8 times: 9x at R = 32, 32x at R = 126 — measured
// L2-bound at 9-12 TB/s, 96 / 267 us per 4K frame).  Here the four bands of a block share the rows through LDS: the block
// walks the 32 + 2R source rows its 32 output rows touch in chunks of SBC_ROWS rows (each wave stages 8 rows with one
// float4 load per lane and row), and every wave runs the 8-step groups of the chunk that fall inside its own tap range
// (band w is 8w rows lower, so its step index is 8w behind: still a multiple of 8, the tap windows stay aligned).
constexpr int SBC_ROWS = 8 * SBC_W;
__global__ __launch_bounds__(64 * SBC_W) void k_sb_cols_lds(const float* __restrict__ src, float* __restrict__ dst, int H, int rowlen, int R,
                                                     const unsigned long long* __restrict__ tp64, int npairs, int nbx, int nby) {
    __shared__ f32x4 tile[SBC_ROWS][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Blocks are dealt to the 8 XCDs round-robin in dispatch order; give each XCD a contiguous run of the (column, band)
    // list with the band running fastest, so that the blocks resident on one XCD at a time are vertical neighbours and
    // find each other's source rows (all but 32 of their 32 + 2R) in that XCD's L2.
    // (1-D grid of 8 * ceil(nbx * nby / 8) blocks: every XCD gets the same count.)
    const int total = nbx * nby;
    const int id = blockIdx.x;
    const int per = gridDim.x >> 3;
    const int v = (id & 7) * per + (id >> 3);
    if (v >= total) return;                          // whole block, before any barrier
    const int bx = v / nby, by = v - bx * nby;
    const int Y0 = by * (SBC_W * SB_N);
    const int Y = Y0 + wave * SB_N;
    const int i0 = (bx * 64 + lane) * 4;
    const bool live = i0 < rowlen;                   // rowlen % 4 == 0
    const int ic = live ? i0 : 0;
    f32x2 acc[4][SB_N / 2];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int p = 0; p < SB_N / 2; ++p) acc[v][p] = f32x2{0.0f, 0.0f};
    const int S8 = sb_steps(R);
    const unsigned long long* __restrict__ pa = tp64;
    const unsigned long long* __restrict__ pb = pa + npairs;
    const int nch = ((SBC_W - 1) * SB_N + S8 + SBC_ROWS - 1) / SBC_ROWS;      // the lowest band's last step reads row Y0 + 24 - R + S8 - 1
    for (int c = 0; c < nch; ++c) {
        const int r0 = Y0 - R + c * SBC_ROWS;
        __syncthreads();
        f32x4 ld[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int sy = min(max(r0 + 8 * wave + k, 0), H - 1);      // BORDER_REPLICATE
            ld[k] = *reinterpret_cast<const f32x4*>(src + (size_t)sy * rowlen + ic);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[8 * wave + k][lane] = ld[k];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < SBC_ROWS / 8; ++g) {
            const int m8 = c * SBC_ROWS - 8 * wave + 8 * g;             // this band's step index of chunk row 8g
            if (m8 >= 0 && m8 < S8 && Y < H) {                          // wave-uniform
                unsigned long long TA[8], TB[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { TA[i] = pa[(m8 >> 1) + i]; TB[i] = pb[(m8 >> 1) + i]; }
                f32x2 s[8][2];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const f32x4 t4 = tile[8 * g + u][lane]; s[u][0] = f32x2{t4.x, t4.y}; s[u][1] = f32x2{t4.z, t4.w}; }
#define SB_STEP(u) sb_fma<u>(acc[0], s[u][0], false, TA, TB); sb_fma<u>(acc[1], s[u][0], true, TA, TB); sb_fma<u>(acc[2], s[u][1], false, TA, TB); sb_fma<u>(acc[3], s[u][1], true, TA, TB);
                SB_STEP(0) SB_STEP(1) SB_STEP(2) SB_STEP(3) SB_STEP(4) SB_STEP(5) SB_STEP(6) SB_STEP(7)
#undef SB_STEP
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < SB_N; ++j)
        if (Y + j < H)
            *reinterpret_cast<float4*>(dst + (size_t)(Y + j) * rowlen + i0) = make_float4(acc[0][j >> 1][j & 1], acc[1][j >> 1][j & 1], acc[2][j >> 1][j & 1], acc[3][j >> 1][j & 1]);
}

#endif  // CRTFX_MAIN_TU

}  // namespace crtfx
