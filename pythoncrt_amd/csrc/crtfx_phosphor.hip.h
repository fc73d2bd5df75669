// crtfx_phosphor.hip.h — the fused Gaussian-bloom chain: k_phosphor<-1> (LDS ring), k_phosphor_rr (register window), k_phosphor_cc (column owner)
// (one of the parts of crtfx_kernels.hip.h; see the chain overview there and DESIGN.md §3)
#pragma once
#include "crtfx_common.hip.h"

namespace crtfx {


// ---------------------------------------------------------------------------------------
// k_phosphor — grade + separable Gaussian bloom + masks + grain.
//
// A block owns a 64-px-wide column strip over `seg_rows` output rows and streams down it in
// blocks of NB rows.  Per block of rows:
//   A  192 threads grade the (64 + 2*pad)-px-wide halo row segments into LDS (planar per channel)
//   B  horizontal pass: a lane produces 4 adjacent pixels of one channel from 16-byte LDS reads,
//      taps accumulated left to right with fmaf (OpenCV RowFilter order) -> ring of H-pass rows
//   C1 vertical pass: wave c owns channel c, lane = column; each ring row is read once and fed
//      to the NB register-resident output rows, taps top to bottom with fmaf (ColumnFilter order)
//   C2 per-pixel: img + strength*blur, triad/scanline/vignette/flicker/grain, store
// LDS: staging NB x 3 x (64+2pad), ring (NB+2R) x 3 x 64, blur NB x 3 x 64, LUTs 2 x 1028 floats.
// RT >= 0 fixes the radius at compile time (loops unroll, dead taps vanish); RT < 0 = runtime R.
// ---------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(K1_THREADS) void k_phosphor(KParams P, KFrame F, KOut O, int seg_rows) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int R = RT >= 0 ? RT : P.R;
    const int pad = (R + 3) & ~3;
    const int SWP = TW + 2 * pad;
    const int ring_rows = NB + 2 * R;
    float* stg = smem;                          // [NB][3][SWP]
    float* ring = stg + NB * 3 * SWP;           // [ring_rows][3][TW]
    float* blr = ring + ring_rows * 3 * TW;     // [NB][3][TW]
    float* lut = blr + NB * 3 * TW;             // [2][LUT_STRIDE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int H = P.H, W = P.W;
    const int x0 = blockIdx.x * TW;
    const int y_begin = blockIdx.y * seg_rows;
    const int y_end = min(H, y_begin + seg_rows);
    if (y_begin >= H) return;

    if ((P.flags & CRTFX_F_TRIAD) && (P.flags & CRTFX_F_TRIAD_LUT)) {
        for (int i = tid; i < LUT_N; i += K1_THREADS) { lut[i] = P.lut_g[i]; lut[LUT_STRIDE + i] = P.lut_inv[i]; }
    }
    const float* taps = P.taps;                 // kernarg-resident
    const int ring_base = y_begin - R;          // ring slot of row y is (y - ring_base) % ring_rows

    for (int hb = y_begin - R; hb < y_end + R; hb += NB) {
        // ---- A: grade halo rows [hb, hb+NB) into the staging tile -------------------------
        const int nrows = min(NB, y_end + R - hb);
        for (int it = tid; it < nrows * SWP; it += K1_THREADS) {
            const int j = it / SWP, i = it - j * SWP;
            const int y = min(max(hb + j, 0), H - 1);           // BORDER_REPLICATE
            const int x = min(max(x0 - pad + i, 0), W - 1);
            float r, g, b;
            fetch_graded(P, F, y, x, r, g, b);
            float* s = stg + (j * 3) * SWP + i;
            s[0] = bloom_src(P, r); s[SWP] = bloom_src(P, g); s[2 * SWP] = bloom_src(P, b);
        }
        __syncthreads();
        // ---- B: horizontal pass -> ring ------------------------------------------------------
        for (int it = tid; it < nrows * 48; it += K1_THREADS) {
            const int j = it / 48, rem = it - j * 48;
            const int c = rem >> 4, gq = rem & 15;
            const float4* srow = reinterpret_cast<const float4*>(stg + (j * 3 + c) * SWP) + gq;
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const int nchunk = (2 * pad + 4) >> 2;
            const int off = pad - R;
#pragma unroll
            for (int q = 0; q < nchunk; ++q) {      // compile-time bound when RT >= 0
                const float4 v = srow[q];
                const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int t = 4 * q + e - i - off;   // tap index: window position minus output position
                        if (t >= 0 && t <= 2 * R) acc[i] = fmaf(ve[e], taps[t], acc[i]);
                    }
            }
            const int slot = (hb + j - ring_base) % ring_rows;
            reinterpret_cast<float4*>(ring + (slot * 3 + c) * TW)[gq] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __syncthreads();
        // ---- C1: vertical pass for the output rows now covered ---------------------------
        const int out_lo = max(y_begin, hb - R);
        const int out_hi = min(y_end, hb + NB - R);
        const int jrows = out_hi - out_lo;
        if (jrows > 0) {
            {
                const int c = tid >> 6;          // wavefront = channel
                float acc[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[j] = 0.0f;
                // Always sweep the full NB + 2R window (a compile-time trip count when RT >= 0, so the
                // tap index rr - j is static and dead taps vanish).  When fewer than NB rows are due
                // (first / last block of the segment) the extra ring rows are stale; they only feed
                // accumulators of rows >= jrows, which are never read.
                int slot = (out_lo - R - ring_base) % ring_rows;
#pragma unroll
                for (int rr = 0; rr < NB + 2 * R; ++rr) {
                    const float v = ring[(slot * 3 + c) * TW + lane];
                    slot = slot + 1 == ring_rows ? 0 : slot + 1;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int t = rr - j;
                        if (t >= 0 && t <= 2 * R) acc[j] = fmaf(v, taps[t], acc[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) blr[(j * 3 + c) * TW + lane] = acc[j];
            }
            __syncthreads();
            // ---- C2: combine + masks + store ----------------------------------------------
            for (int it = tid; it < jrows * TW; it += K1_THREADS) {   // 192 = 3*64: a wavefront stays on one row
                const int j = it >> 6;
                const int y = out_lo + j;
                const int x = x0 + lane;
                const bool live = x < W;
                float r = 0, g = 0, b = 0;
                PixMasks M{};
                if (live) {
                    M = load_masks(P, F, y, x);
                    fetch_graded(P, F, y, x, r, g, b);
                    // ref:611 img = clip(img + bloom_strength * blur)
                    r = clip01(r + P.bloom_strength * blr[(j * 3 + 0) * TW + lane]);
                    g = clip01(g + P.bloom_strength * blr[(j * 3 + 1) * TW + lane]);
                    b = clip01(b + P.bloom_strength * blr[(j * 3 + 2) * TW + lane]);
                }
                emit_pixel(P, F, O, y, x0, lane, live, M, r, g, b, lut, lut + LUT_STRIDE);
            }
        }
        // next A overwrites stg (last read in B, two barriers ago); next B overwrites ring rows
        // older than this block's window; next C1 overwrites blr after the two barriers above.
    }
}

// ---------------------------------------------------------------------------------------
// k_phosphor_rr — the same stage chain as k_phosphor for a compile-time radius RT >= 1, built
// around what the phase stamps showed (profiles/r01_phase_stamps.txt): the blur arithmetic is
// ~12 % of the time; exposed memory latency in the two pointwise phases was 75 %.
//
//   * 256 threads.  Waves 0-2 own one colour channel each in the V pass; all four share the
//     pointwise phases (the NB = 8 output rows of a block split 2-2-2-2).
//   * V pass on a REGISTER window: thread (c = wave, lane = column) keeps the last 2R + NB
//     H-pass values of its column in registers, appends NB rows per block, forms output row j
//     from win[j .. j+2R] oldest first (the oracle's ColumnFilter order) and shifts the window
//     down by NB (2R moves per 8(2R+1) FMAs).  Indices are compile-time constants.  The result
//     overwrites the H-pass value it replaces in LDS (same thread, same address).
//   * phase A is software-pipelined: the uint8 bytes of the NEXT block of rows are requested
//     before the blur phases of the current block and consumed one iteration later.
//   * centre pixels needed again by C2 (img + strength*blur) wait in a small LDS ring of packed
//     bytes instead of being re-fetched; per-column constants (triad RGB, vignette nx^2) sit in
//     registers, per-row ones (scanline gain, vignette ny^2) in LDS.
// LDS at R = 9: staging 8.4 KB + rows 6 KB + LUTs 8.2 KB + centre ring 8 KB + row table ~1.5 KB.
// ---------------------------------------------------------------------------------------
#ifdef CRTFX_STAMP
// Diagnostic build (tools/phase_profile.py): where does a block iteration spend its cycles?
// Never quote this build's run time; read the shares.  Stamp values go only to O.dbg.
#define STAMP(slot) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
                         __builtin_amdgcn_sched_barrier(0); stamp_sum[slot] += t__ - stamp_last; stamp_last = t__; } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

#ifndef CRTFX_RR_WAVES
#define CRTFX_RR_WAVES 3     // min waves per SIMD the register allocator must leave room for (4 forces spills)
#endif

// Wave priorities (s_setprio) of the sections of a trip: the issue arbiter prefers the higher one when several of a SIMD's
// waves are ready.  VH: the packed-FMA bursts (V pass, H pass); C2: the pointwise tail; A: everything else of a consumer
// wave (LDS traffic, prefetch, barriers); HELP: the helper wave.
#ifndef CC_P_VH
#define CC_P_VH 2
#endif
#ifndef CC_P_C2
#define CC_P_C2 1
#endif
#ifndef CC_P_A
#define CC_P_A 0
#endif
#ifndef CC_P_HELP
#define CC_P_HELP 0
#endif
#define CC_PRIO(x) __builtin_amdgcn_s_setprio(x)
#ifndef RR_P_VH
#define RR_P_VH 2
#endif
#ifndef RR_P_C2
#define RR_P_C2 1
#endif
constexpr int RR_THREADS = 256;
typedef __attribute__((address_space(3))) volatile f32x4 lds_cv_f32x4;   // LDS-space, so the read stays a ds_ op

__host__ __device__ constexpr int rr_pad(int R) { return (R + 3) & ~3; }
__host__ __device__ constexpr int rr_swp(int R) { return TW + 2 * rr_pad(R); }
// staging row stride in floats: a multiple of 64 dwords, so the channel planes a ds_read_b128 lane
// group straddles start on the same bank and its 16-byte slots stay disjoint (stride 88 cost ~2x).
// Half frames take a 32-dword multiple instead (96 for every radius) and park their centre pixels as three 16-bit
// planes: 45.9 -> 39.6 KB of LDS per block, i.e. 4 resident blocks per CU like the uint8 build instead of 3.
__host__ __device__ constexpr int rr_sws(int R, int pix = 0) { return pix ? (rr_swp(R) + 31) & ~31 : (rr_swp(R) + 63) & ~63; }
__host__ __device__ constexpr int rr_cring(int R) { return R + 2 * NB; }   // exact: LDS is what caps blocks per CU
// LDS floats: staging, two H/blur row tiles, LUTs, centre ring (u32); then per-row table + pixelate rows
// The runtime-gate build (uint8 frames) parks the GRADED float pixel (3 floats) instead of the packed bytes: it is
// register-limited to 3 resident blocks per CU anyway, so the extra LDS is free and C2 does not redo a1 + a4 (with
// --gamma that is three powf per pixel).
__host__ __device__ constexpr int rr_cring_floats(int R, int pix, bool runtime) {
    return runtime ? rr_cring(R) * TW * 3 : (pix ? (rr_cring(R) * TW * 3 + 1) / 2 : rr_cring(R) * TW);
}
// The gate-folded uint8 build also keeps u / 255.0 for the 256 sample codes in LDS (1 KB): a table read replaces the
// convert + corrected-reciprocal arithmetic of a1 in the A phase and again for the parked centre pixel in C2.
__host__ __device__ constexpr int rr_lds_fixed_floats(int R, int pix, bool runtime = false) {
    return NB * 3 * rr_sws(R, pix || runtime) + 2 * NB * 3 * TW + 2 * LUT_STRIDE + rr_cring_floats(R, pix, runtime) + ((!pix && !runtime) ? 256 : 0);
}

// SF: the stage gates (crtfx_params.flags without CRTFX_F_WARP, which k_phosphor never reads) as a
// compile-time constant, or SF_RUNTIME.  With the gates folded the dead stages, their parameters
// (SGPRs: the runtime-flag build spills ~450 v_readlane/v_writelane) and their branches vanish:
// 178 -> 144 us per 4K frame at equal source.  The host picks the instantiation whose SF equals
// the launch's flags, else the runtime-flag one.
constexpr uint32_t SF_RUNTIME = 0xFFFFFFFFu;
constexpr uint32_t SF_FULL = SF_FULL_GATES;

// The gate-folded build sits right at the 128-VGPR boundary (127..129 depending on small edits):
// one register over and it drops from 4 to 3 waves per SIMD, i.e. from 4 to 3 resident blocks per
// CU and a second, partial round of blocks (+22 % time).  It is therefore pinned to 4 waves/SIMD;
// the runtime-flag build needs ~147 VGPRs and would spill under that cap.
// PIX: pixel format of the frames (folded like the gates); half frames park 2 dwords per centre pixel.
// Radii 13..30 (bloom sigma up to 10, the reference GUI's range): the register window (2R + 8 values) no longer
// fits 128 VGPRs, so those builds run 3 (R <= 20) or 2 resident blocks per CU.
__host__ __device__ constexpr int rr_min_waves(int R, bool folded) {
    return R <= 12 ? (folded ? 4 : CRTFX_RR_WAVES) : (R <= 20 ? (folded ? 3 : 2) : 2);
}
template <int RT, uint32_t SF, int PIX = 0>
__global__ __launch_bounds__(RR_THREADS, rr_min_waves(RT, SF != 0xFFFFFFFFu)) void k_phosphor_rr(KParams Pin, KGroup G, int seg_rows) {
    const KFrame F = G.f[blockIdx.z];
    KOut O = G.o[blockIdx.z];
    KParams P = Pin;
    if constexpr (SF != SF_RUNTIME) P.flags = SF;
    P.pix = PIX;
    O.pix = PIX;
    // declared as float4 so that the 16-byte alignment of the dynamic LDS base is part of the type:
    // with a float[] base hipcc splits every 16-byte LDS access into ds_read2_b32/_b64 pairs, which
    // at a 16-byte lane stride are 4-way / 2-way bank conflicts (ds_read_b128 is conflict-free).
    extern __shared__ float4 smem4[];
    float* smem = reinterpret_cast<float*>(smem4);
    constexpr int R = RT, K = 2 * R + 1;
    constexpr int pad = rr_pad(R);
    constexpr int SWP = rr_swp(R);
    constexpr int SWS = rr_sws(R, PIX || SF == 0xFFFFFFFFu);     // 32-dword multiple for the builds whose LDS budget is tight
    constexpr int L = 2 * R + NB;               // register window length
    constexpr int CR = rr_cring(R);             // centre ring rows: R + 2 NB
    constexpr int A_ITEMS = (NB * SWP + RR_THREADS - 1) / RR_THREADS;
    constexpr int B_ITEMS = (NB * 48 + RR_THREADS - 1) / RR_THREADS;
    constexpr int HT = NB * 3 * TW;             // one H-row tile
    float* stg = smem;                          // [NB][3][SWS] (SWP used)
    float* hrow = stg + NB * 3 * SWS;           // [2][NB][3][TW]  H-pass rows, then blur rows in place
    float* lut = hrow + 2 * HT;                 // [2][LUT_STRIDE]
    uint32_t* cring = reinterpret_cast<uint32_t*>(lut + 2 * LUT_STRIDE);   // uint8: [CR][TW] packed r|g<<8|b<<16.  half: [CR][3][TW] uint16 planes
    uint16_t* cring16 = reinterpret_cast<uint16_t*>(cring);
    float* cringf = reinterpret_cast<float*>(cring);                        // runtime-gate build: [CR][3][TW] graded floats
    uint32_t* rowtab = cring + rr_cring_floats(R, PIX, SF == 0xFFFFFFFFu);                              // [16][5] ring: scan gain bits, ny2 lo, ny2 hi, grain row offset, grain row weight of output row y at (y - y_begin) & 15
    int* ytab = reinterpret_cast<int*>(rowtab + 16 * 5);
    constexpr bool NLUT = (SF != 0xFFFFFFFFu) && PIX == 0;                 // gate-folded uint8 build: a1 from a 256-entry LDS table
    const int ytab_rows = (P.flags & CRTFX_F_PIXELATE) ? seg_rows + 2 * R : 0;      // [seg_rows + 2R]: source row of halo row (pixelate)
    float* nlut = reinterpret_cast<float*>(ytab + ytab_rows);              // [256] behind the pixelate row map (empty without pixelate); gate-folded uint8 builds
    float* glut = reinterpret_cast<float*>(ytab + ytab_rows);              // [3][256] grade table (runtime-gate build: it has no a1 table)

    // The four waves of a block have unequal roles (the V-pass has 192 columns for 256 threads, wave 0 carries the
    // prefetches).  Rotating the roles by the block's dispatch number spreads them over the SIMDs of a CU: measured
    // 4K 165.1 us per 2-frame launch without, 161.6 with the low bits (>>3: 161.5, >>5: 163.1, >>8: 171.9).
    const int wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int tid = (threadIdx.x + ((wg_lin & 3) << 6)) & (RR_THREADS - 1);
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int H = P.H, W = P.W;
    const int x0 = blockIdx.x * TW;
    const int y_begin = G.y0 + (int)blockIdx.y * seg_rows;
    const int y_end = min(G.y1, y_begin + seg_rows);
    if (y_begin >= G.y1) return;
    const uint32_t fl = P.flags;

    if ((fl & CRTFX_F_TRIAD) && (fl & CRTFX_F_TRIAD_LUT)) {
        for (int i = tid; i < LUT_N; i += RR_THREADS) { lut[i] = P.lut_g[i]; lut[LUT_STRIDE + i] = P.lut_inv[i]; }
    }
    // per-row table for this segment (the host only launches this kernel when no per-pixel
    // plane — triad_full, scan_plane, vig_full, noise_plane — and no in-kernel blend is in play)
    const bool row_scan = (fl & CRTFX_F_SCANLINES) != 0;
    const bool row_vig = (fl & CRTFX_F_VIGNETTE) != 0;
    const bool pixelate = (fl & CRTFX_F_PIXELATE) != 0;
    if (pixelate)
        for (int i = tid; i < y_end - y_begin + 2 * R; i += RR_THREADS) ytab[i] = P.ymap[min(max(y_begin - R + i, 0), H - 1)];
    // per-column constants of this lane
    const int xc = min(x0 + lane, W - 1);
    float cm0 = 1.0f, cm1 = 1.0f, cm2 = 1.0f;
    if (fl & CRTFX_F_TRIAD) { cm0 = P.triad_row[xc * 3]; cm1 = P.triad_row[xc * 3 + 1]; cm2 = P.triad_row[xc * 3 + 2]; }
    const double cnx2 = row_vig ? P.vig_nx2[xc] : 0.0;
    // Runtime-gate build only: a per-pixel scanline plane (slanted / shaped scanlines, ref:308-328) and the
    // bilinear upsample of a coarse grain plane (grain_size > 1, ref:637-642).  The gate-folded builds keep
    // neither (their launches never carry them: lean_ok / launch_rr_group).
    constexpr bool RTB = (SF == 0xFFFFFFFFu);
    const bool plane_scan = RTB && row_scan && F.scan_plane != nullptr;
    const bool coarse_grain = RTB && (fl & CRTFX_F_NOISE) && P.grain > 1;
    int cgxo = 0; float cgxa = 0.0f;
    if (coarse_grain) { cgxo = P.gx_ofs[xc]; cgxa = P.gx_a[xc]; }
    float pf_sp[2] = {1.0f, 1.0f}, sp_next[2] = {1.0f, 1.0f}, sp_c2[2] = {1.0f, 1.0f};   // plane gains of this thread's two C2 pixels: in flight, parked, in use
    int pf_gyo = 0; float pf_gya = 0.0f;

    // the Gaussian taps are symmetric (taps[k] == taps[2R-k] bit for bit: tables.gaussian_taps mirrors them),
    // so only R+1 of them are ever read: 10 SGPRs instead of 19 live through both blur phases
    const float* taps = P.taps;
#define TAP(k) taps[(k) <= R ? (k) : 2 * R - (k)]
    // V-pass register window as L / 2 VGPR pairs (2R + NB is even): element i = win2[i >> 1], half i & 1.  One window element
    // feeds two neighbouring output rows with two neighbouring taps = one v_pk_fma_f32 (3.4 cycles against 2 x 2.4 for two
    // v_fmac_f32 with an SGPR tap, profiles/r02_valu_cost.txt); every output still takes its taps top to bottom, fused.
    f32x2 win2[L / 2];
#pragma unroll
    for (int i = 0; i < L / 2; ++i) win2[i] = f32x2{0.0f, 0.0f};
    unsigned long long tpv[R + 1];                   // aligned SGPR pairs (tap[2m], tap[2m+1]); see k_phosphor_cc
#pragma unroll
    for (int m = 0; m <= R; ++m)
        tpv[m] = (unsigned long long)__float_as_uint(taps[2 * m]) | ((unsigned long long)(2 * m + 1 <= 2 * R ? __float_as_uint(taps[2 * m + 1]) : 0u) << 32);
    auto v_pass = [&](float* hcol) {                 // append the tile's NB rows, write the NB blurred rows in their place
#pragma unroll
        for (int j = 0; j < NB; ++j) win2[(2 * R + j) >> 1][j & 1] = hcol[j * 3 * TW];
        f32x2 acc[NB / 2];
#pragma unroll
        for (int jp = 0; jp < NB / 2; ++jp) acc[jp] = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int jp = 0; jp < NB / 2; ++jp) {
                const int t = i - 2 * jp;
                if (t == 0) acc[jp].x = fmaf(win2[i >> 1][i & 1], taps[0], acc[jp].x);
                else if (t >= 1 && t <= 2 * R) pk_fma_bcast(acc[jp], win2[i >> 1], (i & 1) != 0, (t & 1) ? tpv[(t - 1) / 2] : tpv[(2 * R - t) / 2], (t & 1) != 0);
                else if (t == 2 * R + 1) acc[jp].y = fmaf(win2[i >> 1][i & 1], taps[0], acc[jp].y);       // tap[2R] == tap[0]
            }
#pragma unroll
        for (int jp = 0; jp < NB / 2; ++jp) { hcol[(2 * jp) * 3 * TW] = acc[jp].x; hcol[(2 * jp + 1) * 3 * TW] = acc[jp].y; }
    };
    const int hcol_off = min(wave, 2) * TW + lane;   // this thread's column in its channel plane
#ifdef CRTFX_STAMP
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last) :: "memory");
#endif

    // A-phase item u of this thread: staging row j = it / SWP, column i = it % SWP (block-invariant).
    // Its source column (BORDER_REPLICATE clamp, then the pixelate map) is resolved once here so
    // that the loads issued inside the loop depend on no other vector-memory load: a dependent
    // index load in fetch would put an s_waitcnt vmcnt(0) in front of every item's byte loads.
    uint32_t offr[A_ITEMS], offg[A_ITEMS], offb[A_ITEMS];     // element offsets of this item's R, G, B inside a frame row
    // Items past the end of the NB x SWP tile (the last round is partial) redo the tile's last item: same loads, same
    // values, same LDS addresses — so the A phase and its prefetch need no per-item branch and stay one basic block.
#define A_ITEM(u) min(tid + (u) * RR_THREADS, NB * SWP - 1)
#pragma unroll
    for (int u = 0; u < A_ITEMS; ++u) {
        const int it = A_ITEM(u);
        const int i = it - (it / SWP) * SWP;
        int x = min(max(x0 - pad + i, 0), W - 1);
        if (pixelate) x = P.xmap[x];
        int xr = x, xb = x;
        if (P.ab != 0) { xr = wrap(x - P.ab, W); xb = wrap(x + P.ab, W); }      // ref:573-575
        offr[u] = (uint32_t)xr * 3u; offg[u] = (uint32_t)x * 3u + 1u; offb[u] = (uint32_t)xb * 3u + 2u;
    }
    // runtime-gate build: text overlay blended after the grade (ref:588-598), i.e. before the bloom sees the image.
    // The overlay pixel of a staged halo position is the one at its clamped (BORDER_REPLICATE) frame position — the
    // pixelate maps do not apply to it (a3 comes before the overlay).
    const bool ovl_before = RTB && F.overlay_before != nullptr;
    if constexpr (NLUT) { if (tid < 256) nlut[tid] = norm_u8((uint32_t)tid); }      // the same values norm_u8 computes, by construction
    const bool use_glut = RTB && PIX == 0 && P.grade_lut != nullptr;
    if (use_glut)
        for (int i = tid; i < 768; i += RR_THREADS) glut[i] = P.grade_lut[i];
    uint32_t ovx[A_ITEMS], ovpx[A_ITEMS];
#pragma unroll
    for (int u = 0; u < A_ITEMS; ++u) {
        const int it = A_ITEM(u);
        ovx[u] = (uint32_t)min(max(x0 - pad + (it - (it / SWP) * SWP), 0), W - 1);
        ovpx[u] = 0u;
    }
    __syncthreads();                                // ytab / rowtab / lut visible
    RawRGB raw[A_ITEMS];
    float pf_scan = 1.0f;                                  // per-row constants of output row hb - R + tid (threads < NB),
    double pf_ny2 = 0.0;                                   // requested one iteration ahead like the pixel bytes
    const uint32_t row_bytes = (uint32_t)W * 3u;           // elements per frame row
    auto prefetch = [&](int hb) {
        const int nrows = min(NB, y_end + R - hb);
        {
            const int yr = hb - R + tid;
            if (tid < NB && yr >= y_begin && yr < y_end) {
                if (row_scan && !plane_scan) pf_scan = F.scan_row[yr];
                if (row_vig) pf_ny2 = P.vig_ny2[yr];
                if (coarse_grain) { pf_gyo = P.gy_ofs[yr]; pf_gya = P.gy_a[yr]; }
            }
        }
        if (plane_scan) {                 // the two pixels this thread finishes in C2 of block hb: rows hb - R + wave (+ 4), column lane
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int yr = hb - R + wave + 4 * k;
                if (yr >= y_begin && yr < y_end) pf_sp[k] = F.scan_plane[(size_t)yr * W + xc];
            }
        }
#pragma unroll
        for (int u = 0; u < A_ITEMS; ++u) {
            const int it = A_ITEM(u);
            const int j = it / SWP;
            {   // rows past the end of a short last block (j >= nrows) are fetched too: clamped to the frame, never consumed
                const int y = pixelate ? ytab[min(hb + j - (y_begin - R), y_end - y_begin + 2 * R - 1)] : min(max(hb + j, 0), H - 1);   // BORDER_REPLICATE
                const uint32_t ro = (uint32_t)__umul24((uint32_t)y, row_bytes);   // y, row_bytes < 2^24 and the product < 2^32 for any frame the ctx accepts (v_mul_u32_u24: full rate, v_mul_lo_u32 is quarter rate)
                raw[u] = load_raw(PIX, F.in, ro + offr[u], ro + offg[u], ro + offb[u]);
                if (ovl_before)
                    ovpx[u] = reinterpret_cast<const uint32_t*>(F.overlay_before)[(uint32_t)min(max(hb + j, 0), H - 1) * (uint32_t)W + ovx[u]];
            }
        }
    };
    // centre-ring row of the first row of the block being graded (A) / of the block being finished (C2): both advance
    // by NB per iteration modulo CR (wave-uniform; replaces a division by CR per item and per row)
    int crow0 = 0;                 // (hb - (y_begin - R)) % CR
    int c2row0 = NB;               // (hb - NB - y_begin) % CR = CR - R - NB at the first iteration: image row of output row hb - NB - R
    // C2 of the block whose first H-row is hbp: output rows [hbp - R, hbp - R + NB) from tile `ht`
    // the per-pixel inputs of C2 for row j of the block (output row y): parked centre pixel + bloom, masks of the pixel.
    // Lanes past the right edge hold the replicated edge pixel (A parks all 64 centre columns): they run the same
    // arithmetic and only their stores are masked, so there is no branch in here.
    auto c2_inputs = [&](int j, int y, const float* ht, PixMasks& M, float& r, float& g, float& b) {
        int cr = c2row0 + j;                       // (y - (y_begin - R)) % CR without the division
        cr = cr >= CR ? cr - CR : cr;
        uint32_t s0 = 0, s1 = 0, s2 = 0;
        if constexpr (RTB) { const float* cp = cringf + cr * 3 * TW + lane; r = cp[0]; g = cp[TW]; b = cp[2 * TW]; }
        else if constexpr (PIX) { const uint16_t* cp = cring16 + cr * 3 * TW + lane; s0 = cp[0]; s1 = cp[TW]; s2 = cp[2 * TW]; }
        else { const uint32_t pk = cring[cr * TW + lane]; s0 = pk & 255u; s1 = (pk >> 8) & 255u; s2 = (pk >> 16) & 255u; }
        const uint32_t* rt = rowtab + ((y - y_begin) & 15) * 5;
        M.sl = plane_scan ? (j >= 4 ? sp_c2[1] : sp_c2[0]) : __uint_as_float(rt[0]);
        if (coarse_grain) {        // ref:637-642: horizontal lerp of the two coarse rows, then the vertical one
            const int sx = cgxo, sy = (int)rt[3];
            const int sx1 = min(sx + 1, P.gw - 1), sy1 = min(sy + 1, P.gh - 1);
            const float a1 = cgxa, a0 = 1.0f - a1, b1 = __uint_as_float(rt[4]), b0 = 1.0f - b1;
            const float n00 = grain_normal(F.key0, F.key1, (uint32_t)sy * P.gw + sx), n01 = grain_normal(F.key0, F.key1, (uint32_t)sy * P.gw + sx1);
            const float n10 = grain_normal(F.key0, F.key1, (uint32_t)sy1 * P.gw + sx), n11 = grain_normal(F.key0, F.key1, (uint32_t)sy1 * P.gw + sx1);
            M.z = (n00 * a0 + n01 * a1) * b0 + (n10 * a0 + n11 * a1) * b1;
            M.has_z = 1;
        }
        if (fl & CRTFX_F_VIGNETTE) M.vig = vignette_gain(P, cnx2, __hiloint2double((int)rt[2], (int)rt[1]));
        if constexpr (!RTB) {
            if constexpr (NLUT) { r = nlut[s0]; g = nlut[s1]; b = nlut[s2]; }
            else { r = norm_px(PIX, s0); g = norm_px(PIX, s1); b = norm_px(PIX, s2); }
            grade(P, r, g, b);
        }
        r = clip01(r + P.bloom_strength * ht[(j * 3 + 0) * TW + lane]);   // ref:611
        g = clip01(g + P.bloom_strength * ht[(j * 3 + 1) * TW + lane]);
        b = clip01(b + P.bloom_strength * ht[(j * 3 + 2) * TW + lane]);
    };
    // C2 of the block whose first H-row is hbp: output rows [hbp - R, hbp - R + NB) from tile `ht`; wave w handles rows
    // w and w + 4, unrolled: the kernel is latency-bound rather than issue-bound and the LDS / LUT chains of the two rows
    // interleave (4K 164 -> 155 us per 2-frame launch; merging them into one straight-line block by hand adds nothing).
    auto phase_c2 = [&](int hbp, const float* ht) {
        const int x = x0 + lane;
        const bool xin = x < W;
#pragma unroll
        for (int j = wave; j < NB; j += 4) {
            const int y = hbp - R + j;
            if (y >= y_begin && y < y_end) {                  // wave-uniform
                float r = 0, g = 0, b = 0;
                PixMasks M{cm0, cm1, cm2, 1.0f, 1.0, 0.0f, 0};
                c2_inputs(j, y, ht, M, r, g, b);
                emit_pixel<true, RTB>(P, F, O, y, x0, lane, xin, M, r, g, b, lut, lut + LUT_STRIDE);
            }
        }
    };

    // Phase pairing per block n (first H-row hb, tile t = n & 1):
    //     { C1(n-1), A(n) }  barrier  { C2(n-1), B(n) }  barrier
    // The stores of C2(n-1) then have the whole of B(n) + C1(n) + A(n+1) to retire before the next
    // s_waitcnt vmcnt (the prefetched bytes of A(n+1)): vmcnt counts loads and stores in one
    // in-order queue, so a wait placed right behind the stores would expose their latency.
    prefetch(y_begin - R);
    int t = 0;
    for (int hb = y_begin - R; hb < y_end + R; hb += NB, t ^= 1, crow0 = crow0 + NB >= CR ? crow0 + NB - CR : crow0 + NB,
                                                   c2row0 = c2row0 + NB >= CR ? c2row0 + NB - CR : c2row0 + NB) {
        const int nrows = min(NB, y_end + R - hb);
        float* ht = hrow + t * HT;
        // ---- C1(n-1): vertical pass on the register window; row j = blur of output row hb-NB-R+j ----
        if (hb > y_begin - R && wave < 3) {
            CC_PRIO(RR_P_VH);
            v_pass(hrow + (t ^ 1) * HT + hcol_off);
#pragma unroll
            for (int i = 0; i < R; ++i) win2[i] = win2[i + NB / 2];
            CC_PRIO(0);
        }
        STAMP(4);
        // ---- A(n): grade the prefetched halo rows [hb, hb+nrows) into the staging tile ----------
        {
            const int yr = hb - R + tid;                     // output row whose constants arrived with this block's bytes
            if (tid < NB && yr >= y_begin && yr < y_end) {
                uint32_t* rt = rowtab + ((yr - y_begin) & 15) * 5;
                rt[0] = __float_as_uint(pf_scan); rt[1] = (uint32_t)__double2loint(pf_ny2); rt[2] = (uint32_t)__double2hiint(pf_ny2);
                if (coarse_grain) { rt[3] = (uint32_t)pf_gyo; rt[4] = __float_as_uint(pf_gya); }
            }
            if (plane_scan) {      // C2 of the previous block runs later in this iteration with sp_c2; this block's values wait in sp_next
                sp_c2[0] = sp_next[0]; sp_c2[1] = sp_next[1];
                sp_next[0] = pf_sp[0]; sp_next[1] = pf_sp[1];
            }
        }
#pragma unroll
        for (int u = 0; u < A_ITEMS; ++u) {
            const int it = A_ITEM(u);
            const int j = it / SWP, i = it - j * SWP;
            {   // no per-item branch: rows >= nrows of a short last block are graded too and never read
                float r, g, b;
                if (use_glut) { r = glut[raw[u].r]; g = glut[256 + raw[u].g]; b = glut[512 + raw[u].b]; }
                else if constexpr (NLUT) { r = nlut[raw[u].r]; g = nlut[raw[u].g]; b = nlut[raw[u].b]; grade(P, r, g, b); }
                else { r = norm_px(PIX, raw[u].r); g = norm_px(PIX, raw[u].g); b = norm_px(PIX, raw[u].b); grade(P, r, g, b); }
                if (ovl_before) overlay_blend_px<float>(ovpx[u], r, g, b);
                if (i >= pad && i < pad + TW) {     // centre column: park the pixel for C2 (graded floats, or the packed samples)
                    int cr = crow0 + j;                        // (hb + j - (y_begin - R)) % CR without the division
                    cr = cr >= CR ? cr - CR : cr;
                    if constexpr (RTB) { float* cp = cringf + cr * 3 * TW + (i - pad); cp[0] = r; cp[TW] = g; cp[2 * TW] = b; }
                    else if constexpr (PIX) { uint16_t* cp = cring16 + cr * 3 * TW + (i - pad); cp[0] = (uint16_t)raw[u].r; cp[TW] = (uint16_t)raw[u].g; cp[2 * TW] = (uint16_t)raw[u].b; }
                    else cring[cr * TW + (i - pad)] = raw[u].r | (raw[u].g << 8) | (raw[u].b << 16);
                }
                float* s = stg + (j * 3) * SWS + i;
                s[0] = bloom_src(P, r); s[SWS] = bloom_src(P, g); s[2 * SWS] = bloom_src(P, b);
            }
        }
        if (hb + NB < y_end + R) prefetch(hb + NB);     // in flight across C2 / B / C1
        STAMP(0);
        __syncthreads();
        STAMP(1);
        // ---- C2(n-1): combine + masks + store -----------------------------------------------------
        CC_PRIO(RR_P_C2);
        if (hb > y_begin - R) phase_c2(hb - NB, hrow + (t ^ 1) * HT);
        STAMP(6);
        // ---- B(n): horizontal pass -> tile t ----------------------------------------------------------
        CC_PRIO(RR_P_VH);
#pragma unroll
        for (int u = 0; u < B_ITEMS; ++u) {
            const int it = tid + u * RR_THREADS;
            const int j = it / 48, rem = it - j * 48;
            // rows >= nrows of a short last block are filtered too (stale staging rows in, never read out): the only
            // branch left is the wave-uniform one that ends the partial last round (NB * 48 items over RR_THREADS)
            if (it < NB * 48) {
                const int c = rem >> 4, gq = rem & 15;
                // volatile: keeps each 16-byte read whole (ds_read_b128); a plain float4 load is scalarised and
                // re-merged into ds_read2_b32 pairs
                const lds_cv_f32x4* srow = (const lds_cv_f32x4*)smem4 + ((j * 3 + c) * (SWS / 4) + gq);
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                constexpr int off = pad - R;
#pragma unroll
                for (int qq = 0; qq < (2 * pad + 4) / 4; ++qq) {
                    const f32x4 v = srow[qq];
                    const float ve[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int tt = 4 * qq + e - i - off;
                            if (tt >= 0 && tt <= 2 * R) acc[i] = fmaf(ve[e], TAP(tt), acc[i]);
                        }
                }
                smem4[(NB * 3 * SWS + t * HT) / 4 + (j * 3 + c) * (TW / 4) + gq] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
        }
        CC_PRIO(0);
        STAMP(2);
        __syncthreads();
        STAMP(3);
    }
    // ---- drain: C1 and C2 of the last block --------------------------------------------------------
    {
        const int hb_last = y_begin - R + ((y_end + R - (y_begin - R) - 1) / NB) * NB;
        float* htl = hrow + (t ^ 1) * HT;
        if (wave < 3) {
            v_pass(htl + hcol_off);
        }
        __syncthreads();
        if (plane_scan) { sp_c2[0] = sp_next[0]; sp_c2[1] = sp_next[1]; }
        phase_c2(hb_last, htl);
    }
#ifdef CRTFX_STAMP
    if (O.dbg && lane == 0) {
        unsigned long long* d = O.dbg + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) d[i] = stamp_sum[i];
    }
#endif
}
#undef TAP

// ---------------------------------------------------------------------------------------
// k_phosphor_cc — the full-chain gate set (SF_FULL: Gaussian bloom, triad LUTs, row scanlines, analytic vignette,
// grain) of k_phosphor_rr for launches that park a float32 pre-warp image (warp and / or persistence behind it).
// Same arithmetic, expression for expression (tests/test_parity_gpu.py::test_kernel_variants_agree holds the builds
// to identical bits); what changes is who does what, and that each phase of a wave is ONE basic block.  The PMC passes on
// k_phosphor_rr showed a wave issuing one instruction per ~14 cycles: every `if` around a row or an item ends in an
// s_waitcnt, so LDS round trips were paid one after the other, wave 3 idled through the V pass, and the blur went back
// to LDS to be re-read by another thread.
//
//   CONSUMER waves 0-2: thread f owns float f of the strip's 192-float interleaved RGB row segment (pixel f / 3,
//   channel f % 3) in the V pass AND in the pointwise tail, eight rows at a time, everything in registers between them:
//     phase 1   centre samples of block n-1 (LDS -> a1 table, issued first) | C1(n-1) V pass on the register window |
//               A(n): its share of the prefetched halo bytes -> staging tile | prefetch of block n+1
//     phase 2   C2(n-1): img + s*blur, triad LUT pair, scanline, * vignette, + grain for the eight rows stage by stage,
//               eight branch-free stores (256 contiguous bytes per wave; rows / lanes outside the frame go to a trash
//               line, so the stores sit in the same basic block and the compiler counts them exactly in vmcnt) | B(n)
//   HELPER wave 3: lane = pixel column:
//     phase 1   V(n-1): float64 vignette gain of the block's 8 x 64 pixels -> LDS | its share of A(n) | prefetch
//     phase 2   N(n): grain N(0,1) * scale of the NEXT block's 8 x 64 pixels -> LDS (double-buffered) | its share of B(n)
//   two barriers per eight rows, as before.  The two roles run separate copies of the loop (same trip count, same
//   barriers): no role test inside a phase.
// ---------------------------------------------------------------------------------------
// LDS access by BYTE OFFSET from the start of the workgroup's LDS (k_phosphor_cc has no static LDS, so its dynamic
// block starts at 0 — checked once at kernel entry).  hipcc forms the address of lut[idx] as v_lshl_add_u32(idx, 2, 0):
// a 3.4-cycle VOP3 where a 1.9-cycle v_lshlrev_b32 plus the instruction's immediate offset does (33 of them per trip).
typedef __attribute__((address_space(3))) float lds_f32_t;
typedef __attribute__((address_space(3))) double lds_f64_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;
typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
#define LDS_AT(T, off) (*(T*)(uintptr_t)(uint32_t)(off))

// staging plane stride: >= the staged width and == 4 (mod 8) dwords, so that the two (row, channel) planes one 16-lane
// ds_read_b128 group covers in the H pass (8 lanes each, 32 bytes apart) land on disjoint banks
__host__ __device__ constexpr int cc_sws(int R) { return ((rr_swp(R) + 3) & ~7) + 4; }
constexpr int CC_HROW = 3 * TW + 4;       // H-row tile row stride in floats: == 4 (mod 32), the H pass's column-strided stores stay 2-way
__host__ __device__ constexpr int cc_cring_words(int R, int pix) { return pix ? (rr_cring(R) * TW * 3 + 1) / 2 : rr_cring(R) * TW; }   // half: [CR][TW][3] uint16; uint8: [CR][TW] packed r | g<<8 | b<<16
// LDS words: staging, ONE H-row tile, LUTs, centre ring, a1 table (uint8), vignette tile (f64), two grain tiles (f32), row table
__host__ __device__ constexpr int cc_lds_words(int R, int pix) {
    return NB * 3 * cc_sws(R) + NB * CC_HROW + 2 * LUT_STRIDE + cc_cring_words(R, pix) + (pix ? 0 : 256) + NB * TW * 2 + 2 * NB * TW + 16 * 4;
}
__host__ __device__ constexpr int cc_min_waves(int R) { return R <= 12 ? 4 : (R <= 20 ? 3 : 2); }
#ifndef CC_A3
#define CC_A3(na) ((na) / 5)                // A-phase wave-items (64 staged pixels each) of the helper wave; waves 0-2 share the rest
#endif

template <int RT, int PIX>
__global__ __launch_bounds__(RR_THREADS, cc_min_waves(RT)) void k_phosphor_cc(KParams Pin, KGroup G, int seg_rows) {
    const KFrame F = G.f[blockIdx.z];
    const KOut O = G.o[blockIdx.z];
    KParams P = Pin;
    P.flags = SF_FULL;
    P.pix = PIX;
    extern __shared__ float4 smem4[];
    float* smem = reinterpret_cast<float*>(smem4);
    constexpr int R = RT, K = 2 * R + 1;
    constexpr int pad = rr_pad(R);
    constexpr int SWP = rr_swp(R);
    constexpr int SWS = cc_sws(R);
    constexpr int L = 2 * R + NB;
    constexpr int CR = rr_cring(R);
    constexpr int NA = (NB * SWP + 63) / 64;             // A-phase wave-items
    constexpr int A3 = CC_A3(NA);                        // ... of the helper wave (the last A3 items)
    constexpr int AO = (NA - A3 + 2) / 3;                // ... of each consumer wave (items wave, wave + 3, ...)
    constexpr int HT = NB * CC_HROW;
    constexpr bool NLUT = PIX == 0;       // a1 from the LDS table; as arithmetic (v_cvt_f32_ubyte + corrected reciprocal) it is the same speed: 135.5 vs 136.1 us
    // LDS map, byte offsets from 0 (LDS_AT): every hot access is `constant + per-lane offset`, so that the constant rides in the
    // instruction's immediate and the per-lane part is one shift or add
    constexpr uint32_t STG_B = 0;                                            // [NB][3][SWS] float      staging tile
    constexpr uint32_t HROW_B = STG_B + NB * 3 * SWS * 4;                    // [NB][CC_HROW] float     H rows, interleaved like the image row (x, channel)
    constexpr uint32_t LUT_B = HROW_B + HT * 4;                              // [2][LUT_STRIDE] float   triad LUT pair
    constexpr uint32_t CRING_B = LUT_B + 2 * LUT_STRIDE * 4;                 // uint8: [CR][TW] packed dwords; half: [CR][TW][3] uint16   parked centre samples
    constexpr uint32_t NLUT_B = CRING_B + cc_cring_words(R, PIX) * 4;        // [256] float             u / 255.0 (uint8 frames)
    constexpr uint32_t GVIG_B = NLUT_B + (PIX == 0 ? 256 * 4 : 0);           // [NB][TW] double         vignette gain tile
    constexpr uint32_t GN_B = GVIG_B + NB * TW * 8;                          // [2][NB][TW] float       grain tiles
    constexpr uint32_t ROWTAB_B = GN_B + 2 * NB * TW * 4;                    // [16][4] uint32          scan gain bits, ny2 lo, ny2 hi, -
    static_assert(ROWTAB_B + 16 * 4 * 4 == (uint32_t)cc_lds_words(R, PIX) * 4, "LDS map and cc_lds_words disagree");
    float* stg = smem;
    float* hrow = smem + HROW_B / 4;
    float* lut = smem + LUT_B / 4;
    uint16_t* cring16 = reinterpret_cast<uint16_t*>(smem + CRING_B / 4);
    float* nlut = smem + NLUT_B / 4;
    double* gvig = reinterpret_cast<double*>(smem + GVIG_B / 4);
    uint32_t* rowtab = reinterpret_cast<uint32_t*>(smem + ROWTAB_B / 4);
    float* gn = smem + GN_B / 4;
    if ((uint32_t)(uintptr_t)(lds_f32_t*)smem != 0u) __builtin_trap();      // LDS_AT assumes the dynamic block starts at 0

    const int wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int tid = (threadIdx.x + ((wg_lin & 3) << 6)) & (RR_THREADS - 1);          // roles rotate over the SIMDs with the dispatch number
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int H = P.H, W = P.W;
    const int x0 = blockIdx.x * TW;
    const int y_begin = G.y0 + (int)blockIdx.y * seg_rows;
    const int y_end = min(G.y1, y_begin + seg_rows);
    if (y_begin >= G.y1) return;

    for (int i = tid; i < LUT_N; i += RR_THREADS) { lut[i] = P.lut_g[i]; lut[LUT_STRIDE + i] = P.lut_inv[i]; }
    if constexpr (NLUT) { if (tid < 256) nlut[tid] = norm_u8((uint32_t)tid); }
    const float* taps = P.taps;
#define TAP(k) taps[(k) <= R ? (k) : 2 * R - (k)]
    // the taps as R + 1 aligned SGPR pairs (tap[2m], tap[2m+1]); the pair (tap[t], tap[t-1]) a packed FMA wants is pair
    // (t-1)/2 swapped when t is odd and, the kernel being symmetric (tap[k] == tap[2R-k] bit for bit), pair (2R-t)/2 as
    // it stands when t is even
    unsigned long long tp[R + 1];
#pragma unroll
    for (int m = 0; m <= R; ++m)
        tp[m] = (unsigned long long)__float_as_uint(taps[2 * m]) | ((unsigned long long)(2 * m + 1 <= 2 * R ? __float_as_uint(taps[2 * m + 1]) : 0u) << 32);
    // acc.x += w * tap[t], acc.y += w * tap[t-1]   (1 <= t <= 2R)
#define PK_TAPS(acc, wpair, whigh, t) pk_fma_bcast(acc, wpair, whigh, ((t) & 1) ? tp[((t) - 1) / 2] : tp[(2 * R - (t)) / 2], ((t) & 1) != 0)
    const uint32_t row_elems = (uint32_t)W * 3u;
    const int n_iter = (y_end + R - (y_begin - R) + NB - 1) / NB;                    // loop trips (same for both roles)
#ifdef CRTFX_STAMP
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last) :: "memory");
#endif

    // ---- pieces shared by the two roles (instantiated once per role: item counts are compile-time there) -------------
    // source element offsets of A-phase wave-item q for this lane (block-invariant)
    auto a_offsets = [&](int q, uint32_t& o_r, uint32_t& o_g, uint32_t& o_b) {
        const int it = min((q << 6) + lane, NB * SWP - 1);     // lanes past the tile's last item redo it (same loads, same LDS stores)
        const int i = it - (it / SWP) * SWP;
        const int x = min(max(x0 - pad + i, 0), W - 1);
        int xr = x, xb = x;
        if (P.ab != 0) { xr = wrap(x - P.ab, W); xb = wrap(x + P.ab, W); }      // ref:573-575
        o_r = (uint32_t)xr * 3u; o_g = (uint32_t)x * 3u + 1u; o_b = (uint32_t)xb * 3u + 2u;
    };
    auto a_load = [&](int q, int hb, uint32_t o_r, uint32_t o_g, uint32_t o_b) -> RawRGB {
        const int it = min((q << 6) + lane, NB * SWP - 1);
        const int y = min(max(hb + it / SWP, 0), H - 1);                          // BORDER_REPLICATE
        const uint32_t ro = (uint32_t)__umul24((uint32_t)y, row_elems);
        return load_raw(PIX, F.in, ro + o_r, ro + o_g, ro + o_b);
    };
    // a1 of one staged pixel (the table read / the arithmetic), then its stores: callers run the lookups of ALL their items before
    // the first store, so that the LDS round trips overlap instead of queueing item after item
    auto a_lookup = [&](RawRGB v, float (&o)[3]) {
        if constexpr (NLUT) { o[0] = LDS_AT(lds_f32_t, NLUT_B + (v.r << 2)); o[1] = LDS_AT(lds_f32_t, NLUT_B + (v.g << 2)); o[2] = LDS_AT(lds_f32_t, NLUT_B + (v.b << 2)); }
        else { o[0] = norm_px(PIX, v.r); o[1] = norm_px(PIX, v.g); o[2] = norm_px(PIX, v.b); }
    };
    auto a_write = [&](int q, int crow0, RawRGB v, const float (&o)[3]) {
        const int it = min((q << 6) + lane, NB * SWP - 1);
        const int j = it / SWP, i = it - j * SWP;
        if (i >= pad && i < pad + TW) {
            int cr = crow0 + j;
            cr = cr >= CR ? cr - CR : cr;
            if constexpr (PIX) { uint16_t* cp = cring16 + (cr * TW + (i - pad)) * 3; cp[0] = (uint16_t)v.r; cp[1] = (uint16_t)v.g; cp[2] = (uint16_t)v.b; }
            else LDS_AT(lds_u32_t, CRING_B + (uint32_t)((cr * TW + (i - pad)) * 4)) = v.r | (v.g << 8) | (v.b << 16);
        }
        float* sp = stg + (j * 3) * SWS + i;
        sp[0] = o[0]; sp[SWS] = o[1]; sp[2 * SWS] = o[2];
    };
    // H pass of the staging tile by a consumer wave: NB x 3 (row, channel) planes, 8 lanes per plane, 8 adjacent outputs per
    // lane.  Lane -> (plane, octet) goes through the hardware's 16-lane ds_read_b128 groups ({0-3,12-15,20-27},
    // {4-11,16-19,28-31}, ... of each 32): a group reads two consecutive planes, which SWS == 4 (mod 8) keeps on disjoint
    // banks.  Per output the taps run left to right, fused (the oracle's RowFilter order).
    auto h_pass = [&](int w) {
        const int l5 = lane & 31;
        const int hg = ((lane >> 5) << 1) | ((l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28 ? 1 : 0);      // 16-lane group 0..3
        const int pos = (hg & 1) ? (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16)) : (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12));   // 0..15 inside it
        const int plane = 8 * w + 2 * hg + (pos >> 3);          // j * 3 + c
        const int g8 = pos & 7;
        const int j = plane / 3, c = plane - 3 * j;
        const lds_cv_f32x4* srow = (const lds_cv_f32x4*)smem4 + (plane * (SWS / 4) + 2 * g8);
        f32x2 acc2[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};      // outputs (0,1) (2,3) (4,5) (6,7)
        constexpr int off = pad - R;
        constexpr int NQ = (2 * pad + 8) / 4;
        f32x4 vq[NQ];
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) vq[qq] = srow[qq];       // all reads in flight before the first tap (the FMAs then wait quad by quad)
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
            const f32x4 vv = vq[qq];
            const f32x2 vp[2] = {{vv[0], vv[1]}, {vv[2], vv[3]}};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const int t = 4 * qq + e - 2 * pp - off;      // tap of the pair's first output; its second takes t - 1
                    if (t == 0) acc2[pp].x = fmaf(vp[e >> 1][e & 1], taps[0], acc2[pp].x);
                    else if (t >= 1 && t <= 2 * R) PK_TAPS(acc2[pp], vp[e >> 1], (e & 1) != 0, t);
                    else if (t == 2 * R + 1) acc2[pp].y = fmaf(vp[e >> 1][e & 1], taps[0], acc2[pp].y);      // tap[2R] == tap[0]
                }
        }
        float* hp = hrow + j * CC_HROW + 8 * g8 * 3 + c;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) { hp[6 * pp] = acc2[pp].x; hp[6 * pp + 3] = acc2[pp].y; }
    };

    if (wave < 3) {
        // =============================== CONSUMER: waves 0-2 ================================================================
        const int f = wave * 64 + lane;
        const int fcol = f / 3, fch = f - 3 * fcol;
        const bool fin = x0 + fcol < W;
        const float cm = P.triad_row[min(x0 + fcol, W - 1) * 3 + fch];           // a7 mask of this float
        const uint32_t cpl = (uint32_t)(fcol * 4 + fch);                         // this float's byte in a centre-ring row (packed pixel fcol, byte fch)
        const uint32_t gcol8 = (uint32_t)fcol * 8u, gcol4 = (uint32_t)fcol * 4u;  // its pixel in the vignette / grain tiles
        // The pre-warp image is written through a buffer resource (base, H * W * 12 bytes): one SGPR descriptor + a 32-bit
        // byte offset per store, no 64-bit address arithmetic, and an offset past the image is DROPPED by the hardware's
        // range check — so rows outside the segment (offset | all-ones, a scalar mask) and lanes right of the frame
        // (offset pinned out of range) cost no branch: the eight stores sit in C2's basic block and the compiler counts
        // them exactly in every s_waitcnt vmcnt behind them.
        const __amdgpu_buffer_rsrc_t pre_rsrc = __builtin_amdgcn_make_buffer_rsrc(O.pre, 0, (int)((uint32_t)H * (uint32_t)W * 12u), 0x00020000);
        const uint32_t row_b = fin ? (uint32_t)W * 12u : 0u;                     // bytes per pre-warp image row (this lane's stride)
        // V-pass register window as L / 2 VGPR pairs (2R + NB is even): element i = win2[i >> 1], half i & 1
        f32x2 win2[L / 2];
#pragma unroll
        for (int i = 0; i < L / 2; ++i) win2[i] = f32x2{0.0f, 0.0f};
        // C1: append the eight H rows of the tile, form output rows j (x) and j + 1 (y) of each pair from window elements
        // i = j .. j + 2R + 1 oldest first, shift the window down by NB
        auto v_pass = [&](float (&blur)[NB]) {
            const float* hcol = hrow + f;
#pragma unroll
            for (int j = 0; j < NB; ++j) win2[(2 * R + j) >> 1][j & 1] = hcol[j * CC_HROW];
            f32x2 acc[NB / 2];
#pragma unroll
            for (int jp = 0; jp < NB / 2; ++jp) acc[jp] = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < L; ++i)
#pragma unroll
                for (int jp = 0; jp < NB / 2; ++jp) {
                    const int t = i - 2 * jp;
                    if (t == 0) acc[jp].x = fmaf(win2[i >> 1][i & 1], taps[0], acc[jp].x);
                    else if (t >= 1 && t <= 2 * R) PK_TAPS(acc[jp], win2[i >> 1], (i & 1) != 0, t);
                    else if (t == 2 * R + 1) acc[jp].y = fmaf(win2[i >> 1][i & 1], taps[0], acc[jp].y);       // tap[2R] == tap[0]
                }
#pragma unroll
            for (int jp = 0; jp < NB / 2; ++jp) { blur[2 * jp] = acc[jp].x; blur[2 * jp + 1] = acc[jp].y; }
#pragma unroll
            for (int i = 0; i < R; ++i) win2[i] = win2[i + NB / 2];
        };
        uint32_t offr[AO], offg[AO], offb[AO];
        RawRGB raw[AO];
#pragma unroll
        for (int u = 0; u < AO; ++u) a_offsets(min(wave + 3 * u, NA - A3 - 1), offr[u], offg[u], offb[u]);
        __syncthreads();                                // LUTs / a1 table visible
#pragma unroll
        for (int u = 0; u < AO; ++u) raw[u] = a_load(min(wave + 3 * u, NA - A3 - 1), y_begin - R, offr[u], offg[u], offb[u]);
        // eight stores behind the first prefetch, as in every later trip: the loop is entered with the same count of vector
        // memory operations younger than the prefetched bytes as its back edge carries, so A's s_waitcnt vmcnt leaves
        // exactly the stores in flight
#pragma unroll
        for (int j = 0; j < NB; ++j) __builtin_amdgcn_raw_buffer_store_b32(0u, pre_rsrc, 0xFFFFFF00u - 16u * (uint32_t)j, 0, 0);      // out of range: dropped
        CC_PRIO(CC_P_A);
        int crow0 = 0, c2row0 = NB;
        int hb = y_begin - R;
        uint32_t off0 = fin ? (uint32_t)(y_begin - 2 * R - NB) * row_b + ((uint32_t)x0 * 3u + (uint32_t)f) * 4u : 0xFFFFFF00u;      // (row hb - NB - R, float f), modulo 2^32 while that row is < 0
        for (int n = 0; n < n_iter; ++n, hb += NB, off0 += (uint32_t)NB * row_b, crow0 = crow0 + NB >= CR ? crow0 + NB - CR : crow0 + NB,
                                        c2row0 = c2row0 + NB >= CR ? c2row0 + NB - CR : c2row0 + NB) {
            // ---- phase 1 ----
            float v[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {              // centre sample of output row hb - NB - R + j (a1; a2 is in the parked sample); garbage in trip 0
                int cr = c2row0 + j;
                cr = cr >= CR ? cr - CR : cr;
                if constexpr (PIX) v[j] = norm_px(PIX, (uint32_t)cring16[cr * 3 * TW + f]);
                else if constexpr (NLUT) v[j] = LDS_AT(lds_f32_t, NLUT_B + ((uint32_t)LDS_AT(lds_u8_t, CRING_B + (uint32_t)(cr * TW * 4) + cpl) << 2));
                else v[j] = norm_u8((uint32_t)LDS_AT(lds_u8_t, CRING_B + (uint32_t)(cr * TW * 4) + cpl));
            }
            float blur[NB];
            CC_PRIO(CC_P_VH);
            v_pass(blur);
            CC_PRIO(CC_P_A);
            STAMP(4);
            {
                float nv[AO][3];
#pragma unroll
                for (int u = 0; u < AO; ++u) a_lookup(raw[u], nv[u]);
#pragma unroll
                for (int u = 0; u < AO; ++u) a_write(min(wave + 3 * u, NA - A3 - 1), crow0, raw[u], nv[u]);
            }
#pragma unroll
            for (int u = 0; u < AO; ++u) raw[u] = a_load(min(wave + 3 * u, NA - A3 - 1), hb + NB, offr[u], offg[u], offb[u]);   // past the last block: clamped rows, never consumed
            STAMP(0);
            __syncthreads();
            STAMP(1);
            // ---- phase 2: C2 of block n-1 (output rows hb - NB - R + j), stage by stage over the eight rows ----
            CC_PRIO(CC_P_C2);
            const int yb = hb - NB - R;
            // the per-pixel tiles of the helper wave and the row gains first: they depend on nothing in here, and their
            // round trip then runs beside the two LUT gathers instead of behind them
            const uint32_t gt_b = GN_B + (uint32_t)(((n & 1) ^ 1) * NB * TW * 4) + gcol4;
            float sl[NB], gnv[NB];
            double gv[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                sl[j] = __uint_as_float(LDS_AT(lds_u32_t, ROWTAB_B + (uint32_t)(((yb + j - y_begin) & 15) * 16)));
                gv[j] = LDS_AT(lds_f64_t, GVIG_B + (uint32_t)(j * TW * 8) + gcol8);
                gnv[j] = LDS_AT(lds_f32_t, gt_b + (uint32_t)(j * TW * 4));
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) v[j] = clip01(v[j] + P.bloom_strength * blur[j]);          // ref:611
#pragma unroll
            for (int j = 0; j < NB; ++j) v[j] = LDS_AT(lds_f32_t, LUT_B + ((uint32_t)lut_index_unit(v[j]) << 2)) * cm;                   // ref:250-252
#pragma unroll
            for (int j = 0; j < NB; ++j) v[j] = LDS_AT(lds_f32_t, LUT_B + LUT_STRIDE * 4 + ((uint32_t)lut_index(v[j]) << 2));           // ref:261-262
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float r = clip01(v[j] * sl[j]);                                               // ref:617-624
                double d = (double)r * gv[j];                                                       // ref:626-628 (gain in [0,1]: no clip)
                d = clip01(d + (double)gnv[j]);                                                     // ref:646-647
                v[j] = (float)d;
            }
            {
                uint32_t boff = off0;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int y = yb + j;
                    const uint32_t oob = (y >= y_begin && y < y_end) ? 0u : 0xFFFFFFFFu;       // wave-uniform
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j]), pre_rsrc, boff | oob, 0, 0);
                    boff += row_b;
                }
            }
            STAMP(6);
            CC_PRIO(CC_P_VH);
            h_pass(wave);
            CC_PRIO(CC_P_A);
            STAMP(2);
            __syncthreads();
            STAMP(3);
        }
        // ---- drain: C1 and C2 of the last block ----
        {
            float v[NB], blur[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                int cr = c2row0 + j;
                cr = cr >= CR ? cr - CR : cr;
                if constexpr (PIX) v[j] = norm_px(PIX, (uint32_t)cring16[cr * 3 * TW + f]);
                else if constexpr (NLUT) v[j] = LDS_AT(lds_f32_t, NLUT_B + ((uint32_t)LDS_AT(lds_u8_t, CRING_B + (uint32_t)(cr * TW * 4) + cpl) << 2));
                else v[j] = norm_u8((uint32_t)LDS_AT(lds_u8_t, CRING_B + (uint32_t)(cr * TW * 4) + cpl));
            }
            v_pass(blur);
            __syncthreads();
            const float* gt = gn + ((n_iter & 1) ^ 1) * NB * TW;
            const int yb = hb - NB - R;
#pragma unroll
            for (int j = 0; j < NB; ++j) v[j] = clip01(v[j] + P.bloom_strength * blur[j]);
#pragma unroll
            for (int j = 0; j < NB; ++j) v[j] = lut[lut_index_unit(v[j])] * cm;
#pragma unroll
            for (int j = 0; j < NB; ++j) v[j] = lut[LUT_STRIDE + lut_index(v[j])];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int y = yb + j;
                const float sl = __uint_as_float(rowtab[((y - y_begin) & 15) * 4]);
                const float r = clip01(v[j] * sl);
                double d = (double)r * gvig[j * TW + fcol];
                d = clip01(d + (double)gt[j * TW + fcol]);
                if (y >= y_begin && y < y_end && fin) O.pre[((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u + (uint32_t)f] = (float)d;
            }
        }
    } else {
        // =============================== HELPER: wave 3 ======================================================================
        const int xg = x0 + lane;
        const double cnx2 = P.vig_nx2[min(xg, W - 1)];
        constexpr int A3R = A3 > 0 ? A3 : 1;
        uint32_t offr[A3R], offg[A3R], offb[A3R];
        RawRGB raw[A3R];
#pragma unroll
        for (int u = 0; u < A3; ++u) a_offsets(NA - A3 + u, offr[u], offg[u], offb[u]);
        __syncthreads();
        float pf_scan = 1.0f;
        double pf_ny2 = 0.0;
        auto prefetch = [&](int hbn) {
            const int yr = hbn - R + lane;
            if (lane < NB && yr >= y_begin && yr < y_end) { pf_scan = F.scan_row[yr]; pf_ny2 = P.vig_ny2[yr]; }
#pragma unroll
            for (int u = 0; u < A3; ++u) raw[u] = a_load(NA - A3 + u, hbn, offr[u], offg[u], offb[u]);
        };
        prefetch(y_begin - R);
        CC_PRIO(CC_P_HELP);
        int crow0 = 0;
        int hb = y_begin - R;
        for (int n = 0; n < n_iter; ++n, hb += NB, crow0 = crow0 + NB >= CR ? crow0 + NB - CR : crow0 + NB) {
            // ---- phase 1: a9 vignette gain of block n-1's pixels; row constants of block n; its share of A(n) ----
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int y = min(max(hb - NB - R + j, y_begin), y_end - 1);       // rows outside the segment: any valid row, never consumed
                const uint32_t* rt = rowtab + ((y - y_begin) & 15) * 4;
                gvig[j * TW + lane] = vignette_gain(P, cnx2, __hiloint2double((int)rt[2], (int)rt[1]));
            }
            STAMP(4);
            {
                const int yr = hb - R + lane;
                if (lane < NB && yr >= y_begin && yr < y_end) {
                    uint32_t* rt = rowtab + ((yr - y_begin) & 15) * 4;
                    rt[0] = __float_as_uint(pf_scan); rt[1] = (uint32_t)__double2loint(pf_ny2); rt[2] = (uint32_t)__double2hiint(pf_ny2);
                }
            }
            {
                float nv[A3R][3];
#pragma unroll
                for (int u = 0; u < A3; ++u) a_lookup(raw[u], nv[u]);
#pragma unroll
                for (int u = 0; u < A3; ++u) a_write(NA - A3 + u, crow0, raw[u], nv[u]);
            }
            prefetch(hb + NB);
            STAMP(0);
            __syncthreads();
            STAMP(1);
            // ---- phase 2: a11 grain sample * scale of block n's pixels (consumed next trip); its share of B(n) ----
            float* gw = gn + (n & 1) * NB * TW;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int y = min(max(hb - R + j, 0), H - 1);
                const float z = grain_normal(F.key0, F.key1, (uint32_t)y * (uint32_t)W + (uint32_t)xg);
                gw[j * TW + lane] = z * P.noise_scale;
            }
            STAMP(6);

            STAMP(2);
            __syncthreads();
            STAMP(3);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int y = min(max(hb - NB - R + j, y_begin), y_end - 1);
            const uint32_t* rt = rowtab + ((y - y_begin) & 15) * 4;
            gvig[j * TW + lane] = vignette_gain(P, cnx2, __hiloint2double((int)rt[2], (int)rt[1]));
        }
        __syncthreads();
    }
#ifdef CRTFX_STAMP
    if (O.dbg && lane == 0) {
        unsigned long long* d = O.dbg + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) d[i] = stamp_sum[i];
    }
#endif
#undef PK_TAPS
#undef TAP
}


}  // namespace crtfx
