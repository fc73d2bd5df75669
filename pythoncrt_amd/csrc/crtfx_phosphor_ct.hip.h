// crtfx_phosphor_ct.hip.h — k_phosphor_ct: the column-owner kernel of crtfx_phosphor.hip.h (k_phosphor_cc) with fewer LDS
// operations per pixel and a smaller LDS footprint (round 3).  Same stage chain, same arithmetic per sample, same bits
// (tests/test_parity_gpu.py::test_kernel_variants_agree holds the phosphor builds of a launch to identical output).
// (One of the parts of crtfx_kernels.hip.h.)
//
// What k_phosphor_cc's counters said (profiles/r02_z_pmc.json): the LDS pipe is busy 58 % of the kernel, 38 % of that
// on bank conflicts of the three random table gathers per sample (a1 table, lut_g, lut_inv), SQ_WAIT_INST_LDS 21.9 M.
// Per consumer wave and trip of eight rows the tail issued 56 LDS reads; here it issues 24:
//
//   * composite triad table.  With preserve-luma off the two LUT steps of _apply_triad_mask (ref:246-263),
//     lut_inv[idx(lut_g[i] * m)], are a function of the index i and the thread's constant mask value m.  A softened
//     period-3 mask has two distinct interior values for the reference's defaults (on-phosphor / off-phosphor), so the
//     host tabulates T_m[i] = lut_inv[idx(lut_g[i] * m)] for the two most frequent mask values (crtfx_set_params:
//     the same float32 product and truncation the kernels do — bit-identical by construction) and they take the LDS
//     the LUT pair occupied: ONE gather instead of two, and no multiply / index arithmetic between them.  A strip with
//     any other mask value (the replicate-border columns of a softened mask, a 3-valued mask) votes at block start and
//     runs the two-gather form on the LUT pair — a second copy of the consumer loop, chosen per block.
//   * the centre sample of img + s * blur comes back from the frame itself (a byte load through a raw buffer resource
//     with the row offset in an SGPR, issued at the top of phase 1 and consumed behind the barrier; the rows were read by
//     this block's own A phase two trips earlier, so they sit in L2) instead of being parked in an LDS ring by the A phase
//     and re-read with ds_read_u8: no ring stores, no ring reads, 6.4 KB of LDS less.
//   * the scanline gain of a row and the vignette's ny^2 are wave-uniform: scalar loads from the frame's tables through the
//     constant address space (s_load_dword, SGPR operands of the multiply) instead of an LDS row table filled by the
//     helper wave from vector loads.
//
// LDS per block: staging 8.6 KB + H rows 6.1 KB + two tables 8.0 KB + a1 table 1 KB + vignette tile 4 KB + two grain tiles
// 4 KB = 31.8 KB at R = 9 (k_phosphor_cc: 38.3 KB): FIVE blocks per CU where the register budget allows (ct_min_waves).
#pragma once
#include "crtfx_phosphor.hip.h"

namespace crtfx {

#define CONST_AT(T, p) ((const __attribute__((address_space(4))) T*)(uintptr_t)(p))      // wave-uniform index -> s_load

#ifndef CT_WAVES
#define CT_WAVES 5        // resident blocks per CU (= waves per SIMD) the register allocator is asked to leave room for, radii <= 12
#endif
#ifndef CT_NLUT
#define CT_NLUT 0         // a1 of a stored byte from a 256-entry LDS table (1) or as arithmetic (0: 1 KB less LDS — 31 520 B is 25 of gfx950's
#endif                    // 1280-byte LDS granules, five blocks per CU; with the table it is 26 granules and four)
// CT_EXP: timing experiments of build/ab libraries (tools/ab_ct.sh), NEVER part of the product build (crtfx_rr.hip refuses it unless
// CRTFX_TIMING_EXPERIMENT is defined too): each bit removes one part of a trip's work — the frames are then WRONG — to measure
// what that part costs at full occupancy.  1 blur FMAs, 2 the A phase, 4 the pre-warp stores, 8 the helper wave's tiles, 16 the
// tail behind img + s * blur, 32 the loop's barriers, 64 the centre loads, 128 the A phase's frame loads only (a1 + staging writes stay),
// 256 the stores go to a 48 KB window of the scratch image (same instructions, no fabric traffic), 512 the A phase loads ONE dword per
// item instead of three bytes.
#ifndef CT_EXP
#define CT_EXP 0
#endif
#if CT_EXP && !defined(CRTFX_TIMING_EXPERIMENT)
#error "CT_EXP builds write wrong frames: timing experiments only (-DCRTFX_TIMING_EXPERIMENT)"
#endif
#if CT_EXP & 32
#define CT_BARRIER() do {} while (0)
#else
#define CT_BARRIER() __syncthreads()
#endif
__host__ __device__ constexpr int ct_lds_words(int R) {
    return NB * 3 * cc_sws(R) + NB * CC_HROW + 2 * LUT_STRIDE + (CT_NLUT ? 256 : 0) + NB * TW * 2 + 2 * NB * TW;
}
__host__ __device__ constexpr int ct_min_waves(int R) { return R <= 12 ? CT_WAVES : (R <= 20 ? 3 : 2); }

template <int RT>
__global__ __launch_bounds__(RR_THREADS, ct_min_waves(RT)) void k_phosphor_ct(KParams Pin, KGroup G, int seg_rows) {
    const KFrame F = G.f[blockIdx.z];
    const KOut O = G.o[blockIdx.z];
    KParams P = Pin;
    P.flags = SF_FULL;
    P.pix = 0;
    constexpr int PIX = 0;
    extern __shared__ float4 smem4[];
    float* smem = reinterpret_cast<float*>(smem4);
    constexpr int R = RT;
    constexpr int pad = rr_pad(R);
    constexpr int SWP = rr_swp(R);
    constexpr int SWS = cc_sws(R);
    constexpr int L = 2 * R + NB;
    constexpr int NA = (NB * SWP + 63) / 64;             // A-phase wave-items
    constexpr int A3 = CC_A3(NA);                        // ... of the helper wave (the last A3 items)
    constexpr int AO = (NA - A3 + 2) / 3;                // ... of each consumer wave (items wave, wave + 3, ...)
    constexpr int HT = NB * CC_HROW;
    constexpr bool NLUT = CT_NLUT != 0;
    // LDS map, byte offsets from 0 (LDS_AT)
    constexpr uint32_t STG_B = 0;                                            // [NB][3][SWS] float      staging tile
    constexpr uint32_t HROW_B = STG_B + NB * 3 * SWS * 4;                    // [NB][CC_HROW] float     H rows, interleaved like the image row (x, channel)
    constexpr uint32_t LUT_B = HROW_B + HT * 4;                              // [2][LUT_STRIDE] float   composite tables T_m0, T_m1 — or lut_g, lut_inv
    constexpr uint32_t NLUT_B = LUT_B + 2 * LUT_STRIDE * 4;                  // [256] float             u / 255.0
    constexpr uint32_t GVIG_B = NLUT_B + (NLUT ? 256 * 4 : 0);               // [NB][TW] double         vignette gain tile
    constexpr uint32_t GN_B = GVIG_B + NB * TW * 8;                          // [2][NB][TW] float       grain tiles
    static_assert(GN_B + 2 * NB * TW * 4 == (uint32_t)ct_lds_words(R) * 4, "LDS map and ct_lds_words disagree");
    float* stg = smem;
    float* hrow = smem + HROW_B / 4;
    float* lut = smem + LUT_B / 4;
    float* nlut = smem + NLUT_B / 4;
    double* gvig = reinterpret_cast<double*>(smem + GVIG_B / 4);
    float* gn = smem + GN_B / 4;
    if ((uint32_t)(uintptr_t)(lds_f32_t*)smem != 0u) __builtin_trap();      // LDS_AT assumes the dynamic block starts at 0

    const int wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int tid = (threadIdx.x + ((wg_lin & 3) << 6)) & (RR_THREADS - 1);          // roles rotate over the SIMDs with the dispatch number
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int H = P.H, W = P.W;
    const int x0 = blockIdx.x * TW;
    const int y_begin = blockIdx.y * seg_rows;
    const int y_end = min(H, y_begin + seg_rows);
    if (y_begin >= H) return;

    // ---- which triad form this strip runs: every mask value of its 192 floats one of the two tabulated ones? ----------------
    const int f = wave * 64 + lane;                      // consumer threads: float f of the strip's interleaved RGB row segment
    const int fcol = (f < 192 ? f : 0) / 3, fch = (f < 192 ? f : 0) - 3 * fcol;
    const bool fin = x0 + fcol < W;
    const float cm = P.triad_row[min(x0 + fcol, W - 1) * 3 + fch];           // a7 mask of this float
    const uint32_t cmb = __float_as_uint(cm);
    const bool mine = wave == 3 || !fin || cmb == P.comp_m0 || cmb == P.comp_m1;
    // block-wide vote through four words of the (still unused) grain tiles — __syncthreads_and would bring a static LDS word
    // of its own and move the dynamic block off offset 0
    const uint32_t wave_ok = __builtin_amdgcn_ballot_w64(!mine) == 0ull ? 1u : 0u;      // every lane takes part: formed outside the lane test
    if (lane == 0) LDS_AT(lds_u32_t, GN_B + (uint32_t)wave * 4u) = wave_ok;
    __syncthreads();
    const uint32_t votes = LDS_AT(lds_u32_t, GN_B) & LDS_AT(lds_u32_t, GN_B + 4) & LDS_AT(lds_u32_t, GN_B + 8) & LDS_AT(lds_u32_t, GN_B + 12);
    const bool comp = P.triad_comp != nullptr && __builtin_amdgcn_readfirstlane((int)votes) != 0;      // block-uniform, and known to be: a scalar branch
    {
        const float* t0 = comp ? P.triad_comp : P.lut_g;
        const float* t1 = comp ? P.triad_comp + LUT_N : P.lut_inv;
        for (int i = tid; i < LUT_N; i += RR_THREADS) { lut[i] = t0[i]; lut[LUT_STRIDE + i] = t1[i]; }
    }
    if constexpr (NLUT) { if (tid < 256) nlut[tid] = norm_u8((uint32_t)tid); }
    const float* taps = P.taps;
    // the taps as R + 1 aligned SGPR pairs (tap[2m], tap[2m+1]); see k_phosphor_cc
    unsigned long long tp[R + 1];
#pragma unroll
    for (int m = 0; m <= R; ++m)
        tp[m] = (unsigned long long)__float_as_uint(taps[2 * m]) | ((unsigned long long)(2 * m + 1 <= 2 * R ? __float_as_uint(taps[2 * m + 1]) : 0u) << 32);
#define PK_TAPS(acc, wpair, whigh, t) pk_fma_bcast(acc, wpair, whigh, ((t) & 1) ? tp[((t) - 1) / 2] : tp[(2 * R - (t)) / 2], ((t) & 1) != 0)
    const uint32_t row_elems = (uint32_t)W * 3u;
    const int n_iter = (y_end + R - (y_begin - R) + NB - 1) / NB;                    // loop trips (same for both roles)
#ifdef CRTFX_STAMP
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last) :: "memory");
#endif

    // ---- pieces shared by the two roles ---------------------------------------------------------------------------------------
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
#if CT_EXP & 128
        return RawRGB{(ro + o_r) & 255u, (ro + o_g) & 255u, (ro + o_b) & 255u};
#elif CT_EXP & 512
        const uint32_t d = *reinterpret_cast<const uint32_t*>(F.in + ((ro + o_g) & ~3u));
        return RawRGB{d & 255u, (d >> 8) & 255u, (d >> 16) & 255u};
#else
        return load_raw(PIX, F.in, ro + o_r, ro + o_g, ro + o_b);
#endif
    };
    // a1 — u / 255.0 of a stored byte: the LDS table, or three instructions: with c_hi + c_lo = 1/255 to 48 bits,
    // fma(f, c_hi, f * c_lo) is the correctly rounded quotient for every byte (the sum carries f / 255 to ~2^-48 relative and no
    // f / 255 lies that close to a rounding boundary: its bits beyond the mantissa repeat f's own eight; checked exhaustively on
    // the host with exact rationals, and against k_phosphor_cc on the device: tests/test_parity_gpu.py::test_composite_triad_tables)
    auto a1 = [&](uint32_t u) -> float {
        if constexpr (NLUT) return LDS_AT(lds_f32_t, NLUT_B + (u << 2));
        else { const float f = (float)u; return fmaf(f, 0x1.010102p-8f, f * -0x1.fdfdfep-33f); }
    };
    auto a_write = [&](int q, const float (&o)[3]) {
        const int it = min((q << 6) + lane, NB * SWP - 1);
        const int j = it / SWP, i = it - j * SWP;
        float* sp = stg + (j * 3) * SWS + i;
        sp[0] = o[0]; sp[SWS] = o[1]; sp[2 * SWS] = o[2];
    };
    // H pass of the staging tile by a consumer wave (k_phosphor_cc's: 8 adjacent outputs per lane, lanes mapped through the
    // hardware's 16-lane ds_read_b128 groups; taps left to right, fused — the oracle's RowFilter order)
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
        for (int qq = 0; qq < NQ; ++qq) vq[qq] = srow[qq];       // all reads in flight before the first tap
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
            const f32x4 vv = vq[qq];
            const f32x2 vp[2] = {{vv[0], vv[1]}, {vv[2], vv[3]}};
#if CT_EXP & 1
            acc2[qq & 3].x += vv[0] + vv[1]; acc2[qq & 3].y += vv[2] + vv[3]; (void)vp; (void)off;
#else
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const int t = 4 * qq + e - 2 * pp - off;      // tap of the pair's first output; its second takes t - 1
                    if (t == 0) acc2[pp].x = fmaf(vp[e >> 1][e & 1], taps[0], acc2[pp].x);
                    else if (t >= 1 && t <= 2 * R) PK_TAPS(acc2[pp], vp[e >> 1], (e & 1) != 0, t);
                    else if (t == 2 * R + 1) acc2[pp].y = fmaf(vp[e >> 1][e & 1], taps[0], acc2[pp].y);      // tap[2R] == tap[0]
                }
#endif
        }
        float* hp = hrow + j * CC_HROW + 8 * g8 * 3 + c;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) { hp[6 * pp] = acc2[pp].x; hp[6 * pp + 3] = acc2[pp].y; }
    };

    if (wave < 3) {
        // =============================== CONSUMER: waves 0-2 ================================================================
        // COMP: composite tables in LDS (one gather) or the LUT pair (two gathers and the mask multiply between them)
        auto consumer = [&](auto comp_c) __attribute__((always_inline)) {
            constexpr bool COMP = decltype(comp_c)::value;
            const uint32_t tsel = (cmb == P.comp_m1 && P.comp_m1 != P.comp_m0) ? LUT_B + LUT_STRIDE * 4 : LUT_B;      // this float's table (COMP)
            const uint32_t gcol8 = (uint32_t)fcol * 8u, gcol4 = (uint32_t)fcol * 4u;  // its pixel in the vignette / grain tiles
            // pre-warp image out through a buffer resource (k_phosphor_cc): offsets past the image are dropped by the hardware
            // — and the resource covers this block's row segment ONLY, so rows above / below it (the first trips' and the last
            // trip's garbage rows) fall outside by themselves: no per-row test at all
            const __amdgpu_buffer_rsrc_t pre_rsrc = __builtin_amdgcn_make_buffer_rsrc(O.pre + (size_t)y_begin * (size_t)W * 3u, 0,
                                                                                      (int)((uint32_t)(y_end - y_begin) * (uint32_t)W * 12u), 0x00020000);
            const uint32_t row_b = fin ? (uint32_t)W * 12u : 0u;                     // bytes per pre-warp image row (this lane's stride)
            // the frame's scanline row gains through a buffer resource too: rows outside the frame read as 0 (never consumed)
            const __amdgpu_buffer_rsrc_t scan_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(F.scan_row), 0, H * 4, 0x00020000);
            // centre samples in through a buffer resource over the frame: per-thread byte offset inside a row (a2: R from x - d,
            // B from x + d, wrapped, ref:571-577), the row's offset in an SGPR
            const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(F.in), 0, (int)((uint32_t)H * row_elems), 0x00020000);
            uint32_t coff;
            {
                const int x = min(x0 + fcol, W - 1);
                int xs = x;
                if (P.ab != 0 && fch != 1) xs = wrap(fch == 0 ? x - P.ab : x + P.ab, W);
                coff = (uint32_t)xs * 3u + (uint32_t)fch;
            }
            f32x2 win2[L / 2];
#pragma unroll
            for (int i = 0; i < L / 2; ++i) win2[i] = f32x2{0.0f, 0.0f};
            auto v_pass = [&](float (&blur)[NB]) {
                const float* hcol = hrow + f;
#pragma unroll
                for (int j = 0; j < NB; ++j) win2[(2 * R + j) >> 1][j & 1] = hcol[j * CC_HROW];
                f32x2 acc[NB / 2];
#pragma unroll
                for (int jp = 0; jp < NB / 2; ++jp) acc[jp] = f32x2{0.0f, 0.0f};
#if CT_EXP & 1
#pragma unroll
                for (int i = 0; i < L; ++i) { if (i & 1) acc[(i >> 1) & 3].y += win2[i >> 1][1]; else acc[(i >> 1) & 3].x += win2[i >> 1][0]; }
#else
#pragma unroll
                for (int i = 0; i < L; ++i)
#pragma unroll
                    for (int jp = 0; jp < NB / 2; ++jp) {
                        const int t = i - 2 * jp;
                        if (t == 0) acc[jp].x = fmaf(win2[i >> 1][i & 1], taps[0], acc[jp].x);
                        else if (t >= 1 && t <= 2 * R) PK_TAPS(acc[jp], win2[i >> 1], (i & 1) != 0, t);
                        else if (t == 2 * R + 1) acc[jp].y = fmaf(win2[i >> 1][i & 1], taps[0], acc[jp].y);       // tap[2R] == tap[0]
                    }
#endif
#pragma unroll
                for (int jp = 0; jp < NB / 2; ++jp) { blur[2 * jp] = acc[jp].x; blur[2 * jp + 1] = acc[jp].y; }
#pragma unroll
                for (int i = 0; i < R; ++i) win2[i] = win2[i + NB / 2];
            };
            // the centre bytes of output rows yb .. yb + 7.  The whole offset rides in the VGPR operand — the hardware's range check
            // covers the vector offset only, not the scalar one — so a row outside the frame (modulo 2^32 when it is above it)
            // is an offset outside the resource and reads 0 instead of touching memory; those rows are never consumed
            auto centre_load = [&](int yb, uint32_t (&cb)[NB]) {
                uint32_t vo = coff + (uint32_t)yb * row_elems;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
#if CT_EXP & 64
                    cb[j] = vo & 255u;
#else
                    cb[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(in_rsrc, vo, 0, 0);
#endif
                    vo += row_elems;
                }
            };
            // the scanline gains of rows yb .. yb + 7: lane l asks for row yb + (l & 7) — ONE load per wave and trip, one VGPR across the
            // barrier — and the tail reads row j's gain out of lane j (v_readlane: an SGPR operand of its multiply).  Out-of-frame rows read 0.
            const uint32_t lane7x4 = (uint32_t)(lane & 7) * 4u;
            auto scan_load = [&](int yb) -> uint32_t { return __builtin_amdgcn_raw_buffer_load_b32(scan_rsrc, (uint32_t)yb * 4u + lane7x4, 0, 0); };
            // a7 for the eight rows
            auto triad = [&](float (&v)[NB]) {
                if constexpr (COMP) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) v[j] = LDS_AT(lds_f32_t, tsel + ((uint32_t)lut_index_unit(v[j]) << 2));             // ref:250-252 + :261-262 composed
                } else {
#pragma unroll
                    for (int j = 0; j < NB; ++j) v[j] = LDS_AT(lds_f32_t, LUT_B + ((uint32_t)lut_index_unit(v[j]) << 2)) * cm;       // ref:250-252
#pragma unroll
                    for (int j = 0; j < NB; ++j) v[j] = LDS_AT(lds_f32_t, LUT_B + LUT_STRIDE * 4 + ((uint32_t)lut_index(v[j]) << 2));   // ref:261-262
                }
            };
            uint32_t offr[AO], offg[AO], offb[AO];
            RawRGB raw[AO];
#pragma unroll
            for (int u = 0; u < AO; ++u) a_offsets(min(wave + 3 * u, NA - A3 - 1), offr[u], offg[u], offb[u]);
            __syncthreads();                                // tables visible
#pragma unroll
            for (int u = 0; u < AO; ++u) raw[u] = a_load(min(wave + 3 * u, NA - A3 - 1), y_begin - R, offr[u], offg[u], offb[u]);
            // eight stores behind the first prefetch, as in every later trip (k_phosphor_cc: the loop is entered with the same
            // count of vector memory operations younger than the prefetched bytes as its back edge carries)
#pragma unroll
            for (int j = 0; j < NB; ++j) __builtin_amdgcn_raw_buffer_store_b32(0u, pre_rsrc, 0xFFFFFF00u - 16u * (uint32_t)j, 0, 0);      // out of range: dropped
            CC_PRIO(CC_P_A);
            int hb = y_begin - R;
            uint32_t cbp0 = 0u, cbp1 = 0u;               // the trip's eight centre bytes, packed (trip 0: rows above the segment, never consumed)
            uint32_t off0 = fin ? (uint32_t)(-2 * R - NB) * row_b + ((uint32_t)x0 * 3u + (uint32_t)f) * 4u : 0xFFFFFF00u;      // (row hb - NB - R of the segment, float f), modulo 2^32 while that row is above it
            for (int n = 0; n < n_iter; ++n, hb += NB, off0 += (uint32_t)NB * row_b) {
                // ---- phase 1 ----
                const int yb = hb - NB - R;                 // first output row of block n-1 (garbage rows in trip 0)
                const uint32_t slv = scan_load(yb);          // in front of the prefetch below: its wait leaves those loads in flight
                float v[NB];
                {
                    float blur[NB];
                    CC_PRIO(CC_P_VH);
                    v_pass(blur);
                    CC_PRIO(CC_P_A);
#pragma unroll
                    for (int j = 0; j < NB; ++j)         // a1 of the centre byte (v_cvt_f32_ubyteN; a2 is in the load's column); ref:611 — only v[] crosses the barrier
                        v[j] = clip01(a1(((j < 4 ? cbp0 : cbp1) >> (8 * (j & 3))) & 255u) + P.bloom_strength * blur[j]);
                }
                STAMP(4);
#if !(CT_EXP & 2)
                {
                    float nv[AO][3];
#pragma unroll
                    for (int u = 0; u < AO; ++u) { nv[u][0] = a1(raw[u].r); nv[u][1] = a1(raw[u].g); nv[u][2] = a1(raw[u].b); }
#pragma unroll
                    for (int u = 0; u < AO; ++u) a_write(min(wave + 3 * u, NA - A3 - 1), nv[u]);
                }
#endif
                // the centre bytes of the NEXT trip's rows (L2 hits: staged a trip ago), requested IN FRONT of the prefetch and of
                // this trip's stores: vector memory operations complete in order, so a wait for them placed behind either would
                // also be a wait for the prefetch's HBM round trip / the stores' acknowledgements.  Packed into two registers in the tail.
                uint32_t cb[NB];
                centre_load(yb + NB, cb);
                __builtin_amdgcn_sched_barrier(0);          // ... and the scheduler keeps them in front
#if !(CT_EXP & 2)
#pragma unroll
                for (int u = 0; u < AO; ++u) raw[u] = a_load(min(wave + 3 * u, NA - A3 - 1), hb + NB, offr[u], offg[u], offb[u]);   // past the last block: clamped rows, never consumed
#endif
                STAMP(0);
                CT_BARRIER();
                STAMP(1);
                // ---- phase 2: C2 of block n-1 (output rows yb + j), stage by stage over the eight rows ----
                CC_PRIO(CC_P_C2);
                const uint32_t gt_b = GN_B + (uint32_t)(((n & 1) ^ 1) * NB * TW * 4) + gcol4;
#if !(CT_EXP & 16)
                float gnv[NB];
                double gv[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    gv[j] = LDS_AT(lds_f64_t, GVIG_B + (uint32_t)(j * TW * 8) + gcol8);
                    gnv[j] = LDS_AT(lds_f32_t, gt_b + (uint32_t)(j * TW * 4));
                }
                triad(v);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float r = clip01(v[j] * __uint_as_float(__builtin_amdgcn_readlane(slv, j)));   // ref:617-624
                    double d = (double)r * gv[j];                                                       // ref:626-628 (gain in [0,1]: no clip)
                    d = clip01(d + (double)gnv[j]);                                                     // ref:646-647
                    v[j] = (float)d;
                }
#else
                (void)gt_b; (void)slv;
#endif
                cbp0 = cb[0] | (cb[1] << 8) | (cb[2] << 16) | (cb[3] << 24);
                cbp1 = cb[4] | (cb[5] << 8) | (cb[6] << 16) | (cb[7] << 24);
                {
                    uint32_t boff = off0;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
#if CT_EXP & 4
                        asm volatile("" :: "v"(v[j]), "v"(boff));
#elif CT_EXP & 256
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j]), pre_rsrc, (boff < 0xF0000000u ? (boff & 0xFFFFu) : boff), 0, 0);
#else
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j]), pre_rsrc, boff, 0, 0);
#endif
                        boff += row_b;
                    }
                }
                STAMP(6);
                CC_PRIO(CC_P_VH);
                h_pass(wave);
                CC_PRIO(CC_P_A);
                STAMP(2);
                CT_BARRIER();
                STAMP(3);
            }
            // ---- drain: C1 and C2 of the last block ----
            {
                const int yb = hb - NB - R;
                const uint32_t slv = scan_load(yb);
                float v[NB], blur[NB];
                v_pass(blur);
                __syncthreads();
                const float* gt = gn + ((n_iter & 1) ^ 1) * NB * TW;
#pragma unroll
                for (int j = 0; j < NB; ++j) v[j] = clip01(a1(((j < 4 ? cbp0 : cbp1) >> (8 * (j & 3))) & 255u) + P.bloom_strength * blur[j]);
                triad(v);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int y = yb + j;
                    const float r = clip01(v[j] * __uint_as_float(__builtin_amdgcn_readlane(slv, j)));
                    double d = (double)r * gvig[j * TW + fcol];
                    d = clip01(d + (double)gt[j * TW + fcol]);
                    if (y >= y_begin && y < y_end && fin) O.pre[((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u + (uint32_t)f] = (float)d;
                }
            }
        };
        if (comp) consumer(std::true_type{});
        else consumer(std::false_type{});
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
#pragma unroll
        for (int u = 0; u < A3; ++u) raw[u] = a_load(NA - A3 + u, y_begin - R, offr[u], offg[u], offb[u]);
        CC_PRIO(CC_P_HELP);
        int hb = y_begin - R;
        // a9 vignette gain of the 8 x 64 pixels of output rows yb .. yb + 7: ny^2 of a row is wave-uniform (s_load_dwordx2)
        const __amdgpu_buffer_rsrc_t ny2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P.vig_ny2), 0, H * 8, 0x00020000);
        auto vig_tile = [&](int yb) {
            // ny^2 of rows yb .. yb + 7: 8-byte loads at wave-uniform addresses, the offset in the (range-checked) vector operand;
            // rows outside the frame read 0 (never consumed)
            const uint32_t vo = (uint32_t)yb * 8u;
            uint64_t q[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) q[j] = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(ny2_rsrc, vo + 8u * (uint32_t)j, 0, 0));
#pragma unroll
            for (int j = 0; j < NB; ++j) gvig[j * TW + lane] = vignette_gain(P, cnx2, __builtin_bit_cast(double, q[j]));
        };
        for (int n = 0; n < n_iter; ++n, hb += NB) {
            // ---- phase 1: a9 vignette gain of block n-1's pixels; its share of A(n) ----
#if !(CT_EXP & 8)
            vig_tile(hb - NB - R);
#endif
            STAMP(4);
#if !(CT_EXP & 2)
            {
                float nv[A3R][3];
#pragma unroll
                for (int u = 0; u < A3; ++u) { nv[u][0] = a1(raw[u].r); nv[u][1] = a1(raw[u].g); nv[u][2] = a1(raw[u].b); }
#pragma unroll
                for (int u = 0; u < A3; ++u) a_write(NA - A3 + u, nv[u]);
            }
#pragma unroll
            for (int u = 0; u < A3; ++u) raw[u] = a_load(NA - A3 + u, hb + NB, offr[u], offg[u], offb[u]);
#endif
            STAMP(0);
            CT_BARRIER();
            STAMP(1);
            // ---- phase 2: a11 grain sample * scale of block n's pixels (consumed next trip) ----
#if !(CT_EXP & 8)
            float* gw = gn + (n & 1) * NB * TW;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int y = min(max(hb - R + j, 0), H - 1);
                const float z = grain_normal(F.key0, F.key1, (uint32_t)y * (uint32_t)W + (uint32_t)xg);
                gw[j * TW + lane] = z * P.noise_scale;
            }
#endif
            STAMP(6);
            STAMP(2);
            CT_BARRIER();
            STAMP(3);
        }
        vig_tile(hb - NB - R);
        __syncthreads();
    }
#ifdef CRTFX_STAMP
    if (O.dbg && lane == 0) {
        unsigned long long* d = O.dbg + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) d[i] = stamp_sum[i];
    }
#endif
#undef PK_TAPS
}

}  // namespace crtfx
