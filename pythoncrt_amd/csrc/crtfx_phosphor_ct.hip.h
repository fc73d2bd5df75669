// crtfx_phosphor_ct.hip.h — k_phosphor_ct: the column-owner kernel of crtfx_phosphor.hip.h (k_phosphor_cc) rebuilt around what
// round 3's ablation runs showed bounds it: VECTOR-MEMORY INSTRUCTIONS, not VALU or LDS.  Same stage chain, same arithmetic per
// sample, same bits (tests/test_parity_gpu.py::test_kernel_variants_agree, ::test_composite_triad_tables).
// (One of the parts of crtfx_kernels.hip.h.)
//
// The measurements (profiles/r03_ct_ablation.txt; 4K, R = 9, us per 2-frame launch, k_phosphor_cc = 128): removing the blur's
// FMAs saves 4 %, the whole pointwise tail 0 %, the barriers 3 % — but the A phase's frame loads alone 11 %, the pre-warp
// stores 16 % (10 % of it fabric traffic) and eight byte loads per wave and trip for the centre samples 6 %.  k_phosphor_cc
// issues 62 vector-memory instructions per block and trip of eight rows — 33 of them single-byte loads of the frame, three per
// wave-item of 64 staged pixels, because R and B are fetched at a shifted column (a2) — and the texture-address unit is busy
// 15-18 cycles per instruction whatever its width (profiles/r03_cc_vmem.json: TA busy 51 % of the kernel, its address / command
// FIFOs full 1.1-1.4 M times per launch; this kernel: 32 %, never full).  So:
//
//   * the A phase loads DWORDS.  A strip's staged row segment (64 + 2 pad pixels, R and B displaced by the aberration) is one
//     contiguous window of the frame row: (88 + 2|d|) * 3 bytes = 68 aligned dwords at R = 9, d = 1.  A lane loads one dword
//     of one row (9 wave-loads per trip instead of 33 byte loads), converts its four bytes (v_cvt_f32_ubyte0..3 — the byte
//     select is free — and a two-instruction exact u / 255, see a1) and scatters them to the (channel, column) slots of the
//     staging tile they belong to; the slots are block-invariant and sit in registers.  The raw dword also goes into an LDS
//     ring of row windows (32 rows of 256 bytes: the window dwords that overlap the centre pixels), from which the tail
//     reads its centre sample with ds_read_u8 — eight rows per trip at one address + immediates: no second fetch of the
//     frame, no byte loads at all.  Rows need W % 4 == 0 (dword-aligned rows) and a window inside the frame: the first and last
//     strip(s), where BORDER_REPLICATE and the aberration's wrap bend the window, run k_phosphor_cc's byte-wise A phase
//     (a second copy of the loop, chosen per block; blocks of both kinds share a launch).
//   * composite triad table (the same blocks).  With preserve-luma off the two LUT steps of _apply_triad_mask (ref:246-263),
//     lut_inv[idx(lut_g[i] * m)], are a function of the index i and the thread's constant mask value m.  A softened
//     period-3 mask has two distinct interior values for the reference's defaults, so the host tabulates
//     T_m[i] = lut_inv[idx(lut_g[i] * m)] for the two most frequent mask values (crtfx_set_params: the kernels' own float32
//     product and truncation) and they take the LDS the LUT pair occupied: one gather instead of two.  A strip with any
//     other mask value votes at block start and runs the two-gather form.
//   * no vector-memory wait ever covers a store or the next trip's prefetch: the waits are counted (vmcnt) so that the
//     eight stores and the prefetched dwords stay in flight across them — vector-memory operations complete in order, and a
//     wait placed behind the stores would also be a wait for their acknowledgements from the fabric.
//   * the scanline gain of a row is wave-uniform: lane l loads row yb + (l & 7) (one load per wave and trip), the tail takes
//     row j's gain out of lane j with v_readlane (an SGPR operand of its multiply); the vignette's ny^2 likewise comes from
//     wave-uniform buffer loads in the helper wave.  No LDS row table.
//   * stores through a buffer resource over the block's OWN row segment: rows above / below it fall outside the resource
//     and are dropped by the hardware — no per-row test.
#pragma once
#include "crtfx_phosphor.hip.h"

namespace crtfx {

#ifndef CT_WAVES
#define CT_WAVES 4        // resident blocks per CU (= waves per SIMD) the register allocator is asked to leave room for (radii 13 .. 15: with 4 - 6 VGPRs spilled,
                          // still 15 - 20 % ahead of three blocks without spills)
#endif
// The ablation builds behind profiles/r03_ct_ablation.txt (-DCT_EXP=n: one part of a trip removed, frames wrong, timing only), the
// a1-from-an-LDS-table and packed-a1 variants lived in this file up to the commit that recorded their results; they are not part of
// the product source (git log -S CT_EXP -- this file).
#ifndef CT_WARM_SKIP
#define CT_WARM_SKIP 1    // skip the taps and the tail of the trips whose output rows all lie above the block's segment (window fill only)
#endif

// a frame-row window in dwords, at most: (staged pixels + 2 * 8 of aberration) * 3 bytes, + 3 of alignment slack, + 1
__host__ __device__ constexpr int ct_ndmax(int R) { return ((rr_swp(R) + 16) * 3 + 5) / 4 + 1; }
// the centre ring: CT_RING_ROWS rows (a power of two >= R + 2 NB for every radius this kernel serves) of TW dwords
#ifndef CT_RING_ROWS_N
#define CT_RING_ROWS_N 32  // (overridable only so that tests/test_evidence_tools.py's resource guard can be shown to fail: -DCT_RING_ROWS_N=64 -> 47.9 KB, three blocks per CU)
#endif
constexpr int CT_RING_ROWS = CT_RING_ROWS_N;
// half frames (PIX = 1): no ring.  A sample is two bytes, so a ring row of frame-row windows would be 512 bytes and the ring 16 KB (three blocks
// per CU).  Instead the A phase parks the window QWORDS (four half samples each) of the trip's eight rows in a one-trip RAW TILE (8 rows of 64
// qwords = 4 KB) and each consumer thread — which owns one float of the strip row in the V pass and the tail — keeps the raw centre samples of
// its float for the R + NB rows between staging and use in a REGISTER WINDOW of packed halves ((R + NB + 1) / 2 VGPRs: 9 at R = 9), next to the
// V pass's row window: 34.8 KB of LDS at R = 9, four blocks per CU.
__host__ __device__ constexpr int ct_ring_words(int R, int pix = 0) { return pix ? NB * TW * 2 : CT_RING_ROWS * TW; }
// LDS words: staging, one H-row tile, two tables, the centre ring / raw tile, vignette tile (f64), two grain tiles (f32): 38.8 KB at R = 9 (uint8)
__host__ __device__ constexpr int ct_lds_words(int R, int pix = 0) {
    return NB * 3 * cc_sws(R) + NB * CC_HROW + 2 * LUT_STRIDE + ct_ring_words(R, pix) + NB * TW * 2 + 2 * NB * TW;
}
__host__ __device__ constexpr int ct_min_waves(int R) { return R <= 15 ? CT_WAVES : (R <= 20 ? 3 : 2); }
#define CT_HALF_C_HI 0x1.0101p-8f          // float(h) / 255 as fma(f, C_HI, f * C_LO): 1/255 rounded down to float32 ...
#define CT_HALF_C_LO 0x1.010102p-32f       // ... and float32(1/255 - C_HI) > 0
#ifndef CT_HALF_MAX_RADIUS
#define CT_HALF_MAX_RADIUS 15      // the half build's radii (crtfx_rr.hip, launch_rr_group).  Its centre window takes (R + 9) / 2 VGPRs more than the uint8 build: no spills up to
                                   // radius 12 (126 VGPRs), 15 - 22 spilled at four blocks from 13 — still ahead of or level with k_phosphor_rr<half> at three
                                   // (4K half frames, us per frame: R 13 126 against 131 - 154, R 14 140 / 139 - 159, R 15 139 - 155 / 144 - 156; profiles/r05_half_sigma.txt)
#endif

template <int RT, int PIX = 0>
__global__ __launch_bounds__(RR_THREADS, ct_min_waves(RT)) void k_phosphor_ct(KParams Pin, KGroup G, int seg_rows) {
    const KFrame F = G.f[blockIdx.z];
    const KOut O = G.o[blockIdx.z];
    KParams P = Pin;
    P.flags = SF_FULL;
    P.pix = PIX;
    constexpr bool HALF = PIX != 0;
    extern __shared__ float4 smem4[];
    float* smem = reinterpret_cast<float*>(smem4);
    constexpr int R = RT;
    constexpr int pad = rr_pad(R);
    constexpr int SWP = rr_swp(R);
    constexpr int SWS = cc_sws(R);
    constexpr int L = 2 * R + NB;
    constexpr int CR = rr_cring(R);
    constexpr int HT = NB * CC_HROW;
    // byte-wise A phase (edge strips): wave-items of 64 staged pixels, as in k_phosphor_cc
    constexpr int NA = (NB * SWP + 63) / 64;
    constexpr int A3 = CC_A3(NA);                        // ... of the helper wave (the last A3 items)
    constexpr int AO = (NA - A3 + 2) / 3;                // ... of each consumer wave (items wave, wave + 3, ...)
    // dword A phase: wave-items of 64 (row, dword) pairs of the NB row windows
    constexpr int NDMAX = ct_ndmax(R);
    // centre ring geometry: 32 rows of 256 bytes.  A staged row r sits in ring row (r - (y_begin - R) + RSH) & 31, RSH = R % NB: the eight rows
    // the tail reads in one trip then start on a multiple of eight and never wrap — ONE address per trip + immediates j * 256 — and the
    // modulo of the (per-lane) write row is a mask.  A row holds the 64 packed centre pixels (byte-wise path) or the dwords of the frame-row
    // window that overlap the centre pixels' bytes (at most 62 for |d| <= 8; dword 63 takes the window's other dwords)
    constexpr int RSH = R % NB;
    static_assert(HALF || (CR <= CT_RING_ROWS && (CT_RING_ROWS & (CT_RING_ROWS - 1)) == 0), "the centre ring holds R + 2 NB rows in a power-of-two ring");
    constexpr uint32_t RING_MASK = (uint32_t)(CT_RING_ROWS * TW * 4 - 1);
    constexpr int NQF = (NB * NDMAX + 63) / 64;
    constexpr int FO = (NQF + 4) / 5;                    // ... of each consumer wave (items wave, wave + 3, ...)
    constexpr int FH = NQF - 3 * FO > 0 ? NQF - 3 * FO : 0;      // ... of the helper wave (the last ones: mostly past a short window's end)
    // LDS map, byte offsets from 0 (LDS_AT)
    constexpr uint32_t STG_B = 0;                                            // [NB][3][SWS] float      staging tile
    constexpr uint32_t HROW_B = STG_B + NB * 3 * SWS * 4;                    // [NB][CC_HROW] float     H rows, interleaved like the image row (x, channel)
    constexpr uint32_t LUT_B = HROW_B + HT * 4;                              // [2][LUT_STRIDE] float   composite tables T_m0, T_m1 — or lut_g, lut_inv
    constexpr uint32_t RING_B = LUT_B + 2 * LUT_STRIDE * 4;                  // [32][TW] dword          centre ring: frame-row window dwords (fast path) / packed centre pixels (byte-wise path)
                                                                             // half: [NB][TW] qword    raw tile of this trip's rows: window qwords / the centre pixels' 3 x uint16
    constexpr uint32_t GVIG_B = RING_B + ct_ring_words(R, PIX) * 4;          // [NB][TW] double         vignette gain tile
    constexpr uint32_t GN_B = GVIG_B + NB * TW * 8;                          // [2][NB][TW] float       grain tiles
    static_assert(GN_B + 2 * NB * TW * 4 == (uint32_t)ct_lds_words(R, PIX) * 4, "LDS map and ct_lds_words disagree");
    static_assert(SWS - SWP >= 4, "the dword A phase parks the bytes it does not stage in the four pad floats behind a staging plane");
    float* stg = smem;
    float* hrow = smem + HROW_B / 4;
    float* lut = smem + LUT_B / 4;
    double* gvig = reinterpret_cast<double*>(smem + GVIG_B / 4);
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
    const uint32_t row_elems = (uint32_t)W * 3u;

    // ---- which form this strip runs ---------------------------------------------------------------------------------------------
    // (1) every mask value of its 192 floats one of the two tabulated ones?  A block-wide vote through four words of the (still
    // unused) grain tiles — __syncthreads_and would bring a static LDS word of its own and move the dynamic block off offset 0
    const int f = wave * 64 + lane;                      // consumer threads: float f of the strip's interleaved RGB row segment
    const int fcol = (f < 192 ? f : 0) / 3, fch = (f < 192 ? f : 0) - 3 * fcol;
    const bool fin = x0 + fcol < W;
    const float cm = P.triad_row[min(x0 + fcol, W - 1) * 3 + fch];           // a7 mask of this float
    const uint32_t cmb = __float_as_uint(cm);
    const bool mine = wave == 3 || !fin || cmb == P.comp_m0 || cmb == P.comp_m1;
    const uint32_t wave_ok = __builtin_amdgcn_ballot_w64(!mine) == 0ull ? 1u : 0u;      // every lane takes part: formed outside the lane test
    if (lane == 0) LDS_AT(lds_u32_t, GN_B + (uint32_t)wave * 4u) = wave_ok;
    __syncthreads();
    const uint32_t votes = LDS_AT(lds_u32_t, GN_B) & LDS_AT(lds_u32_t, GN_B + 4) & LDS_AT(lds_u32_t, GN_B + 8) & LDS_AT(lds_u32_t, GN_B + 12);
    // (2) the row window inside the frame (no BORDER_REPLICATE clamp, no aberration wrap) and dword-aligned rows?
    const int aab = P.ab < 0 ? -P.ab : P.ab;
    const int px_lo = x0 - pad - aab, px_hi = x0 - pad + SWP - 1 + aab;      // first / last frame column the window touches
    const bool interior = px_lo >= 0 && px_hi <= W - 1 && (W & 3) == 0;
    bool fast = P.triad_comp != nullptr && __builtin_amdgcn_readfirstlane((int)votes) != 0 && interior;      // block-uniform, and known to be: a scalar branch
    const uint32_t a_lo = ((uint32_t)(px_lo > 0 ? px_lo : 0) * 3u) & ~3u;    // the window's first byte in a frame row, dword-aligned
    const int ND = interior ? (int)(((uint32_t)px_hi * 3u + 2u - a_lo) / 4u) + 1 : 1;      // its dwords (<= NDMAX)
    {
        const float* t0 = fast ? P.triad_comp : P.lut_g;
        const float* t1 = fast ? P.triad_comp + LUT_N : P.lut_inv;
        for (int i = tid; i < LUT_N; i += RR_THREADS) { lut[i] = t0[i]; lut[LUT_STRIDE + i] = t1[i]; }
    }
    const float* taps = P.taps;
    // the taps as R + 1 aligned SGPR pairs (tap[2m], tap[2m+1]); see k_phosphor_cc
    unsigned long long tp[R + 1];
#pragma unroll
    for (int m = 0; m <= R; ++m)
        tp[m] = (unsigned long long)__float_as_uint(taps[2 * m]) | ((unsigned long long)(2 * m + 1 <= 2 * R ? __float_as_uint(taps[2 * m + 1]) : 0u) << 32);
#define PK_TAPS(acc, wpair, whigh, t) pk_fma_bcast(acc, wpair, whigh, ((t) & 1) ? tp[((t) - 1) / 2] : tp[(2 * R - (t)) / 2], ((t) & 1) != 0)
    const int n_iter = (y_end + R - (y_begin - R) + NB - 1) / NB;                    // loop trips (same for both roles)
#ifdef CRTFX_STAMP
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last) :: "memory");
#endif

    // ---- pieces shared by the two roles ---------------------------------------------------------------------------------------
    // a1 — u / 255.0 of a stored byte in three instructions: with c_hi + c_lo = 1/255 to 48 bits, fma(f, c_hi, f * c_lo) is the
    // correctly rounded quotient for every byte (the sum carries f / 255 to ~2^-48 relative and no f / 255 lies that close to a
    // rounding boundary: its bits beyond the mantissa repeat f's own eight; checked with exact rationals on the host, and
    // against k_phosphor_cc's table of IEEE quotients on the device: tests/test_parity_gpu.py::test_composite_triad_tables)
    // Half frames: the same two instructions behind the conversion, with the pair split the OTHER way — c_hi = 1/255 rounded DOWN, c_lo > 0.  For
    // every one of the 65 536 half bit patterns fma(f, c_hi, f * c_lo) is the IEEE quotient float(h) / 255.0f: correctly rounded for the finite
    // ones, and with a positive c_lo the signed zeros, the infinities and NaN come out as the division gives them too (with the byte pair, whose
    // c_lo is negative, -0 would turn into +0 and inf into NaN) — checked with exact rationals by tests/test_host_tables.py::
    // test_half_quotient_constants_exact, and on the device against norm_px's form in k_phosphor_rr (test_fp16_column_owner_kernel).
    auto a1 = [&](uint32_t u) -> float {
        if constexpr (HALF) {
            const float fh = (float)__builtin_bit_cast(_Float16, (unsigned short)u);
            return fmaf(fh, CT_HALF_C_HI, fh * CT_HALF_C_LO);
        }
        const float fu = (float)u; return fmaf(fu, 0x1.010102p-8f, fu * -0x1.fdfdfep-33f);
    };
    // two bytes at once: the multiply and the fma as ONE packed instruction each (v_pk_mul_f32, v_pk_fma_f32) — the same two roundings per byte
    auto a1x2 = [&](uint32_t u0, uint32_t u1, float& o0, float& o1) {
        o0 = a1(u0); o1 = a1(u1);
    };
    // -- byte-wise A phase (k_phosphor_cc's): source element offsets of wave-item q for this lane (block-invariant)
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
    auto a_write = [&](int q, uint32_t crow0s, RawRGB v) {      // crow0s: byte offset of the ring row of this trip's first staged row, before the shift
        const int it = min((q << 6) + lane, NB * SWP - 1);
        const int j = it / SWP, i = it - j * SWP;
        if (i >= pad && i < pad + TW) {                    // a centre pixel: parked as packed bytes for the tail (half: its three samples in the raw tile)
            if constexpr (HALF) {
                const uint32_t cb = RING_B + (uint32_t)(j * TW * 8 + (i - pad) * 6);
                LDS_AT(lds_u16_t, cb) = (uint16_t)v.r; LDS_AT(lds_u16_t, cb + 2u) = (uint16_t)v.g; LDS_AT(lds_u16_t, cb + 4u) = (uint16_t)v.b;
            } else
            LDS_AT(lds_u32_t, RING_B + ((crow0s + (uint32_t)(((j + RSH) * TW + (i - pad)) * 4)) & RING_MASK)) = v.r | (v.g << 8) | (v.b << 16);
        }
        float* sp = stg + (j * 3) * SWS + i;
        sp[0] = a1(v.r); sp[SWS] = a1(v.g); sp[2 * SWS] = a1(v.b);
    };
    // -- dword A phase: item = (row j, dword k of the row window); block-invariant per lane: the dword's byte offset in a frame row,
    // its row, and the staging slots of its four bytes (a byte that belongs to no staged sample goes to a pad float behind its plane).
    // Half frames: everything below counts SAMPLES, the unit of four is a qword, and the raw unit goes to the one-trip raw tile.
    using FRaw = std::conditional_t<HALF, uint2, uint32_t>;
    struct FItem { uint32_t ld, j, s[4], rq; };       // rq: (row j + RSH) * 256 + 4 * (its dword of the ring row); half: row j * 512 + 8 * (its qword of the tile row)
    const int kc_lo = (int)(((uint32_t)(x0 - aab) * 3u - a_lo) >> 2), kc_hi = (int)(((uint32_t)(x0 + TW - 1 + aab) * 3u + 2u - a_lo) >> 2);      // window dwords holding centre bytes
    auto f_setup = [&](int q) -> FItem {
        FItem it;
        const int idx = min((q << 6) + lane, NB * ND - 1);      // items past the window's end redo the last one
        const int j = idx / ND, k = idx - j * ND;
        it.ld = a_lo + 4u * (uint32_t)k;
        it.j = (uint32_t)j;
        if constexpr (HALF) it.rq = (uint32_t)(j * TW * 8) + 8u * (uint32_t)((k >= kc_lo && k <= kc_hi) ? k - kc_lo : TW - 1);
        else it.rq = (uint32_t)((j + RSH) * TW * 4) + 4u * (uint32_t)((k >= kc_lo && k <= kc_hi) ? k - kc_lo : TW - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int b = (int)it.ld + e;                  // byte of the frame row
            const int p = b / 3, c = b - 3 * p;            // its pixel and channel
            const int i = p - (x0 - pad) + (c == 0 ? P.ab : (c == 2 ? -P.ab : 0));      // staged pixel i takes R from column x - d and B from x + d (ref:573-575)
            const bool ok = i >= 0 && i < SWP;
            it.s[e] = STG_B + (uint32_t)(((j * 3 + c) * SWS + (ok ? i : SWP + e)) * 4);
        }
        return it;
    };
    auto f_load = [&](const FItem& it, int hb) -> FRaw {
        const int y = min(max(hb + (int)it.j, 0), H - 1);                         // BORDER_REPLICATE
        if constexpr (HALF) return *reinterpret_cast<const uint2*>(F.in + 2u * ((uint32_t)__umul24((uint32_t)y, row_elems) + it.ld));      // frame bytes < 2^32 (launch_rr_group)
        else return *reinterpret_cast<const uint32_t*>(F.in + ((uint32_t)__umul24((uint32_t)y, row_elems) + it.ld));
    };
    auto f_write = [&](const FItem& it, uint32_t crow0s, FRaw d) {
        float o[4];
        if constexpr (HALF) {
            LDS_AT(lds_u64_t, RING_B + it.rq) = (unsigned long long)d.x | ((unsigned long long)d.y << 32);      // the raw window qword
            a1x2(d.x & 0xFFFFu, d.x >> 16, o[0], o[1]);
            a1x2(d.y & 0xFFFFu, d.y >> 16, o[2], o[3]);
        } else {
            LDS_AT(lds_u32_t, RING_B + ((crow0s + it.rq) & RING_MASK)) = d;      // the raw window dword
            a1x2(d & 255u, (d >> 8) & 255u, o[0], o[1]);
            a1x2((d >> 16) & 255u, d >> 24, o[2], o[3]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) LDS_AT(lds_f32_t, it.s[e]) = o[e];
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
        // thread f owns float f of the strip's 192-float interleaved RGB row segment (pixel f / 3, channel f % 3) in the V pass AND
        // in the pointwise tail, eight rows at a time.  FAST: dword A phase + frame-row ring + composite tables; else the byte-wise
        // A phase, the packed-pixel ring and the LUT pair (two gathers and the mask multiply between them).
        auto consumer = [&](auto fast_c) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_c)::value;
            constexpr int NI = FAST ? FO : AO;           // this wave's A items per trip
            const uint32_t tsel = (cmb == P.comp_m1 && P.comp_m1 != P.comp_m0) ? LUT_B + LUT_STRIDE * 4 : LUT_B;      // this float's table (FAST)
            const uint32_t gcol8 = (uint32_t)fcol * 8u, gcol4 = (uint32_t)fcol * 4u;  // its pixel in the vignette / grain tiles
            // pre-warp image out through a buffer resource over this block's row segment: one SGPR descriptor + a 32-bit byte
            // offset per store, and an offset outside the segment — rows above / below it (the first trips' and the last trip's
            // garbage rows), lanes right of the frame (offset pinned out of range) — is DROPPED by the hardware's range check
            const __amdgpu_buffer_rsrc_t pre_rsrc = __builtin_amdgcn_make_buffer_rsrc(O.pre + (size_t)y_begin * (size_t)W * 3u, 0,
                                                                                      (int)((uint32_t)(y_end - y_begin) * (uint32_t)W * 12u), 0x00020000);
            const uint32_t row_b = fin ? (uint32_t)W * 12u : 0u;                     // bytes per pre-warp image row (this lane's stride)
            // the frame's scanline row gains through a buffer resource too: rows outside the frame read as 0 (never consumed)
            const __amdgpu_buffer_rsrc_t scan_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(F.scan_row), 0, H * 4, 0x00020000);
            // this float's centre byte inside a ring row: FAST — its byte of the frame-row window (a2: R from column x - d, B from
            // x + d; the window is inside the frame, no wrap); else byte fch of packed pixel fcol
            const uint32_t cpl = FAST ? ((uint32_t)((x0 + fcol + (fch == 0 ? -P.ab : (fch == 2 ? P.ab : 0))) * 3 + fch) - a_lo - 4u * (uint32_t)kc_lo) * (HALF ? 2u : 1u)
                                      : (HALF ? (uint32_t)(fcol * 6 + fch * 2) : (uint32_t)(fcol * 4 + fch));
            // half frames: the raw centre samples of this float, rows yb .. yb + R + NB - 1 (yb = the next trip's first output row), packed two
            // per VGPR: element i = cw[i >> 1], half i & 1.  A trip takes elements 0 .. 7 (phase 1), moves the rest down by eight and appends
            // the eight rows the A phase has just parked in the raw tile (phase 2).
            constexpr int NW = R + NB;
            uint32_t cw[HALF ? (NW + 1) / 2 : 1];
#pragma unroll
            for (int i = 0; i < (HALF ? (NW + 1) / 2 : 1); ++i) cw[i] = 0u;
            auto cw_take = [&](float (&v)[NB]) {
                if constexpr (HALF) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) v[j] = a1((j & 1) ? cw[j >> 1] >> 16 : cw[j >> 1] & 0xFFFFu);
                }
            };
            auto cw_shift = [&]() {
                if constexpr (HALF) {
#pragma unroll
                    for (int i = 0; i < (R + 1) / 2; ++i) cw[i] = cw[i + NB / 2];
                }
            };
            auto cw_append = [&]() {
                if constexpr (HALF) {
                    uint32_t hs[NB];
#pragma unroll
                    for (int j = 0; j < NB; ++j) hs[j] = (uint32_t)LDS_AT(lds_u16_t, RING_B + (uint32_t)(j * TW * 8) + cpl);
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int i = R + j;
                        if ((i & 1) == 0 && j + 1 < NB) cw[i >> 1] = hs[j] | (hs[j + 1] << 16);
                        else if ((i & 1) == 0) cw[i >> 1] = hs[j];      // the window's last element: its upper half is unused
                        else if (j == 0) cw[i >> 1] = (cw[i >> 1] & 0xFFFFu) | (hs[0] << 16);      // R odd: element R shares its VGPR with element R - 1
                    }
                }
            };
            f32x2 win2[L / 2];
#pragma unroll
            for (int i = 0; i < L / 2; ++i) win2[i] = f32x2{0.0f, 0.0f};
            // C1: append the eight H rows of the tile, form output rows j (x) and j + 1 (y) of each pair from window elements
            // i = j .. j + 2R + 1 oldest first, shift the window down by NB
            auto v_pass = [&](float (&blur)[NB], bool warm) {
                const float* hcol = hrow + f;
#pragma unroll
                for (int j = 0; j < NB; ++j) win2[(2 * R + j) >> 1][j & 1] = hcol[j * CC_HROW];
                f32x2 acc[NB / 2];
#pragma unroll
                for (int jp = 0; jp < NB / 2; ++jp) acc[jp] = f32x2{0.0f, 0.0f};
                if (!warm) {
#pragma unroll
                for (int i = 0; i < L; ++i)
#pragma unroll
                    for (int jp = 0; jp < NB / 2; ++jp) {
                        const int t = i - 2 * jp;
                        if (t == 0) acc[jp].x = fmaf(win2[i >> 1][i & 1], taps[0], acc[jp].x);
                        else if (t >= 1 && t <= 2 * R) PK_TAPS(acc[jp], win2[i >> 1], (i & 1) != 0, t);
                        else if (t == 2 * R + 1) acc[jp].y = fmaf(win2[i >> 1][i & 1], taps[0], acc[jp].y);       // tap[2R] == tap[0]
                    }
                }
#pragma unroll
                for (int jp = 0; jp < NB / 2; ++jp) { blur[2 * jp] = acc[jp].x; blur[2 * jp + 1] = acc[jp].y; }
#pragma unroll
                for (int i = 0; i < R; ++i) win2[i] = win2[i + NB / 2];
            };
            // a1 of the centre samples of output rows c2row0 .. + 7 of the ring (a2 is in the byte's column)
            auto centre = [&](uint32_t c2row0s, float (&v)[NB]) {      // c2row0s: byte offset of the ring row of output row yb (a multiple of 8 rows: no wrap inside)
                uint32_t cb[NB];
                const uint32_t cbase = c2row0s + cpl;
#pragma unroll
                for (int j = 0; j < NB; ++j) cb[j] = (uint32_t)LDS_AT(lds_u8_t, RING_B + (uint32_t)(j * TW * 4) + cbase);
#pragma unroll
                for (int j = 0; j < NB; j += 2) a1x2(cb[j], cb[j + 1], v[j], v[j + 1]);
            };
            // the scanline gains of rows yb .. yb + 7: lane l asks for row yb + (l & 7) — ONE load per wave and trip, one VGPR across
            // the barrier — and the tail reads row j's gain out of lane j (v_readlane: an SGPR operand of its multiply).  The offset
            // rides in the VGPR operand, which the hardware range-checks: out-of-frame rows read 0 and touch no memory.
            const uint32_t lane7x4 = (uint32_t)(lane & 7) * 4u;
            auto scan_load = [&](int yb) -> uint32_t { return __builtin_amdgcn_raw_buffer_load_b32(scan_rsrc, (uint32_t)yb * 4u + lane7x4, 0, 0); };
            // a7 for the eight rows
            auto triad = [&](float (&v)[NB]) {
                if constexpr (FAST) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) v[j] = LDS_AT(lds_f32_t, tsel + ((uint32_t)lut_index_unit(v[j]) << 2));             // ref:250-252 + :261-262 composed
                } else {
#pragma unroll
                    for (int j = 0; j < NB; ++j) v[j] = LDS_AT(lds_f32_t, LUT_B + ((uint32_t)lut_index_unit(v[j]) << 2)) * cm;       // ref:250-252
#pragma unroll
                    for (int j = 0; j < NB; ++j) v[j] = LDS_AT(lds_f32_t, LUT_B + LUT_STRIDE * 4 + ((uint32_t)lut_index(v[j]) << 2));   // ref:261-262
                }
            };
            // ---- A items of this wave ----
            FItem fit[FAST ? FO : 1];
            FRaw fraw[FAST ? FO : 1];
            uint32_t offr[FAST ? 1 : AO], offg[FAST ? 1 : AO], offb[FAST ? 1 : AO];
            RawRGB raw[FAST ? 1 : AO];
            if constexpr (FAST) {
#pragma unroll
                for (int u = 0; u < FO; ++u) fit[u] = f_setup(wave + 3 * u);
            } else {
#pragma unroll
                for (int u = 0; u < AO; ++u) a_offsets(min(wave + 3 * u, NA - A3 - 1), offr[u], offg[u], offb[u]);
            }
            auto prefetch = [&](int hbn) {
                if constexpr (FAST) {
#pragma unroll
                    for (int u = 0; u < FO; ++u) fraw[u] = f_load(fit[u], hbn);
                } else {
#pragma unroll
                    for (int u = 0; u < AO; ++u) raw[u] = a_load(min(wave + 3 * u, NA - A3 - 1), hbn, offr[u], offg[u], offb[u]);   // past the last block: clamped rows, never consumed
                }
            };
            auto stage = [&](uint32_t crow0) {
                if constexpr (FAST) {
#pragma unroll
                    for (int u = 0; u < FO; ++u) f_write(fit[u], crow0, fraw[u]);
                } else {
#pragma unroll
                    for (int u = 0; u < AO; ++u) a_write(min(wave + 3 * u, NA - A3 - 1), crow0, raw[u]);
                }
            };
            (void)NI;
            __syncthreads();                                // tables visible
            prefetch(y_begin - R);
            // eight stores behind the first prefetch, as in every later trip: the loop is entered with the same count of vector
            // memory operations younger than the prefetched data as its back edge carries
#pragma unroll
            for (int j = 0; j < NB; ++j) __builtin_amdgcn_raw_buffer_store_b32(0u, pre_rsrc, 0xFFFFFF00u - 16u * (uint32_t)j, 0, 0);      // out of range: dropped
            CC_PRIO(CC_P_A);
            // ring rows as byte offsets: trip n stages rows hb + j into ring rows (n * NB + j + RSH) & 31 and reads the centre rows of
            // output rows yb + j = hb - NB - R + j from ring rows (n * NB - NB - (R - RSH) + j) & 31
            uint32_t crow0 = 0u, c2row0 = (uint32_t)((-NB - (R - RSH)) * TW * 4) & RING_MASK;
            int hb = y_begin - R;
            uint32_t off0 = fin ? (uint32_t)(-2 * R - NB) * row_b + ((uint32_t)x0 * 3u + (uint32_t)f) * 4u : 0xFFFFFF00u;      // (row hb - NB - R of the segment, float f), modulo 2^32 while that row is above it
            for (int n = 0; n < n_iter; ++n, hb += NB, off0 += (uint32_t)NB * row_b, crow0 = (crow0 + NB * TW * 4) & RING_MASK,
                                            c2row0 = (c2row0 + NB * TW * 4) & RING_MASK) {
                // ---- phase 1 ----
                const int yb = hb - NB - R;                 // first output row of block n-1 (garbage rows in trip 0)
                const uint32_t slv = scan_load(yb);          // in front of the prefetch below: its wait leaves those loads in flight
                // the first trips of a block only fill the row window: all their output rows lie above the segment (their stores
                // are dropped by the buffer's range check), so the taps and the tail are skipped — a wave-uniform test
                const bool warm = CT_WARM_SKIP && yb + NB <= y_begin;
                float v[NB];
                if constexpr (HALF) { if (!warm) cw_take(v); cw_shift(); }
                else if (!warm) centre(c2row0, v);          // issued first: the LDS round trip runs under the V pass
                {
                    float blur[NB];
                    CC_PRIO(CC_P_VH);
                    v_pass(blur, warm);
                    CC_PRIO(CC_P_A);
                    if (!warm) {
#pragma unroll
                        for (int j = 0; j < NB; ++j) v[j] = clip01(v[j] + P.bloom_strength * blur[j]);      // ref:611 — only v[] crosses the barrier
                    } else {
#pragma unroll
                        for (int j = 0; j < NB; ++j) v[j] = 0.0f;
                    }
                }
                STAMP(4);
                stage(crow0);                               // A(n): the data requested one trip ago -> staging tile + ring
                prefetch(hb + NB);
                STAMP(0);
                __syncthreads();
                STAMP(1);
                // ---- phase 2: C2 of block n-1 (output rows yb + j), stage by stage over the eight rows ----
                CC_PRIO(CC_P_C2);
                if constexpr (HALF) cw_append();
                const uint32_t gt_b = GN_B + (uint32_t)(((n & 1) ^ 1) * NB * TW * 4) + gcol4;
                if (!warm) {
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
                }
                {
                    uint32_t boff = off0;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j]), pre_rsrc, boff, 0, 0);
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
                const int yb = hb - NB - R;
                const uint32_t slv = scan_load(yb);
                float v[NB], blur[NB];
                if constexpr (HALF) cw_take(v); else centre(c2row0, v);
                v_pass(blur, false);
                __syncthreads();
                const float* gt = gn + ((n_iter & 1) ^ 1) * NB * TW;
#pragma unroll
                for (int j = 0; j < NB; ++j) v[j] = clip01(v[j] + P.bloom_strength * blur[j]);
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
        if (fast) consumer(std::true_type{});
        else consumer(std::false_type{});
    } else {
        // =============================== HELPER: wave 3 ======================================================================
        // lane = pixel column: the float64 vignette gain tile of block n-1 (phase 1), the grain tile of block n (phase 2), and its
        // share of the A items
        auto helper = [&](auto fast_c) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_c)::value;
            const int xg = x0 + lane;
            const double cnx2 = P.vig_nx2[min(xg, W - 1)];
            constexpr int NH = FAST ? (FH > 0 ? FH : 1) : (A3 > 0 ? A3 : 1);
            FItem fit[FAST ? NH : 1];
            FRaw fraw[FAST ? NH : 1];
            uint32_t offr[FAST ? 1 : NH], offg[FAST ? 1 : NH], offb[FAST ? 1 : NH];
            RawRGB raw[FAST ? 1 : NH];
            if constexpr (FAST) {
#pragma unroll
                for (int u = 0; u < FH; ++u) fit[u] = f_setup(3 * FO + u);
            } else {
#pragma unroll
                for (int u = 0; u < A3; ++u) a_offsets(NA - A3 + u, offr[u], offg[u], offb[u]);
            }
            // a short window (small aberration) leaves the last wave-items wholly past its end: skipped (a wave-uniform test)
            auto live = [&](int u) { return (3 * FO + u) * 64 < NB * ND; };
            auto prefetch = [&](int hbn) {
                if constexpr (FAST) {
#pragma unroll
                    for (int u = 0; u < FH; ++u) if (live(u)) fraw[u] = f_load(fit[u], hbn);
                } else {
#pragma unroll
                    for (int u = 0; u < A3; ++u) raw[u] = a_load(NA - A3 + u, hbn, offr[u], offg[u], offb[u]);
                }
            };
            auto stage = [&](uint32_t crow0) {
                if constexpr (FAST) {
#pragma unroll
                    for (int u = 0; u < FH; ++u) if (live(u)) f_write(fit[u], crow0, fraw[u]);
                } else {
#pragma unroll
                    for (int u = 0; u < A3; ++u) a_write(NA - A3 + u, crow0, raw[u]);
                }
            };
            __syncthreads();
            prefetch(y_begin - R);
            CC_PRIO(CC_P_HELP);
            uint32_t crow0 = 0u;
            int hb = y_begin - R;
            // a9 vignette gain of the 8 x 64 pixels of output rows yb .. yb + 7: ny^2 of a row from 8-byte loads at wave-uniform
            // addresses, the offset in the (range-checked) vector operand; rows outside the frame read 0 (never consumed)
            const __amdgpu_buffer_rsrc_t ny2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P.vig_ny2), 0, H * 8, 0x00020000);
            auto vig_tile = [&](int yb) {
                const uint32_t vo = (uint32_t)yb * 8u;
                uint64_t q[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) q[j] = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(ny2_rsrc, vo + 8u * (uint32_t)j, 0, 0));
#pragma unroll
                for (int j = 0; j < NB; ++j) gvig[j * TW + lane] = vignette_gain(P, cnx2, __builtin_bit_cast(double, q[j]));
            };
            for (int n = 0; n < n_iter; ++n, hb += NB, crow0 = (crow0 + NB * TW * 4) & RING_MASK) {
                // ---- phase 1: a9 vignette gain of block n-1's pixels; its share of A(n) ----
                if (!(CT_WARM_SKIP && hb - R <= y_begin)) vig_tile(hb - NB - R);        // rows above the segment: never consumed
                STAMP(4);
                stage(crow0);
                prefetch(hb + NB);
                STAMP(0);
                __syncthreads();
                STAMP(1);
                // ---- phase 2: a11 grain sample * scale of block n's pixels (consumed next trip) ----
                float* gw = gn + (n & 1) * NB * TW;
                if (!(CT_WARM_SKIP && hb - R + NB <= y_begin)) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int y = min(max(hb - R + j, 0), H - 1);
                        const float z = grain_normal(F.key0, F.key1, (uint32_t)y * (uint32_t)W + (uint32_t)xg);
                        gw[j * TW + lane] = z * P.noise_scale;
                    }
                }
                STAMP(6);
                STAMP(2);
                __syncthreads();
                STAMP(3);
            }
            vig_tile(hb - NB - R);
            __syncthreads();
        };
        if (fast) helper(std::true_type{});
        else helper(std::false_type{});
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
