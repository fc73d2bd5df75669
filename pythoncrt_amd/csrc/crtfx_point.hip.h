// crtfx_point.hip.h — the pointwise chain: k_half(_group), k_point, k_point_sel(_seq), k_point_lean(_seq)
// (one of the parts of crtfx_kernels.hip.h; see the chain overview there and DESIGN.md §3)
#pragma once
#include "crtfx_common.hip.h"

namespace crtfx {

// ---------------------------------------------------------------------------------------
// k_point — bloom off: the chain is pointwise.  One thread per pixel, 4 rows x 64 px per block.
// ---------------------------------------------------------------------------------------
#ifdef CRTFX_MAIN_TU
// Fast bloom (ref:605-607): ds = cv2.resize(src, (W//2, H//2), INTER_LINEAR); blur = cv2.resize(ds, (W, H), INTER_LINEAR).
// k_half writes ds (graded + thresholded source at half resolution) into the ctx scratch P.ds:
//   * exact 2x decimation (W, H even): OpenCV's INTER_AREA fast path, (p00 + p01 + p10 + p11) * 0.25;
//   * otherwise the generic bilinear taps from the dx/dy axis tables.
// SF / PIX: gate word and pixel format folded at compile time for a plain render frame (see k_point_lean), or
// SF = 0xFFFFFFFF for the general build.
// a1..a4 (+ overlay-before) of the pixel whose SAMPLES sit at the already pixelate-mapped (ys, xs); (y, x): the pixel itself (overlay).
__device__ __forceinline__ void fetch_graded_mapped(const KParams& P, const KFrame& F, int ys, int xs, int y, int x, float (&v)[3]) {
    const uint32_t row = (uint32_t)ys * (uint32_t)P.W * 3u;
    int xr = xs, xb = xs;
    if (P.ab != 0) { xr = wrap(xs - P.ab, P.W); xb = wrap(xs + P.ab, P.W); }      // ref:573-575
    const RawRGB raw = load_raw(P.pix, F.in, row + (uint32_t)xr * 3u, row + (uint32_t)xs * 3u + 1u, row + (uint32_t)xb * 3u + 2u);
    if (P.grade_lut && (P.flags & CRTFX_F_GAMMA)) { v[0] = P.grade_lut[raw.r]; v[1] = P.grade_lut[256 + raw.g]; v[2] = P.grade_lut[512 + raw.b]; }
    else { v[0] = norm_px(P.pix, raw.r); v[1] = norm_px(P.pix, raw.g); v[2] = norm_px(P.pix, raw.b); grade(P, v[0], v[1], v[2]); }
    if (F.overlay_before) overlay_blend<float>(F.overlay_before, (uint32_t)y * (uint32_t)P.W + (uint32_t)x, v[0], v[1], v[2]);
}

template <uint32_t SF, int PIX>
__device__ __forceinline__ void half_body(const KParams& Pin, const KFrame& Fin, float* __restrict__ ds) {
    KParams P = Pin;
    KFrame F = Fin;
    if constexpr (SF != 0xFFFFFFFFu) { P.flags = SF; P.pix = PIX; F.overlay_before = nullptr; }
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= P.hw || j >= P.hh) return;
    // the four taps: (x0, y0) (x1, y0) (x0, y1) (x1, y1)
    int x0, x1, y0, y1;
    float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
    const bool mean4 = !P.dx_ofs;
    if (mean4) { x0 = 2 * i; x1 = 2 * i + 1; y0 = 2 * j; y1 = 2 * j + 1; }
    else {
        x0 = P.dx_ofs[i]; y0 = P.dy_ofs[j];
        x1 = min(x0 + 1, P.W - 1); y1 = min(y0 + 1, P.H - 1);
        a1 = P.dx_a[i]; a0 = 1.0f - a1; b1 = P.dy_a[j]; b0 = 1.0f - b1;
    }
    float a[3], b[3], c[3], d[3];
    if (P.flags & CRTFX_F_PIXELATE) {
        // With pixelate on, neighbouring taps usually read the SAME source pixel (pixel size 2, the reference CLI's default: all
        // four of a 2x2 mean) — fetch and grade each distinct one once.  Identical inputs, so the sums below are the same bits.
        const int mx0 = P.xmap[x0], mx1 = P.xmap[x1], my0 = P.ymap[y0], my1 = P.ymap[y1];
        const bool share = !F.overlay_before;       // (an overlay is blended per output pixel: no sharing then)
        const bool same_x = share && mx1 == mx0, same_y = share && my1 == my0;
        fetch_graded_mapped(P, F, my0, mx0, y0, x0, a);
        if (same_x) { b[0] = a[0]; b[1] = a[1]; b[2] = a[2]; } else fetch_graded_mapped(P, F, my0, mx1, y0, x1, b);
        if (same_y) { c[0] = a[0]; c[1] = a[1]; c[2] = a[2]; } else fetch_graded_mapped(P, F, my1, mx0, y1, x0, c);
        if (same_x) { d[0] = c[0]; d[1] = c[1]; d[2] = c[2]; }
        else if (same_y) { d[0] = b[0]; d[1] = b[1]; d[2] = b[2]; }
        else fetch_graded_mapped(P, F, my1, mx1, y1, x1, d);
    } else {
        fetch_graded(P, F, y0, x0, a[0], a[1], a[2]);
        fetch_graded(P, F, y0, x1, b[0], b[1], b[2]);
        fetch_graded(P, F, y1, x0, c[0], c[1], c[2]);
        fetch_graded(P, F, y1, x1, d[0], d[1], d[2]);
    }
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (mean4) o[k] = (((bloom_src(P, a[k]) + bloom_src(P, b[k])) + bloom_src(P, c[k])) + bloom_src(P, d[k])) * 0.25f;
        else o[k] = (bloom_src(P, a[k]) * a0 + bloom_src(P, b[k]) * a1) * b0 + (bloom_src(P, c[k]) * a0 + bloom_src(P, d[k]) * a1) * b1;
    }
    float* q = ds + ((size_t)j * P.hw + i) * 3;
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
}
template <uint32_t SF, int PIX>
__global__ __launch_bounds__(256) void k_half(KParams Pin, KFrame Fin) { half_body<SF, PIX>(Pin, Fin, Pin.ds); }
// the frames of a group, blockIdx.z = frame, each into its own slot of the scratch (slot stride = hh * hw * 3 floats)
template <uint32_t SF, int PIX>
__global__ __launch_bounds__(256) void k_half_group(KParams Pin, KGroup G) {
    half_body<SF, PIX>(Pin, G.f[blockIdx.z], Pin.ds + (size_t)blockIdx.z * ((size_t)Pin.hh * Pin.hw * 3));
}

// k_point — no Gaussian bloom: the chain is pointwise (plus, for fast bloom, a 2x2 gather from the
// half-res image k_half left in P.ds).  One thread per pixel, 4 rows x 64 px per block.
// Block = 64 px x (blockDim.x / 64) rows; the host launches 1024 threads (16 rows) so that the two gamma LUTs
// (8 KB) are staged into LDS once per 1024 pixels.  (A loop over row tiles inside a 256-thread block instead keeps
// the whole kernel-argument block live across the loop: 101 SGPR spills, 88 VGPRs, 43 us instead of 34 at 1080p.)
// SF: the gate word folded at compile time (see k_phosphor_rr), or SF_RUNTIME.
template <uint32_t SF>
__global__ __launch_bounds__(1024) void k_point(KParams Pin, KFrame F, KOut O) {
    __shared__ float lut[2 * LUT_STRIDE];
    KParams P = Pin;
    if constexpr (SF != 0xFFFFFFFFu) P.flags = SF;
    const bool use_lut = (P.flags & CRTFX_F_TRIAD) && (P.flags & CRTFX_F_TRIAD_LUT);
    if (use_lut) {
        for (int i = threadIdx.x; i < LUT_N; i += blockDim.x) { lut[i] = P.lut_g[i]; lut[LUT_STRIDE + i] = P.lut_inv[i]; }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TW;
    const int y = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (y >= P.H) return;                      // whole wavefront exits together
    const int x = x0 + lane;
    const bool live = x < P.W;
    float r = 0, g = 0, b = 0;
    PixMasks M{};
    if (live) {
        M = load_masks(P, F, y, x);
        fetch_graded(P, F, y, x, r, g, b);
        if (P.flags & CRTFX_F_BLOOM_FAST) {
            const int sx = P.ux_ofs[x], sy = P.uy_ofs[y];
            const int sx1 = min(sx + 1, P.hw - 1), sy1 = min(sy + 1, P.hh - 1);
            const float a1 = P.ux_a[x], a0 = 1.0f - a1, b1 = P.uy_a[y], b0 = 1.0f - b1;
            const float* p00 = P.ds + ((size_t)sy * P.hw + sx) * 3;
            const float* p01 = P.ds + ((size_t)sy * P.hw + sx1) * 3;
            const float* p10 = P.ds + ((size_t)sy1 * P.hw + sx) * 3;
            const float* p11 = P.ds + ((size_t)sy1 * P.hw + sx1) * 3;
            const float bl0 = (p00[0] * a0 + p01[0] * a1) * b0 + (p10[0] * a0 + p11[0] * a1) * b1;
            const float bl1 = (p00[1] * a0 + p01[1] * a1) * b0 + (p10[1] * a0 + p11[1] * a1) * b1;
            const float bl2 = (p00[2] * a0 + p01[2] * a1) * b0 + (p10[2] * a0 + p11[2] * a1) * b1;
            r = clip01(r + P.bloom_strength * bl0); g = clip01(g + P.bloom_strength * bl1); b = clip01(b + P.bloom_strength * bl2);   // ref:611
        } else if (P.flags & CRTFX_F_BLOOM) {          // split Gaussian bloom: the blurred plane k_sb_cols left in P.ds
            const F3 bl = *reinterpret_cast<const F3*>(P.ds + ((size_t)y * P.W + x) * 3);
            r = clip01(r + P.bloom_strength * bl.x); g = clip01(g + P.bloom_strength * bl.y); b = clip01(b + P.bloom_strength * bl.z);   // ref:611
        }
    }
    emit_pixel(P, F, O, y, x0, lane, live, M, r, g, b, lut, lut + LUT_STRIDE);
}

// a11 with grain_size > 1 for the branch-free point kernels: the pixel's N(0,1) sample = bilinear upsample (ref:637-642) of the coarse
// plane of hashed normals (or of an injected coarse plane) from tap indices / weights the caller has already loaded.
__device__ __forceinline__ float coarse_grain(const KParams& P, const KFrame& F, int sx, int sy, float a1, float b1) {
    const int sx1 = min(sx + 1, P.gw - 1), sy1 = min(sy + 1, P.gh - 1);
    const float a0 = 1.0f - a1, b0 = 1.0f - b1;
    const uint32_t i00 = (uint32_t)sy * P.gw + sx, i01 = (uint32_t)sy * P.gw + sx1;
    const uint32_t i10 = (uint32_t)sy1 * P.gw + sx, i11 = (uint32_t)sy1 * P.gw + sx1;
    float n00, n01, n10, n11;
    if (F.noise_plane) { n00 = F.noise_plane[i00]; n01 = F.noise_plane[i01]; n10 = F.noise_plane[i10]; n11 = F.noise_plane[i11]; }
    else { n00 = grain_normal(F.key0, F.key1, i00); n01 = grain_normal(F.key0, F.key1, i01);
           n10 = grain_normal(F.key0, F.key1, i10); n11 = grain_normal(F.key0, F.key1, i11); }
    return (n00 * a0 + n01 * a1) * b0 + (n10 * a0 + n11 * a1) * b1;
}

// k_point_sel — the pointwise chain for ANY gate set with the loads made branch-free.  hipcc ends every conditional
// block that contains a load with an s_waitcnt vmcnt(0), so the gate-by-gate k_point above pays one memory round trip
// per enabled stage (eight in a row for the reference CLI's defaults with one knob changed).  Here every stage's
// address is a wave-uniform SELECT between its real table and a small constant buffer (ones / zeros), the loads are
// issued unconditionally in two groups (tables and planes; then the samples and half-res taps that need the index
// tables) and the stage arithmetic is gated afterwards (branches without loads cost nothing).  Same arithmetic, same bits as k_point (test_kernel_variants_agree); grain_size > 1 stays on k_point.
template <typename T>
__device__ __forceinline__ F3 point_finish(const KParams& P, const KFrame& F, const KOut& O, int y, int x, uint32_t pix, bool row_live,
                                             const PixMasks& M, float r, float g, float b, const float* lut, uint32_t ov_after, F3 st,
                                             int x0, int lane) {
    T v0, v1, v2;
    tail_masks<T, false>(P, F, y, x, M, r, g, b, lut, lut + LUT_STRIDE, v0, v1, v2);
    if (O.pre) {                                 // two-kernel path: park the pre-warp pixel for k_warp
        if (row_live) *reinterpret_cast<F3*>(O.pre + pix * 3u) = F3{(float)v0, (float)v1, (float)v2};
        return F3{(float)v0, (float)v1, (float)v2};
    }
    if (O.overlay_after) overlay_blend_px<T>(ov_after, v0, v1, v2);      // the pixel was loaded above; no load inside this branch
    if (O.out_f32 && row_live) *reinterpret_cast<F3*>(O.out_f32 + pix * 3u) = F3{(float)v0, (float)v1, (float)v2};
    const T p = (T)O.p, q = (T)O.q;
    if (O.blend == CRTFX_BLEND_RENDER) {          // ref:1092
        v0 = clip01(p * (T)st.x + q * v0); v1 = clip01(p * (T)st.y + q * v1); v2 = clip01(p * (T)st.z + q * v2);
    } else if (O.blend == CRTFX_BLEND_PREVIEW) {  // ref:693 addWeighted = fma(prev, a, img*b)
        if constexpr (sizeof(T) == 8) { v0 = fma((T)st.x, p, v0 * q); v1 = fma((T)st.y, p, v1 * q); v2 = fma((T)st.z, p, v2 * q); }
        else { v0 = fmaf(st.x, p, v0 * q); v1 = fmaf(st.y, p, v1 * q); v2 = fmaf(st.z, p, v2 * q); }
    }
    const float f0 = (float)v0, f1 = (float)v1, f2 = (float)v2;
    if (O.state && row_live) *reinterpret_cast<F3*>(O.state + pix * 3u) = F3{f0, f1, f2};
    if (O.out_u8 && row_live) {
        PackedPix pk;
        if (O.pix == CRTFX_PIX_F16) { pk.lo = quant_f16(f0) | (quant_f16(f1) << 16); pk.hi = quant_f16(f2); }
        else { pk.lo = quant_u8x3(f0, f1, f2); pk.hi = 0; }
        store_row_pix(O, (size_t)y * P.W + x0, lane, min(64, P.W - x0), pk);
    }
    return F3{f0, f1, f2};
}

// ONE: neither pixelate nor fast bloom is on (the host checks), so no sample address waits for an index-table load and
// both load groups issue as one: a single memory round trip per wavefront (4K split-bloom chain: 104 -> see DESIGN.md).
template <int PIX, bool ONE = false>
__global__ __launch_bounds__(1024) void k_point_sel(KParams Pin, KFrame F, KOut Oin) {
    __shared__ float lut[2 * LUT_STRIDE];
    KParams P = Pin;
    P.pix = PIX; P.grain = 1;
    const bool gr = (Pin.flags & CRTFX_F_NOISE) && Pin.grain > 1;       // coarse grain: the sample is formed here, not in the tail
    KOut O = Oin;
    O.pix = PIX;
    const uint32_t fl = P.flags;
    const float* ones = P.consts;
    const float* zf = P.consts + 4;
    const int* zi = reinterpret_cast<const int*>(zf);
    const double* zd = reinterpret_cast<const double*>(zf);
    const uint32_t* zu = reinterpret_cast<const uint32_t*>(zf);
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TW;
    const int yraw = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool row_live = yraw < P.H;              // wave-uniform; rows past the bottom redo the last row without storing
    const int y = min(yraw, P.H - 1);
    const int x = min(x0 + lane, P.W - 1);         // lanes past the right edge redo the last pixel: same values, same stores
    const uint32_t pix = (uint32_t)y * (uint32_t)P.W + (uint32_t)x;
    // ---- group 1: loads whose addresses need no other load --------------------------------------------------
    const bool pxl = !ONE && (fl & CRTFX_F_PIXELATE) != 0;
    const bool fb = !ONE && (fl & CRTFX_F_BLOOM) && (fl & CRTFX_F_BLOOM_FAST);
    int xm = 0, ym = 0, ux = 0, uy = 0;
    float ua = 0.0f, ub = 0.0f;
    if constexpr (!ONE) {
        xm = *(pxl ? P.xmap + x : zi); ym = *(pxl ? P.ymap + y : zi);
        ux = *(fb ? P.ux_ofs + x : zi); uy = *(fb ? P.uy_ofs + y : zi);
        ua = *(fb ? P.ux_a + x : zf); ub = *(fb ? P.uy_a + y : zf);
    }
    const bool tri = (fl & CRTFX_F_TRIAD) != 0;
    const F3 tm = *reinterpret_cast<const F3*>(tri ? (P.triad_full ? P.triad_full + (size_t)pix * 3 : P.triad_row + x * 3) : ones);
    const float sl = *((fl & CRTFX_F_SCANLINES) ? (F.scan_plane ? F.scan_plane + pix : F.scan_row + y) : ones);
    const bool vg = (fl & CRTFX_F_VIGNETTE) != 0, vfull = vg && P.vig_full != nullptr;
    const double vfv = *(vfull ? P.vig_full + pix : zd);
    const double nx2 = *((vg && !vfull) ? P.vig_nx2 + x : zd), ny2 = *((vg && !vfull) ? P.vig_ny2 + y : zd);
    const uint32_t ov_before = *(F.overlay_before ? reinterpret_cast<const uint32_t*>(F.overlay_before) + pix : zu);
    const uint32_t ov_after = *(O.overlay_after ? reinterpret_cast<const uint32_t*>(O.overlay_after) + pix : zu);
    const float* sin = O.state_in ? O.state_in : O.state;
    const F3 st = *reinterpret_cast<const F3*>((O.blend != CRTFX_BLEND_NONE) ? sin + (size_t)pix * 3 : zf);
    const float zn = *((F.noise_plane && !gr) ? F.noise_plane + pix : zf);
    const int gsx = *(gr ? P.gx_ofs + x : zi), gsy = *(gr ? P.gy_ofs + y : zi);
    const float ga1 = *(gr ? P.gx_a + x : zf), gb1 = *(gr ? P.gy_a + y : zf);
    const bool gb = (fl & CRTFX_F_BLOOM) && !(fl & CRTFX_F_BLOOM_FAST);      // split Gaussian bloom: the blurred plane in P.ds
    const F3 gbl = *reinterpret_cast<const F3*>(gb ? P.ds + (size_t)pix * 3 : zf);
    if ((fl & CRTFX_F_TRIAD) && (fl & CRTFX_F_TRIAD_LUT))
        for (int i = threadIdx.x; i < LUT_N; i += blockDim.x) { lut[i] = P.lut_g[i]; lut[LUT_STRIDE + i] = P.lut_inv[i]; }
    // ---- group 2: the samples (through the pixelate maps) and the half-res taps (through the upsample axes) ------
    const int xs = pxl ? xm : x, ys = pxl ? ym : y;
    int xr = xs, xb = xs;
    if (P.ab != 0) { xr = wrap(xs - P.ab, P.W); xb = wrap(xs + P.ab, P.W); }      // ref:573-575
    const uint32_t row = (uint32_t)ys * (uint32_t)P.W * 3u;
    const RawRGB raw = load_raw(PIX, F.in, row + (uint32_t)xr * 3u, row + (uint32_t)xs * 3u + 1u, row + (uint32_t)xb * 3u + 2u);
    const int hw = fb ? P.hw : 1, hh = fb ? P.hh : 1;
    const float* dsb = fb ? P.ds : zf;
    const int ux1 = min(ux + 1, hw - 1), uy1 = min(uy + 1, hh - 1);
    F3 p00{0, 0, 0}, p01{0, 0, 0}, p10{0, 0, 0}, p11{0, 0, 0};
    if constexpr (!ONE) {
        p00 = *reinterpret_cast<const F3*>(dsb + ((size_t)uy * hw + ux) * 3);
        p01 = *reinterpret_cast<const F3*>(dsb + ((size_t)uy * hw + ux1) * 3);
        p10 = *reinterpret_cast<const F3*>(dsb + ((size_t)uy1 * hw + ux) * 3);
        p11 = *reinterpret_cast<const F3*>(dsb + ((size_t)uy1 * hw + ux1) * 3);
    }
    __syncthreads();                               // LUTs visible (every thread of the block gets here)
    // ---- arithmetic, gated ----------------------------------------------------------------------------------------
    float r, g, b;
    if (P.grade_lut && (fl & CRTFX_F_GAMMA)) { r = P.grade_lut[raw.r]; g = P.grade_lut[256 + raw.g]; b = P.grade_lut[512 + raw.b]; }
    else { r = norm_px(PIX, raw.r); g = norm_px(PIX, raw.g); b = norm_px(PIX, raw.b); grade(P, r, g, b); }
    if (F.overlay_before) overlay_blend_px<float>(ov_before, r, g, b);     // the pixel was loaded above; no load inside this branch
    if (fb) {
        const float a1 = ua, a0 = 1.0f - a1, b1 = ub, b0 = 1.0f - b1;
        const float bl0 = (p00.x * a0 + p01.x * a1) * b0 + (p10.x * a0 + p11.x * a1) * b1;
        const float bl1 = (p00.y * a0 + p01.y * a1) * b0 + (p10.y * a0 + p11.y * a1) * b1;
        const float bl2 = (p00.z * a0 + p01.z * a1) * b0 + (p10.z * a0 + p11.z * a1) * b1;
        r = clip01(r + P.bloom_strength * bl0); g = clip01(g + P.bloom_strength * bl1); b = clip01(b + P.bloom_strength * bl2);   // ref:611
    }
    if (gb) { r = clip01(r + P.bloom_strength * gbl.x); g = clip01(g + P.bloom_strength * gbl.y); b = clip01(b + P.bloom_strength * gbl.z); }   // ref:611
    PixMasks M{tm.x, tm.y, tm.z, sl, vfull ? vfv : vignette_gain(P, nx2, ny2), zn, F.noise_plane != nullptr};
    if (gr) { M.z = coarse_grain(P, F, gsx, gsy, ga1, gb1); M.has_z = 1; }
    if (promotes(P)) point_finish<double>(P, F, O, y, x, pix, row_live, M, r, g, b, lut, ov_after, st, x0, lane);
    else point_finish<float>(P, F, O, y, x, pix, row_live, M, r, g, b, lut, ov_after, st, x0, lane);
}

// k_point_sel_seq — k_point_sel for a RUN of frames (crtfx_process_batch): frames that all blend with their predecessor
// (persistence: the state travels in registers, see k_warp_lean) or that do not blend at all, one after the other in each
// thread; the triad LUTs are staged once, the frame-invariant loads (index maps, mask, vignette, overlays) issue once.
// Frame jf's half-res bloom source sits in slot jf of the scratch (k_half_group).  Same arithmetic per frame as k_point_sel.
template <int PIX, bool ONE>
__global__ __launch_bounds__(1024) void k_point_sel_seq(KParams Pin, KGroup G, int nseq) {
    __shared__ float lut[2 * LUT_STRIDE];
    KParams P = Pin;
    P.pix = PIX; P.grain = 1;
    const bool gr = (Pin.flags & CRTFX_F_NOISE) && Pin.grain > 1;       // coarse grain: the sample is formed here, not in the tail
    const uint32_t fl = P.flags;
    const float* ones = P.consts;
    const float* zf = P.consts + 4;
    const int* zi = reinterpret_cast<const int*>(zf);
    const double* zd = reinterpret_cast<const double*>(zf);
    const uint32_t* zu = reinterpret_cast<const uint32_t*>(zf);
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TW;
    const int yraw = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool row_live = yraw < P.H;              // wave-uniform; rows past the bottom redo the last row without storing
    const int y = min(yraw, P.H - 1);
    const int x = min(x0 + lane, P.W - 1);         // lanes past the right edge redo the last pixel: same values, same stores
    const uint32_t pix = (uint32_t)y * (uint32_t)P.W + (uint32_t)x;
    if ((fl & CRTFX_F_TRIAD) && (fl & CRTFX_F_TRIAD_LUT))
        for (int i = threadIdx.x; i < LUT_N; i += blockDim.x) { lut[i] = P.lut_g[i]; lut[LUT_STRIDE + i] = P.lut_inv[i]; }
    // ---- frame-invariant loads -------------------------------------------------------------------------------------
    const bool pxl = !ONE && (fl & CRTFX_F_PIXELATE) != 0;
    const bool fb = !ONE && (fl & CRTFX_F_BLOOM) && (fl & CRTFX_F_BLOOM_FAST);
    int xm = 0, ym = 0, ux = 0, uy = 0;
    float ua = 0.0f, ub = 0.0f;
    if constexpr (!ONE) {
        xm = *(pxl ? P.xmap + x : zi); ym = *(pxl ? P.ymap + y : zi);
        ux = *(fb ? P.ux_ofs + x : zi); uy = *(fb ? P.uy_ofs + y : zi);
        ua = *(fb ? P.ux_a + x : zf); ub = *(fb ? P.uy_a + y : zf);
    }
    const bool tri = (fl & CRTFX_F_TRIAD) != 0;
    const F3 tm = *reinterpret_cast<const F3*>(tri ? (P.triad_full ? P.triad_full + (size_t)pix * 3 : P.triad_row + x * 3) : ones);
    const bool vg = (fl & CRTFX_F_VIGNETTE) != 0, vfull = vg && P.vig_full != nullptr;
    const double vfv = *(vfull ? P.vig_full + pix : zd);
    const double nx2 = *((vg && !vfull) ? P.vig_nx2 + x : zd), ny2 = *((vg && !vfull) ? P.vig_ny2 + y : zd);
    const int gsx = *(gr ? P.gx_ofs + x : zi), gsy = *(gr ? P.gy_ofs + y : zi);
    const float ga1 = *(gr ? P.gx_a + x : zf), gb1 = *(gr ? P.gy_a + y : zf);
    const KOut O0 = G.o[0];
    const float* sin0 = O0.state_in ? O0.state_in : O0.state;
    F3 st = *reinterpret_cast<const F3*>((O0.blend != CRTFX_BLEND_NONE) ? sin0 + (size_t)pix * 3 : zf);
    const int xs = pxl ? xm : x, ys = pxl ? ym : y;
    int xr = xs, xb = xs;
    if (P.ab != 0) { xr = wrap(xs - P.ab, P.W); xb = wrap(xs + P.ab, P.W); }      // ref:573-575
    const uint32_t row = (uint32_t)ys * (uint32_t)P.W * 3u;
    const int hw = fb ? P.hw : 1, hh = fb ? P.hh : 1;
    const int ux1 = min(ux + 1, hw - 1), uy1 = min(uy + 1, hh - 1);
    const double vgain = vfull ? vfv : vignette_gain(P, nx2, ny2);
    const size_t slot = (size_t)P.hh * P.hw * 3;
    __syncthreads();                               // LUTs visible
    for (int jf = 0; jf < nseq; ++jf) {
        const KFrame F = G.f[jf];                  // wave-uniform index: scalar loads
        KOut O = G.o[jf];
        O.pix = PIX;
        const bool chain = O.blend == CRTFX_BLEND_RENDER;
        const bool keep_state = !chain || jf == nseq - 1 || G.o[jf + 1].state != O.state;
        // ---- this frame's loads: one group -------------------------------------------------------------------------
        const float sl = *((fl & CRTFX_F_SCANLINES) ? (F.scan_plane ? F.scan_plane + pix : F.scan_row + y) : ones);
        const uint32_t ov_before = *(F.overlay_before ? reinterpret_cast<const uint32_t*>(F.overlay_before) + pix : zu);
        const uint32_t ov_after = *(O.overlay_after ? reinterpret_cast<const uint32_t*>(O.overlay_after) + pix : zu);
        const float zn = *((F.noise_plane && !gr) ? F.noise_plane + pix : zf);
        const RawRGB raw = load_raw(PIX, F.in, row + (uint32_t)xr * 3u, row + (uint32_t)xs * 3u + 1u, row + (uint32_t)xb * 3u + 2u);
        F3 p00{0, 0, 0}, p01{0, 0, 0}, p10{0, 0, 0}, p11{0, 0, 0};
        if constexpr (!ONE) {
            const float* dsb = fb ? P.ds + (size_t)jf * slot : zf;
            p00 = *reinterpret_cast<const F3*>(dsb + ((size_t)uy * hw + ux) * 3);
            p01 = *reinterpret_cast<const F3*>(dsb + ((size_t)uy * hw + ux1) * 3);
            p10 = *reinterpret_cast<const F3*>(dsb + ((size_t)uy1 * hw + ux) * 3);
            p11 = *reinterpret_cast<const F3*>(dsb + ((size_t)uy1 * hw + ux1) * 3);
        }
        // ---- arithmetic, gated ---------------------------------------------------------------------------------------
        float r, g, b;
        if (P.grade_lut && (fl & CRTFX_F_GAMMA)) { r = P.grade_lut[raw.r]; g = P.grade_lut[256 + raw.g]; b = P.grade_lut[512 + raw.b]; }
        else { r = norm_px(PIX, raw.r); g = norm_px(PIX, raw.g); b = norm_px(PIX, raw.b); grade(P, r, g, b); }
        if (F.overlay_before) overlay_blend_px<float>(ov_before, r, g, b);
        if (fb) {
            const float a1 = ua, a0 = 1.0f - a1, b1 = ub, b0 = 1.0f - b1;
            const float bl0 = (p00.x * a0 + p01.x * a1) * b0 + (p10.x * a0 + p11.x * a1) * b1;
            const float bl1 = (p00.y * a0 + p01.y * a1) * b0 + (p10.y * a0 + p11.y * a1) * b1;
            const float bl2 = (p00.z * a0 + p01.z * a1) * b0 + (p10.z * a0 + p11.z * a1) * b1;
            r = clip01(r + P.bloom_strength * bl0); g = clip01(g + P.bloom_strength * bl1); b = clip01(b + P.bloom_strength * bl2);   // ref:611
        }
        PixMasks M{tm.x, tm.y, tm.z, sl, vgain, zn, F.noise_plane != nullptr};
        if (gr) { M.z = coarse_grain(P, F, gsx, gsy, ga1, gb1); M.has_z = 1; }
        KOut Ow = O;
        if (!keep_state) Ow.state = nullptr;       // the next frame of the run takes the state from this thread's registers
        F3 fin;
        if (promotes(P)) fin = point_finish<double>(P, F, Ow, y, x, pix, row_live, M, r, g, b, lut, ov_after, st, x0, lane);
        else fin = point_finish<float>(P, F, Ow, y, x, pix, row_live, M, r, g, b, lut, ov_after, st, x0, lane);
        if (chain) st = fin;
    }
}

// k_point_lean — k_point for a plain render frame: gate word, pixel format and blend mode are compile-time, no
// per-pixel planes, overlays or float output (the host checks).  With every gate folded the body is one
// basic block: the index-table loads, then the byte / half-res / mask / state loads issue together instead of one
// memory round trip per stage (the general k_point waits at every branch that contains a load: ~5 dependent round
// trips per wavefront made the 1080p reference-CLI-default chain latency-bound at 33 us).
#ifndef CRTFX_POINT_ROWS
#define CRTFX_POINT_ROWS 2      // output rows per k_point_lean thread (rows y, y + waves): their load chains interleave
#endif
template <uint32_t SF, int PIX, int BLENDM>
__global__ __launch_bounds__(1024) void k_point_lean(KParams Pin, KFrame Fin, KOut Oin) {
    __shared__ float lut[2 * LUT_STRIDE];
    constexpr int ROWS = CRTFX_POINT_ROWS;
    KParams P = Pin;
    P.flags = SF; P.pix = PIX; P.triad_full = nullptr; P.vig_full = nullptr; P.grain = 1;
    KFrame F = Fin;
    F.scan_plane = nullptr; F.noise_plane = nullptr; F.overlay_before = nullptr;
    KOut O = Oin;
    O.blend = BLENDM; O.overlay_after = nullptr; O.out_f32 = nullptr; O.pix = PIX;
    if constexpr ((SF & CRTFX_F_TRIAD) && (SF & CRTFX_F_TRIAD_LUT)) {
        for (int i = threadIdx.x; i < LUT_N; i += blockDim.x) { lut[i] = P.lut_g[i]; lut[LUT_STRIDE + i] = P.lut_inv[i]; }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TW;
    const int waves = blockDim.x >> 6;
    const int ybase = blockIdx.y * (waves * ROWS) + (threadIdx.x >> 6);
    if (ybase >= P.H) return;
    const int x = min(x0 + lane, P.W - 1);       // lanes past the right edge redo the last pixel: same values, same stores
    using T = typename std::conditional<(SF & (CRTFX_F_VIGNETTE | CRTFX_F_FLICKER)) != 0, double, float>::type;
    T v[ROWS][3];
    int yr[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int y = yr[k] = min(ybase + k * waves, P.H - 1);     // a row past the bottom redoes the last one; its stores are skipped
        const PixMasks M = load_masks(P, F, y, x);
        float r, g, b;
        fetch_graded(P, F, y, x, r, g, b);
        if constexpr ((SF & CRTFX_F_BLOOM_FAST) != 0) {
            const int sx = P.ux_ofs[x], sy = P.uy_ofs[y];
            const int sx1 = min(sx + 1, P.hw - 1), sy1 = min(sy + 1, P.hh - 1);
            const float a1 = P.ux_a[x], a0 = 1.0f - a1, b1 = P.uy_a[y], b0 = 1.0f - b1;
            const F3 p00 = *reinterpret_cast<const F3*>(P.ds + ((size_t)sy * P.hw + sx) * 3);
            const F3 p01 = *reinterpret_cast<const F3*>(P.ds + ((size_t)sy * P.hw + sx1) * 3);
            const F3 p10 = *reinterpret_cast<const F3*>(P.ds + ((size_t)sy1 * P.hw + sx) * 3);
            const F3 p11 = *reinterpret_cast<const F3*>(P.ds + ((size_t)sy1 * P.hw + sx1) * 3);
            const float bl0 = (p00.x * a0 + p01.x * a1) * b0 + (p10.x * a0 + p11.x * a1) * b1;
            const float bl1 = (p00.y * a0 + p01.y * a1) * b0 + (p10.y * a0 + p11.y * a1) * b1;
            const float bl2 = (p00.z * a0 + p01.z * a1) * b0 + (p10.z * a0 + p11.z * a1) * b1;
            r = clip01(r + P.bloom_strength * bl0); g = clip01(g + P.bloom_strength * bl1); b = clip01(b + P.bloom_strength * bl2);   // ref:611
        }
        tail_masks<T, false>(P, F, y, x, M, r, g, b, lut, lut + LUT_STRIDE, v[k][0], v[k][1], v[k][2]);
    }
    if (O.pre) {                                 // two-kernel path: park the pre-warp pixels for k_warp
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
            if (ybase + k * waves < P.H)
                *reinterpret_cast<F3*>(O.pre + ((uint32_t)yr[k] * (uint32_t)P.W + (uint32_t)x) * 3u) = F3{(float)v[k][0], (float)v[k][1], (float)v[k][2]};
        return;
    }
    if (ybase + (ROWS - 1) * waves < P.H) {      // every row of this wave is inside the frame (wave-uniform): one block for all commits
        PackedPix pk[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
            pk[k] = commit_pixel<T, true>(O, (uint32_t)yr[k] * (uint32_t)P.W + (uint32_t)x, v[k][0], v[k][1], v[k][2]);
        if (O.out_u8) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) store_row_pix(O, (size_t)yr[k] * P.W + x0, lane, min(64, P.W - x0), pk[k]);
        }
    } else {                                     // bottom edge: only the rows that exist are committed (the state must be blended once)
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
            if (ybase + k * waves < P.H) {
                const PackedPix pk = commit_pixel<T, true>(O, (uint32_t)yr[k] * (uint32_t)P.W + (uint32_t)x, v[k][0], v[k][1], v[k][2]);
                if (O.out_u8) store_row_pix(O, (size_t)yr[k] * P.W + x0, lane, min(64, P.W - x0), pk);
            }
    }
}

// k_point_lean_seq — the persistence chain of the pointwise render chain (no warp behind it; the reference CLI's defaults:
// fast bloom, persistence 0.2, ref:1171-1191): the nseq frames of G one after the other in each thread, its pixels' state in
// registers (see k_warp_lean): the float32 state is read for the first frame and written behind the last only (or behind
// every frame whose record names a state buffer of its own), and the upsample taps' indices and weights are computed
// once.  Frame jf's half-res bloom source sits in slot jf of the scratch (k_half_group).  Same operations per pixel in
// the same order as k_point_lean<SF, PIX, CRTFX_BLEND_RENDER> frame by frame: the same bits.
// BLENDM = CRTFX_BLEND_NONE: the same grouping for independent frames (persistence 0): no state, the rest as above.
// SF = SF_LEAN_RT (round 6): the same kernel with the gate word left at run time (wave-uniform branches) — every setting of the reference CLI
// that needs no per-pixel plane: a colour grade, a bloom threshold, any of triad / scanlines / vignette / grain off, flicker, preserve-luma.
// One knob away from the defaults used to mean the general k_point_sel_seq at half the frame rate (profiles/r06_cli_variants.txt).
template <uint32_t SF>
__device__ __forceinline__ uint32_t lean_flags(const KParams& Pin) {
    if constexpr (SF == SF_LEAN_RT) return Pin.flags & ~(uint32_t)CRTFX_F_WARP;
    else if constexpr ((SF & KF_GRADE_RT) != 0) return (SF & ~KF_GRADE_RT) | (Pin.flags & GRADE_RT_MASK);
    // the bloom threshold (ref:602-604) only ever acts on the bloom SOURCE — k_half's planes, k_point_fused_seq's prologue — never inside the frame
    // loop: it stays a run-time bit in every folded build at the price of one scalar branch per half-resolution entry
    else return (SF & ~(KF_GRADE_LUT | KF_COARSE | KF_SCANPLANE)) | (Pin.flags & CRTFX_F_BLOOM_THR);
}
template <uint32_t SF>
constexpr bool lean_scanplane() { return SF != SF_LEAN_RT && (SF & KF_SCANPLANE) != 0; }
template <uint32_t SF>
constexpr bool lean_coarse() { return SF != SF_LEAN_RT && (SF & KF_COARSE) != 0; }
// a1 + a4 of one pixel of a lean build from its raw samples: the grade table staged in LDS (KF_GRADE_LUT builds: uint8 samples), or normalise + grade
template <uint32_t SF, int PIX>
__device__ __forceinline__ void lean_graded(const KParams& P, const float* glut, const RawRGB& raw, float& r, float& g, float& b) {
    if constexpr (SF != SF_LEAN_RT && (SF & KF_GRADE_LUT) != 0) { r = glut[raw.r]; g = glut[256 + raw.g]; b = glut[512 + raw.b]; }
    else if (P.grade_lut && (P.flags & CRTFX_F_GAMMA)) { r = P.grade_lut[raw.r]; g = P.grade_lut[256 + raw.g]; b = P.grade_lut[512 + raw.b]; }
    else { r = norm_px(PIX, raw.r); g = norm_px(PIX, raw.g); b = norm_px(PIX, raw.b); grade(P, r, g, b); }
}
template <uint32_t SF>
constexpr int lean_glut_floats() { return (SF != SF_LEAN_RT && (SF & KF_GRADE_LUT) != 0) ? 768 : 1; }

template <uint32_t SF, int PIX, int BLENDM>
__global__ __launch_bounds__(1024) void k_point_lean_seq(KParams Pin, KGroup G, int nseq) {
    __shared__ float lut[2 * LUT_STRIDE];
    __shared__ float glut[lean_glut_floats<SF>()];
    constexpr int ROWS = CRTFX_POINT_ROWS;
    KParams P = Pin;
    P.flags = lean_flags<SF>(Pin); P.pix = PIX; P.triad_full = nullptr; P.vig_full = nullptr; P.grain = 1;
    const bool fastb = (P.flags & CRTFX_F_BLOOM) && (P.flags & CRTFX_F_BLOOM_FAST);      // (compile-time in the folded builds)
    if constexpr (lean_glut_floats<SF>() > 1) { for (int i = threadIdx.x; i < 768; i += blockDim.x) glut[i] = P.grade_lut[i]; }
    if (((P.flags & CRTFX_F_TRIAD) && (P.flags & CRTFX_F_TRIAD_LUT)) || lean_glut_floats<SF>() > 1) {
        if ((P.flags & CRTFX_F_TRIAD) && (P.flags & CRTFX_F_TRIAD_LUT))
            for (int i = threadIdx.x; i < LUT_N; i += blockDim.x) { lut[i] = P.lut_g[i]; lut[LUT_STRIDE + i] = P.lut_inv[i]; }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TW;
    const int waves = blockDim.x >> 6;
    const int ybase = blockIdx.y * (waves * ROWS) + (threadIdx.x >> 6);
    if (ybase >= P.H) return;
    const int x = min(x0 + lane, P.W - 1);       // lanes past the right edge redo the last pixel: same values, same stores
    int yr[ROWS];
    uint32_t o00[ROWS], o01[ROWS], o10[ROWS], o11[ROWS];      // BYTE offsets of the four half-res taps inside a slot (32-bit: the loads take scalar base + vector offset)
    uint32_t er[ROWS], eg[ROWS], eb[ROWS];                    // element offsets of the pixel's three samples inside a frame (pixelate map and aberration wrap resolved once)
    float a0[ROWS], a1[ROWS], b0[ROWS], b1[ROWS];
    F3 st[ROWS];
    PixMasks M0[ROWS];                           // triad mask and vignette gain of the pixel: frame-invariant (the scanline gain is not)
    const float* state_in = G.o[0].state_in ? G.o[0].state_in : G.o[0].state;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int y = yr[k] = min(ybase + k * waves, P.H - 1);     // a row past the bottom redoes the last one; its stores are skipped
        {
            KFrame F0 = G.f[0];
            F0.scan_plane = nullptr;
            M0[k] = load_masks(P, F0, y, x);
        }
        o00[k] = o01[k] = o10[k] = o11[k] = 0u; a0[k] = a1[k] = b0[k] = b1[k] = 0.0f;
        if (fastb) {
            const int sx = P.ux_ofs[x], sy = P.uy_ofs[y];
            const int sx1 = min(sx + 1, P.hw - 1), sy1 = min(sy + 1, P.hh - 1);
            a1[k] = P.ux_a[x]; a0[k] = 1.0f - a1[k]; b1[k] = P.uy_a[y]; b0[k] = 1.0f - b1[k];
            o00[k] = (uint32_t)(sy * P.hw + sx) * 12u; o01[k] = (uint32_t)(sy * P.hw + sx1) * 12u;
            o10[k] = (uint32_t)(sy1 * P.hw + sx) * 12u; o11[k] = (uint32_t)(sy1 * P.hw + sx1) * 12u;
        }
        {   // = fetch_raw's addressing (ref:573-583), frame-invariant
            int xs = x, ys = y;
            if (P.flags & CRTFX_F_PIXELATE) { xs = P.xmap[x]; ys = P.ymap[y]; }
            const uint32_t row = (uint32_t)ys * (uint32_t)P.W * 3u;
            int xr = xs, xb = xs;
            if (P.ab != 0) { xr = wrap(xs - P.ab, P.W); xb = wrap(xs + P.ab, P.W); }
            er[k] = row + (uint32_t)xr * 3u; eg[k] = row + (uint32_t)xs * 3u + 1u; eb[k] = row + (uint32_t)xb * 3u + 2u;
        }
        if constexpr (BLENDM == CRTFX_BLEND_RENDER) st[k] = *reinterpret_cast<const F3*>(state_in + ((uint32_t)y * (uint32_t)P.W + (uint32_t)x) * 3u);
    }
    const size_t slot = (size_t)P.hh * P.hw * 3;
    // the frames of the run, in the image type the reference has at this point: float64 once the vignette / flicker promotes (ref:626-633)
    auto run = [&](auto tzero) {
        using T = decltype(tzero);
        for (int jf = 0; jf < nseq; ++jf) {
            KFrame F = G.f[jf];                        // wave-uniform index: scalar loads
            F.scan_plane = nullptr; F.noise_plane = nullptr; F.overlay_before = nullptr;
            KOut O = G.o[jf];
            O.pix = PIX;
            const bool keep_state = jf == nseq - 1 || G.o[jf + 1].state != O.state;
            const float* __restrict__ ds = P.ds + (size_t)jf * slot;
            T v[ROWS][3];
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const int y = yr[k];
                PixMasks M = M0[k];
                if (P.flags & CRTFX_F_SCANLINES) M.sl = F.scan_row[y];
                float r, g, b;
                lean_graded<SF, PIX>(P, glut, load_raw(PIX, F.in, er[k], eg[k], eb[k]), r, g, b);      // = fetch_graded (no overlay in the lean build)
                if (fastb) {
                    const char* dsb = reinterpret_cast<const char*>(ds);
                    const F3 p00 = *reinterpret_cast<const F3*>(dsb + o00[k]);
                    const F3 p01 = *reinterpret_cast<const F3*>(dsb + o01[k]);
                    const F3 p10 = *reinterpret_cast<const F3*>(dsb + o10[k]);
                    const F3 p11 = *reinterpret_cast<const F3*>(dsb + o11[k]);
                    const float bl0 = (p00.x * a0[k] + p01.x * a1[k]) * b0[k] + (p10.x * a0[k] + p11.x * a1[k]) * b1[k];
                    const float bl1 = (p00.y * a0[k] + p01.y * a1[k]) * b0[k] + (p10.y * a0[k] + p11.y * a1[k]) * b1[k];
                    const float bl2 = (p00.z * a0[k] + p01.z * a1[k]) * b0[k] + (p10.z * a0[k] + p11.z * a1[k]) * b1[k];
                    r = clip01(r + P.bloom_strength * bl0); g = clip01(g + P.bloom_strength * bl1); b = clip01(b + P.bloom_strength * bl2);   // ref:611
                }
                tail_masks<T, false>(P, F, y, x, M, r, g, b, lut, lut + LUT_STRIDE, v[k][0], v[k][1], v[k][2]);
            }
            const T p = (T)O.p, q = (T)O.q;
            if (O.pre) {                                 // a warp follows: park the pre-warp pixels of this frame for k_warp_lean
#pragma unroll
                for (int k = 0; k < ROWS; ++k)
                    if (ybase + k * waves < P.H)
                        *reinterpret_cast<F3*>(O.pre + ((uint32_t)yr[k] * (uint32_t)P.W + (uint32_t)x) * 3u) = F3{(float)v[k][0], (float)v[k][1], (float)v[k][2]};
                continue;
            }
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                if (ybase + k * waves < P.H) {           // wave-uniform
                    const uint32_t pix = (uint32_t)yr[k] * (uint32_t)P.W + (uint32_t)x;
                    float f0, f1, f2;
                    if constexpr (BLENDM == CRTFX_BLEND_RENDER) {
                        f0 = (float)clip01(p * (T)st[k].x + q * v[k][0]);      // ref:1092
                        f1 = (float)clip01(p * (T)st[k].y + q * v[k][1]);
                        f2 = (float)clip01(p * (T)st[k].z + q * v[k][2]);
                        st[k] = F3{f0, f1, f2};
                    } else { f0 = (float)v[k][0]; f1 = (float)v[k][1]; f2 = (float)v[k][2]; }
                    if (O.state && (keep_state || BLENDM != CRTFX_BLEND_RENDER)) { float* sp = O.state + pix * 3u; sp[0] = f0; sp[1] = f1; sp[2] = f2; }
                    if (O.out_u8) {
                        PackedPix pk;
                        if constexpr (PIX == CRTFX_PIX_F16) { pk.lo = quant_f16(f0) | (quant_f16(f1) << 16); pk.hi = quant_f16(f2); }
                        else { pk.lo = quant_u8x3(f0, f1, f2); pk.hi = 0; }
                        store_row_pix(O, (size_t)yr[k] * P.W + x0, lane, min(64, P.W - x0), pk);
                    }
                }
            }
        }
    };
    if constexpr (SF == SF_LEAN_RT) { if (promotes(P)) run(0.0); else run(0.0f); }
    else if constexpr ((SF & (CRTFX_F_VIGNETTE | CRTFX_F_FLICKER)) != 0) run(0.0);
    else run(0.0f);
}

// k_point_fused_seq — k_point_lean_seq with the fast-bloom source formed INSIDE the kernel (round 6): no k_half_group launch in front of it, no
// quarter-size plane in memory, the frame fetched once.  (ref:605-607: ds = cv2.resize(src, (W//2, H//2)); blur = cv2.resize(ds, (W, H)).)
// A block owns a 64 x (ROWS * waves)-pixel tile.  The bilinear 2x upsample of its pixels reads half-resolution columns x0/2 - 1 .. x0/2 + 32 and rows
// y0/2 - 1 .. y0/2 + waves * ROWS / 2: a (34 x (waves + 2))-entry tile of ds.  The block's first threads form one entry each — the 2 x 2 mean
// of the graded, thresholded source pixels, half_body's arithmetic in half_body's order (with --pixel-size 2, the reference CLI's default, the four
// samples of a cell are one pixel: fetched and graded once) — for ALL nseq frames of the run, in a prologue, into nseq LDS tiles (float4 per entry;
// dynamic LDS, nseq * 34 * (waves + 2) * 16 bytes: 43.5 KB for 8 frames and 8 wavefronts), then ONE barrier; the frame loop behind it is
// k_point_lean_seq's, barrier-free, with the four taps of a pixel read from LDS instead of from the plane.  (The first build formed frame j + 1's tile
// beside frame j's pixels, two tiles, one barrier per frame: 105 us per 8 1080p frames against 76 + 27 for the two launches — the per-frame barrier
// cost more than the launch it saved, profiles/r06_fused_half_ab.txt.)  Exact 2x decimation only (W and H even: the host checks); same bits as the
// two-kernel path.  SF = SF_LEAN_RT: the gate word at run time, as in k_point_lean_seq (the host launches this kernel only with fast bloom on).
constexpr int FUSED_TWH = TW / 2 + 2;                       // 34 half-resolution columns per 64-pixel tile
template <uint32_t SF, int PIX, int BLENDM>
__global__ __launch_bounds__(1024) void k_point_fused_seq(KParams Pin, KGroup G, int nseq) {
    static_assert((SF == SF_LEAN_RT || (SF & CRTFX_F_BLOOM_FAST) != 0) && CRTFX_POINT_ROWS == 2, "the fused build is the fast-bloom chain, two rows per thread");
    __shared__ float lut[2 * LUT_STRIDE];
    __shared__ float glut[lean_glut_floats<SF>()];
    extern __shared__ float4 dst[];                // [nseq][34 * (waves + 2)]
    constexpr int ROWS = CRTFX_POINT_ROWS;
    KParams P = Pin;
    P.flags = lean_flags<SF>(Pin); P.pix = PIX; P.triad_full = nullptr; P.vig_full = nullptr;
    if constexpr (!lean_coarse<SF>()) P.grain = 1;
    if ((P.flags & CRTFX_F_TRIAD) && (P.flags & CRTFX_F_TRIAD_LUT)) {
        for (int i = threadIdx.x; i < LUT_N; i += blockDim.x) { lut[i] = P.lut_g[i]; lut[LUT_STRIDE + i] = P.lut_inv[i]; }
    }
    if constexpr (lean_glut_floats<SF>() > 1) {
        // the grade table feeds the prologue already: staged and made visible first (a barrier of its own; the tile barrier below stays the only
        // one of the frame loop's)
        for (int i = threadIdx.x; i < 768; i += blockDim.x) glut[i] = P.grade_lut[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TW;
    const int waves = blockDim.x >> 6;
    const int y0 = blockIdx.y * (waves * ROWS);
    const int ybase = y0 + (threadIdx.x >> 6);     // (a wave whose rows all lie below the frame stays: it takes part in the barrier and stores nothing)
    const int x = min(x0 + lane, P.W - 1);       // lanes past the right edge redo the last pixel: same values, same stores
    // ---- the entry of the half-resolution tile this thread forms (frame-invariant addressing) ----
    const int i0 = (x0 >> 1) - 1, j0 = (y0 >> 1) - 1;
    const int thh = (waves * ROWS) / 2 + 2;
    const int nent = FUSED_TWH * thh;
    const bool maker = (int)threadIdx.x < nent;
    uint32_t ea[6] = {0, 0, 0, 0, 0, 0};           // element offsets inside a row: R, G, B of the cell's left pixel, then of its right pixel
    uint32_t erow0 = 0, erow1 = 0;                 // element offsets of the cell's two rows
    bool same_x = false, same_y = false;
    if (maker) {
        const int ii = (int)threadIdx.x % FUSED_TWH, jj = (int)threadIdx.x / FUSED_TWH;
        const int i = min(max(i0 + ii, 0), P.hw - 1), j = min(max(j0 + jj, 0), P.hh - 1);
        int mx0 = 2 * i, mx1 = 2 * i + 1, my0 = 2 * j, my1 = 2 * j + 1;
        if (P.flags & CRTFX_F_PIXELATE) {
            mx0 = P.xmap[mx0]; mx1 = P.xmap[mx1]; my0 = P.ymap[my0]; my1 = P.ymap[my1];
            same_x = mx1 == mx0; same_y = my1 == my0;       // pixel size 2: all four samples of the cell are one pixel
        }
        erow0 = (uint32_t)my0 * (uint32_t)P.W * 3u; erow1 = (uint32_t)my1 * (uint32_t)P.W * 3u;
        int xr0 = mx0, xb0 = mx0, xr1 = mx1, xb1 = mx1;
        if (P.ab != 0) { xr0 = wrap(mx0 - P.ab, P.W); xb0 = wrap(mx0 + P.ab, P.W); xr1 = wrap(mx1 - P.ab, P.W); xb1 = wrap(mx1 + P.ab, P.W); }      // ref:573-575
        ea[0] = (uint32_t)xr0 * 3u; ea[1] = (uint32_t)mx0 * 3u + 1u; ea[2] = (uint32_t)xb0 * 3u + 2u;
        ea[3] = (uint32_t)xr1 * 3u; ea[4] = (uint32_t)mx1 * 3u + 1u; ea[5] = (uint32_t)xb1 * 3u + 2u;
    }
    // a1..a4 of one source pixel from its raw samples (= fetch_graded_mapped without an overlay: the lean build has none)
    auto graded = [&](const RawRGB& raw, float (&v)[3]) { lean_graded<SF, PIX>(P, glut, raw, v[0], v[1], v[2]); };
    // the entry from its four graded pixels: half_body's mean4 form, its operation order (bloom source ref:601-604)
    auto put_entry = [&](int jf, const float (&a)[3], const float (&b)[3], const float (&c)[3], const float (&d)[3]) {
        float o[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = (((bloom_src(P, a[k]) + bloom_src(P, b[k])) + bloom_src(P, c[k])) + bloom_src(P, d[k])) * 0.25f;
        if (jf < nseq) dst[jf * nent + (int)threadIdx.x] = float4{o[0], o[1], o[2], 0.0f};
    };
    // The loads of ALL frames are issued before the first is consumed (a frame past the run's end reads the last frame again and stores nothing):
    // taken frame by frame, behind `if (jf < nseq)`, the prologue was eight dependent round trips to HBM — 16 of the kernel's 95 us per 8 1080p
    // frames (profiles/r06_fused_half_ab.txt).  Whether a wave's cells are single pixels is a wave-uniform question (pixel size 2: always).
    const bool one_px = __ballot(maker && !(same_x && same_y)) == 0ull;
    if (one_px) {
        RawRGB raw[MAX_GROUP];
#pragma unroll
        for (int jf = 0; jf < MAX_GROUP; ++jf) raw[jf] = load_raw(PIX, G.f[min(jf, nseq - 1)].in, erow0 + ea[0], erow0 + ea[1], erow0 + ea[2]);
        if (maker) {
            if constexpr (SF == SF_LEAN_RT || (SF & KF_GRADE_RT) != 0) {
                // a colour grade at run time: its code (three inlined powf) once, not once per frame — unrolled, the kernel was 101 KB of code,
                // past the instruction cache.  The raw samples wait in the entry's own LDS slot (a rolled loop cannot index registers).
#pragma unroll
                for (int jf = 0; jf < MAX_GROUP; ++jf)
                    if (jf < nseq) dst[jf * nent + (int)threadIdx.x] = float4{__uint_as_float(raw[jf].r), __uint_as_float(raw[jf].g), __uint_as_float(raw[jf].b), 0.0f};
#pragma unroll 1
                for (int jf = 0; jf < nseq; ++jf) {
                    const float4 w = dst[jf * nent + (int)threadIdx.x];
                    const RawRGB rw{__float_as_uint(w.x), __float_as_uint(w.y), __float_as_uint(w.z)};
                    float a[3];
                    graded(rw, a);
                    put_entry(jf, a, a, a, a);
                }
            } else {
#pragma unroll
                for (int jf = 0; jf < MAX_GROUP; ++jf) {
                    float a[3];
                    graded(raw[jf], a);
                    put_entry(jf, a, a, a, a);
                }
            }
        }
    } else {
#pragma unroll 1
        for (int jf = 0; jf < nseq; ++jf) {
            const uint8_t* in = G.f[jf].in;
            const RawRGB ra = load_raw(PIX, in, erow0 + ea[0], erow0 + ea[1], erow0 + ea[2]), rb = load_raw(PIX, in, erow0 + ea[3], erow0 + ea[4], erow0 + ea[5]);
            const RawRGB rc = load_raw(PIX, in, erow1 + ea[0], erow1 + ea[1], erow1 + ea[2]), rd = load_raw(PIX, in, erow1 + ea[3], erow1 + ea[4], erow1 + ea[5]);
            if (maker) {
                float a[3], b[3], c[3], d[3];
                graded(ra, a); graded(rb, b); graded(rc, c); graded(rd, d);      // (equal samples give equal values: same bits as half_body's shared fetches)
                put_entry(jf, a, b, c, d);
            }
        }
    }
    // ---- the thread's own pixels (frame-invariant part: k_point_lean_seq's) ----
    int yr[ROWS];
    uint32_t t00[ROWS];                                       // entry index of a pixel's top-left tap inside the LDS tile; the others sit at + 1, + 34, + 35
    uint32_t er[ROWS], eg[ROWS], eb[ROWS];
    float a0[ROWS], a1[ROWS], b0[ROWS], b1[ROWS];
    F3 st[ROWS];
    PixMasks M0[ROWS];
    int gsx[ROWS], gsy[ROWS];                                 // coarse grain (KF_COARSE builds): the pixel's taps into the coarse plane of normals, frame-invariant
    float ga1[ROWS], gb1[ROWS];
    const float* state_in = G.o[0].state_in ? G.o[0].state_in : G.o[0].state;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int y = yr[k] = min(ybase + k * waves, P.H - 1);     // a row past the bottom redoes the last one; its stores are skipped
        {
            KFrame F0 = G.f[0];
            if constexpr (!lean_scanplane<SF>()) F0.scan_plane = nullptr;      // (a plane build's frames carry no row table: the gain of frame 0 is read here, and again below)
            M0[k] = load_masks(P, F0, y, x);
        }
        {
            const int sx = P.ux_ofs[x], sy = P.uy_ofs[y];
            a1[k] = P.ux_a[x]; a0[k] = 1.0f - a1[k]; b1[k] = P.uy_a[y]; b0[k] = 1.0f - b1[k];
            // Tile coordinates.  The right / lower tap of cv2.resize is min(s + 1, last): entry (l + 1) of the tile holds exactly that, because the
            // tile is filled with ds(clamp(j0 + jj), clamp(i0 + ii)) — so the four taps sit at FIXED offsets from the first (one address
            // register per pixel, immediates for the rest).  The clamps only ever bind for rows below the frame's last block row (not stored).
            const int lx = min(max(sx - i0, 0), FUSED_TWH - 2), ly = min(max(sy - j0, 0), thh - 2);
            t00[k] = (uint32_t)(ly * FUSED_TWH + lx);
        }
        if constexpr (lean_coarse<SF>()) { gsx[k] = P.gx_ofs[x]; gsy[k] = P.gy_ofs[y]; ga1[k] = P.gx_a[x]; gb1[k] = P.gy_a[y]; }
        else { gsx[k] = gsy[k] = 0; ga1[k] = gb1[k] = 0.0f; }
        {   // = fetch_raw's addressing (ref:573-583), frame-invariant
            int xs = x, ys = y;
            if (P.flags & CRTFX_F_PIXELATE) { xs = P.xmap[x]; ys = P.ymap[y]; }
            const uint32_t row = (uint32_t)ys * (uint32_t)P.W * 3u;
            int xr = xs, xb = xs;
            if (P.ab != 0) { xr = wrap(xs - P.ab, P.W); xb = wrap(xs + P.ab, P.W); }
            er[k] = row + (uint32_t)xr * 3u; eg[k] = row + (uint32_t)xs * 3u + 1u; eb[k] = row + (uint32_t)xb * 3u + 2u;
        }
        if constexpr (BLENDM == CRTFX_BLEND_RENDER) st[k] = *reinterpret_cast<const F3*>(state_in + ((uint32_t)y * (uint32_t)P.W + (uint32_t)x) * 3u);
    }
    __syncthreads();                               // the tiles and the LUTs are visible: the only barrier of the kernel
    auto run = [&](auto tzero) {
        using T = decltype(tzero);
        for (int jf = 0; jf < nseq; ++jf) {
            KFrame F = G.f[jf];                        // wave-uniform index: scalar loads
            if constexpr (!lean_scanplane<SF>()) F.scan_plane = nullptr;
            F.noise_plane = nullptr; F.overlay_before = nullptr;
            KOut O = G.o[jf];
            O.pix = PIX;
            const bool keep_state = jf == nseq - 1 || G.o[jf + 1].state != O.state;
            const float4* __restrict__ tile = dst + jf * nent;
            T v[ROWS][3];
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const int y = yr[k];
                PixMasks M = M0[k];
                if constexpr (lean_scanplane<SF>()) M.sl = F.scan_plane[(uint32_t)y * (uint32_t)P.W + (uint32_t)x];
                else if (P.flags & CRTFX_F_SCANLINES) M.sl = F.scan_row[y];
                if constexpr (lean_coarse<SF>()) { M.z = coarse_grain(P, F, gsx[k], gsy[k], ga1[k], gb1[k]); M.has_z = 1; }      // = k_point_sel_seq's
                float r, g, b;
                lean_graded<SF, PIX>(P, glut, load_raw(PIX, F.in, er[k], eg[k], eb[k]), r, g, b);      // = fetch_graded (no overlay in the lean build)
                {
                    const float4* tp = tile + t00[k];
                    const float4 p00 = tp[0], p01 = tp[1], p10 = tp[FUSED_TWH], p11 = tp[FUSED_TWH + 1];
                    const float bl0 = (p00.x * a0[k] + p01.x * a1[k]) * b0[k] + (p10.x * a0[k] + p11.x * a1[k]) * b1[k];
                    const float bl1 = (p00.y * a0[k] + p01.y * a1[k]) * b0[k] + (p10.y * a0[k] + p11.y * a1[k]) * b1[k];
                    const float bl2 = (p00.z * a0[k] + p01.z * a1[k]) * b0[k] + (p10.z * a0[k] + p11.z * a1[k]) * b1[k];
                    r = clip01(r + P.bloom_strength * bl0); g = clip01(g + P.bloom_strength * bl1); b = clip01(b + P.bloom_strength * bl2);   // ref:611
                }
                tail_masks<T, false>(P, F, y, x, M, r, g, b, lut, lut + LUT_STRIDE, v[k][0], v[k][1], v[k][2]);
            }
            const T p = (T)O.p, q = (T)O.q;
            if (O.pre) {                                 // a warp follows: park the pre-warp pixels of this frame for k_warp_lean
#pragma unroll
                for (int k = 0; k < ROWS; ++k)
                    if (ybase + k * waves < P.H)
                        *reinterpret_cast<F3*>(O.pre + ((uint32_t)yr[k] * (uint32_t)P.W + (uint32_t)x) * 3u) = F3{(float)v[k][0], (float)v[k][1], (float)v[k][2]};
                continue;
            }
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                if (ybase + k * waves < P.H) {           // wave-uniform
                    const uint32_t pix = (uint32_t)yr[k] * (uint32_t)P.W + (uint32_t)x;
                    float f0, f1, f2;
                    if constexpr (BLENDM == CRTFX_BLEND_RENDER) {
                        f0 = (float)clip01(p * (T)st[k].x + q * v[k][0]);      // ref:1092
                        f1 = (float)clip01(p * (T)st[k].y + q * v[k][1]);
                        f2 = (float)clip01(p * (T)st[k].z + q * v[k][2]);
                        st[k] = F3{f0, f1, f2};
                    } else { f0 = (float)v[k][0]; f1 = (float)v[k][1]; f2 = (float)v[k][2]; }
                    if (O.state && (keep_state || BLENDM != CRTFX_BLEND_RENDER)) { float* sp = O.state + pix * 3u; sp[0] = f0; sp[1] = f1; sp[2] = f2; }
                    if (O.out_u8) {
                        PackedPix pk;
                        if constexpr (PIX == CRTFX_PIX_F16) { pk.lo = quant_f16(f0) | (quant_f16(f1) << 16); pk.hi = quant_f16(f2); }
                        else { pk.lo = quant_u8x3(f0, f1, f2); pk.hi = 0; }
                        store_row_pix(O, (size_t)yr[k] * P.W + x0, lane, min(64, P.W - x0), pk);
                    }
                }
            }
        }
    };
    if constexpr (SF == SF_LEAN_RT) { if (promotes(P)) run(0.0); else run(0.0f); }
    else if constexpr ((SF & (CRTFX_F_VIGNETTE | CRTFX_F_FLICKER)) != 0) run(0.0);
    else run(0.0f);
}
#endif  // CRTFX_MAIN_TU

}  // namespace crtfx
