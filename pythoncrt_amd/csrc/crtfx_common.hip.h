// crtfx_common.hip.h — kernel-side parameter blocks and the per-pixel stages shared by every kernel of the chain:
// a1 normalise, a2 aberration fetch, a4 grade, overlay blend, grain RNG, a7-a11 tail (tail_masks), a15 quantise, row stores,
// the commit epilogue, the packed-FMA helper.  (One of the parts of crtfx_kernels.hip.h.)
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "crtfx.h"

namespace crtfx {


constexpr int GENERIC_MAX_RADIUS = 64;    // the LDS-ring kernel k_phosphor<-1>: its ring of (NB + 2R) rows must fit LDS
constexpr int MAX_RADIUS = GENERIC_MAX_RADIUS;   // largest radius whose taps travel in the kernel arguments; beyond it (by default beyond 30) the split path, any radius
constexpr int MAX_TAPS = 2 * MAX_RADIUS + 1;

// ---------------------------------------------------------------------------------------
// kernel-side parameter blocks (passed by value as kernel arguments)
// ---------------------------------------------------------------------------------------
struct KParams {
    int H, W;
    int pix;             // crtfx_pixfmt of the frames: 0 = uint8, 1 = IEEE half on the same 0..255 scale
    uint32_t flags;
    int ab;
    int R;
    int grain;
    float sat, r_gain, b_gain, contrast, brightness, inv_gamma;
    float thr, thr_den, bloom_strength;
    float noise_scale;
    float warp_k, cx, cy;
    double vig_strength;
    float taps[MAX_TAPS];   // by value: lives in the kernarg segment -> scalar loads, provably invariant
    const float* __restrict__ triad_row;
    const float* __restrict__ triad_full;
    const float* __restrict__ lut_g;
    const float* __restrict__ lut_inv;
    const float* __restrict__ grade_lut;   // [3][256]: a1 + a4 per channel and uint8 code (saturation off), or nullptr
    const float* __restrict__ triad_comp;  // [2][LUT_N]: T_m[i] = lut_inv[idx(lut_g[i] * m)] for the mask values comp_m0 / comp_m1 (k_phosphor_ct), or nullptr
    uint32_t comp_m0, comp_m1;             // bit patterns of the two tabulated mask values (the most frequent ones of the triad row)
    const double* __restrict__ vig_nx2;
    const double* __restrict__ vig_ny2;
    const double* __restrict__ vig_full;
    const float* __restrict__ xhat;
    const float* __restrict__ yhat;
    const int* __restrict__ xmap;
    const int* __restrict__ ymap;
    // bilinear resize axes (cv2.resize INTER_LINEAR): source index of the first tap and weight of the second
    const int* __restrict__ gx_ofs; const float* __restrict__ gx_a;     // grain upsample, per output column  (ref:642)
    const int* __restrict__ gy_ofs; const float* __restrict__ gy_a;     //                 per output row
    int gw, gh;                                                         // small grain plane size
    const int* __restrict__ ux_ofs; const float* __restrict__ ux_a;     // fast bloom: half-res -> full upsample (ref:607)
    const int* __restrict__ uy_ofs; const float* __restrict__ uy_a;
    const int* __restrict__ dx_ofs; const float* __restrict__ dx_a;     // fast bloom: full -> half downsample when not an exact 2x (ref:606)
    const int* __restrict__ dy_ofs; const float* __restrict__ dy_a;
    int hw, hh;                                                         // half-res size (max(1, W//2), max(1, H//2))
    float* ds;                                                          // half-res thresholded source, hh x hw x 3 float32 (ctx scratch)
    const float* consts;                                                // ctx-owned: float 1,1,1,1 then 112 zero bytes — a valid address for loads a disabled stage would make (k_point_sel)
};

struct KFrame {
    const uint8_t* __restrict__ in;
    const float* __restrict__ scan_row;
    const float* __restrict__ scan_plane;
    const float* __restrict__ noise_plane;
    const uint8_t* __restrict__ overlay_before;   // H x W x 4 RGBA blended after the grade (ref:588-598), or nullptr
    double flicker;
    uint32_t key0, key1;
};

struct KOut {
    float* pre;          // pre-warp float image (two-kernel path) or nullptr
    float* out_f32;      // final static float image or nullptr
    uint8_t* out_u8;     // quantised frame or nullptr
    float* state;        // persistence state in/out or nullptr
    const float* state_in;   // previous state when it lives elsewhere than `state` (batch with per-frame states); nullptr = `state`
    int pix;             // crtfx_pixfmt of out_u8 (the quantised frame): uint8, or half = |x*255| unrounded
    int blend;           // crtfx_blend
    double p, q;         // persistence, 1 - persistence (double, as python computes them)
    const uint8_t* __restrict__ overlay_after;   // H x W x 4 RGBA blended after the warp (ref:653-663), or nullptr
    const int* __restrict__ glitch_offs;         // x offsets of the glitch band (ref:679-682 / 853-855), or nullptr
    int glitch_y0, glitch_cols, glitch_seg_len;  // first band row; offsets per row (1, W, or segments of glitch_seg_len pixels)
    unsigned long long* dbg;   // CRTFX_STAMP diagnostic build only: per-wave phase cycle sums
};

// Up to MAX_GROUP frames per launch (blockIdx.z = frame): small frames then fill the block slots of the
// chip without short, halo-heavy blocks, and fewer launches are needed.  The per-frame records travel by
// value in the kernel-argument segment and are picked with a wave-uniform index (scalar loads).
#ifndef CRTFX_MAX_GROUP
#define CRTFX_MAX_GROUP 8
#endif
constexpr int MAX_GROUP = CRTFX_MAX_GROUP;
// y0, y1 / y0: a launch may cover a BAND of a frame only — the rows [y0, y1) of k_phosphor_* (cut into row segments from y0), output rows
// from y0 of k_warp_lean — so that a frame whose float32 pre-warp image is larger than the Infinity Cache (8K: 398 MB) can be produced
// and consumed band by band (crtfx_process_batch); the host sets y0 = 0, y1 = H for whole-frame launches.
struct KGroup { KFrame f[MAX_GROUP]; KOut o[MAX_GROUP]; int y0, y1; };
struct KWarpGroup { const float* pre[MAX_GROUP]; KOut o[MAX_GROUP]; int y0; };

// internal gate (set by crtfx_set_params, never by callers): the analytic vignette gain lies in [0,1]
// (0 <= strength <= 1), so clip(x * gain) of an x in [0,1] is the identity and is skipped.
constexpr uint32_t KF_VIG_UNIT = 1u << 24;
// the full-chain gate set of BASELINE configs 2-5 (everything but the bloom flavour, warp and pixelate)
constexpr uint32_t SF_FULL_GATES = CRTFX_F_BLOOM | CRTFX_F_TRIAD | CRTFX_F_TRIAD_LUT | CRTFX_F_SCANLINES | CRTFX_F_VIGNETTE | CRTFX_F_NOISE | KF_VIG_UNIT;
// ... with the fast half-res bloom (the reference CLI's default), without / with pixelate
constexpr uint32_t SF_FAST = SF_FULL_GATES | CRTFX_F_BLOOM_FAST;
constexpr uint32_t SF_FAST_PIX = SF_FAST | CRTFX_F_PIXELATE;
constexpr uint32_t SF_LEAN_RT = 0xFFFFFFFEu;      // template value of the lean pointwise kernels: the gate word is read at run time (any plane-free gate set)
// ... and the middle form: the gates that decide which LOADS a pixel makes (bloom flavour, triad + LUT, scanlines, vignette, grain, pixelate) folded,
// the purely arithmetic ones — the colour grade, the bloom threshold, preserve-luma, flicker — read at run time: SF = SF_FAST[_PIX] | KF_GRADE_RT.
// The folded body stays one basic block per stage with its loads issued together; a grade costs its arithmetic and a few scalar branches.
constexpr uint32_t KF_GRADE_RT = 1u << 30;
// ... and, for uint8 frames graded WITHOUT a saturation change (every other stage of apply_color_adjustments is per channel, ref:292-304): the whole of
// a1 + a4 as one 3 x 256-entry table read (KParams::grade_lut, built on the host with the reference's own expressions) — no run-time branch at
// all, the body stays one basic block: SF = SF_FAST[_PIX] | KF_GRADE_LUT.  Brightness / contrast / gamma / temperature: the knobs users turn first.
constexpr uint32_t KF_GRADE_LUT = 1u << 29;
// ... and coarse grain (--grain-size > 1, ref:637-642: N(0,1) drawn at (H // g) x (W // g), bilinearly upsampled): the pixel's sample formed from four
// hashed normals with frame-invariant taps, as k_point_sel_seq does, in a folded build of k_point_fused_seq: SF = SF_FAST[_PIX] | KF_COARSE
constexpr uint32_t KF_COARSE = 1u << 28;
// ... and slanted / shaped scanlines (--scanline-angle, --scanline-thickness; make_scanline_mask_2d ref:308-328): the gain is a per-pixel, per-frame
// plane (crtfx_frame.scan_plane_dev) instead of a row table: one more load per pixel, SF = SF_FAST[_PIX] | KF_SCANPLANE
constexpr uint32_t KF_SCANPLANE = 1u << 27;
constexpr uint32_t GRADE_RT_MASK = CRTFX_F_SATURATION | CRTFX_F_TEMPERATURE | CRTFX_F_BRIGHTCON | CRTFX_F_GAMMA | CRTFX_F_BLOOM_THR | CRTFX_F_TRIAD_LUMA | CRTFX_F_FLICKER;

constexpr int TW = 64;            // strip width in pixels (one wavefront of columns)
constexpr int NB = 8;             // rows per H-pass block / register-blocked V outputs
constexpr int K1_THREADS = 192;   // 3 wavefronts: wave w owns channel w in the V pass
constexpr int LUT_N = 1025;
constexpr int LUT_STRIDE = 1028;

__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ double clip01(double v) { return fmin(fmax(v, 0.0), 1.0); }

// a1 — u8/255.0 correctly rounded without the full division sequence: one Newton correction
// of the reciprocal product; exhaustively equal to IEEE division for 0..255 (tests/test_parity_gpu).
__device__ __forceinline__ float norm_u8(uint32_t u) {
    const float f = (float)u;
    const float rcp = 1.0f / 255.0f;
    const float q = f * rcp;
    const float r = fmaf(-q, 255.0f, f);
    return fmaf(r, rcp, q);
}

// (x mod W) for x in [-8, W+8): |aberration| <= 8 (ref:1230), so one conditional add/subtract
// replaces the integer division unless the image is narrower than the shift.
__device__ __forceinline__ int wrap(int x, int W) {
    if (W > 8) return x < 0 ? x + W : (x >= W ? x - W : x);
    x %= W;
    return x < 0 ? x + W : x;
}

// a1+a2(+a3): one RGB sample of the aberrated (and pixelated) float image; (y, x) in range.
// ref:569-584 — R'[x] = R[(x-d) mod W], B'[x] = B[(x+d) mod W]; pixelate = index maps.
struct RawRGB { uint32_t r, g, b; };
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };   // one RGB float pixel: a 12-byte, 4-aligned load   // the three stored samples of a pixel: bytes, or half bit patterns

// a1 for either pixel format: uint8 -> u/255 (norm_u8); half -> float(h)/255 with a true division
// (ref:569 `frame.astype(np.float32) / 255.0` applied to a float16 frame array).
__device__ __forceinline__ float norm_px(int pix, uint32_t s) {
    if (pix == CRTFX_PIX_F16) {
        // float(h) / 255.0f by the same corrected reciprocal product as norm_u8: equal to the IEEE quotient for
        // every finite half (all 63 488 checked, tests/test_parity_gpu.py::test_fp16_normalise_exhaustive)
        const float f = (float)__builtin_bit_cast(_Float16, (unsigned short)s);
        const float rcp = 1.0f / 255.0f;
        const float q = f * rcp;
        return fmaf(fmaf(-q, 255.0f, f), rcp, q);
    }
    return norm_u8(s);
}
__device__ __forceinline__ RawRGB load_raw(int pix, const uint8_t* __restrict__ in, uint32_t er, uint32_t eg, uint32_t eb) {
    RawRGB v;      // er/eg/eb: ELEMENT offsets of the three samples from the frame base
    if (pix == CRTFX_PIX_F16) {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(in);
        v.r = p[er]; v.g = p[eg]; v.b = p[eb];
    } else {
        v.r = in[er]; v.g = in[eg]; v.b = in[eb];
    }
    return v;
}
// (y, x) already mapped through the pixelate index maps (or pixelate off): no dependent loads.
__device__ __forceinline__ RawRGB fetch_raw(const KParams& P, const uint8_t* __restrict__ in, int y, int x) {
    if (P.flags & CRTFX_F_PIXELATE) { x = P.xmap[x]; y = P.ymap[y]; }
    const uint32_t row = (uint32_t)y * (uint32_t)P.W * 3u;
    int xr = x, xb = x;
    if (P.ab != 0) { xr = wrap(x - P.ab, P.W); xb = wrap(x + P.ab, P.W); }
    return load_raw(P.pix, in, row + (uint32_t)xr * 3u, row + (uint32_t)x * 3u + 1u, row + (uint32_t)xb * 3u + 2u);
}
__device__ __forceinline__ void fetch_rgb(const KParams& P, const uint8_t* __restrict__ in, int y, int x,
                                          float& r, float& g, float& b) {
    const RawRGB v = fetch_raw(P, in, y, x);
    r = norm_px(P.pix, v.r); g = norm_px(P.pix, v.g); b = norm_px(P.pix, v.b);
}

// a4 — apply_color_adjustments (ref:279-305), float32 throughout.
__device__ __forceinline__ void grade(const KParams& P, float& r, float& g, float& b) {
    if (P.flags & CRTFX_F_SATURATION) {
        const float luma = (0.2126f * r + 0.7152f * g) + 0.0722f * b;
        r = clip01(luma + (r - luma) * P.sat);
        g = clip01(luma + (g - luma) * P.sat);
        b = clip01(luma + (b - luma) * P.sat);
    }
    if (P.flags & CRTFX_F_TEMPERATURE) {
        r = clip01(r * P.r_gain);
        b = clip01(b * P.b_gain);
    }
    if (P.flags & CRTFX_F_BRIGHTCON) {
        r = clip01(((r - 0.5f) * P.contrast + 0.5f) + P.brightness);
        g = clip01(((g - 0.5f) * P.contrast + 0.5f) + P.brightness);
        b = clip01(((b - 0.5f) * P.contrast + 0.5f) + P.brightness);
    }
    if (P.flags & CRTFX_F_GAMMA) {
        r = clip01(powf(r, P.inv_gamma));
        g = clip01(powf(g, P.inv_gamma));
        b = clip01(powf(b, P.inv_gamma));
    }
}

// text overlay (ref:588-598 / 653-663): alpha = a/255, rgb = c/255 (float32); img*(1-alpha) + rgb*alpha in the
// image dtype, the rgb*alpha product in float32 (both factors are float32 arrays), then clip.
template <typename T>
__device__ __forceinline__ void overlay_blend_px(uint32_t px, T& v0, T& v1, T& v2);
template <typename T>
__device__ __forceinline__ void overlay_blend(const uint8_t* __restrict__ ov, uint32_t pix, T& v0, T& v1, T& v2) {
    overlay_blend_px<T>(*reinterpret_cast<const uint32_t*>(ov + (size_t)pix * 4), v0, v1, v2);
}
template <typename T>
__device__ __forceinline__ void overlay_blend_px(uint32_t px, T& v0, T& v1, T& v2) {
    const float a = norm_u8(px >> 24), ia = 1.0f - a;
    const float c0 = norm_u8(px & 255u) * a, c1 = norm_u8((px >> 8) & 255u) * a, c2 = norm_u8((px >> 16) & 255u) * a;
    v0 = clip01(v0 * (T)ia + (T)c0); v1 = clip01(v1 * (T)ia + (T)c1); v2 = clip01(v2 * (T)ia + (T)c2);
}

// a1..a4 (+ overlay-before) of one pixel of the general-purpose kernels; (y, x) in range.
__device__ __forceinline__ void fetch_graded(const KParams& P, const KFrame& F, int y, int x, float& r, float& g, float& b) {
    if (P.grade_lut && (P.flags & CRTFX_F_GAMMA)) {       // three table reads (L1-resident) instead of three powf
        const RawRGB v = fetch_raw(P, F.in, y, x);
        r = P.grade_lut[v.r]; g = P.grade_lut[256 + v.g]; b = P.grade_lut[512 + v.b];
    } else {
        fetch_rgb(P, F.in, y, x, r, g, b);
        grade(P, r, g, b);
    }
    if (F.overlay_before) overlay_blend<float>(F.overlay_before, (uint32_t)y * (uint32_t)P.W + (uint32_t)x, r, g, b);
}

// bloom source (ref:601-604)
__device__ __forceinline__ float bloom_src(const KParams& P, float v) {
    if (P.flags & CRTFX_F_BLOOM_THR) return clip01((v - P.thr) / P.thr_den);
    return v;
}

// a11 RNG: counter-based (stateless) hash -> Box-Muller.  One N(0,1) per pixel, shared by the
// three channels (ref:646-647).  Keyed by (seed, frame) through key0/key1.
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float grain_normal(uint32_t key0, uint32_t key1, uint32_t idx) {
    // one avalanche hash per pixel, split into two 16-bit uniforms (grain is added at ~1/255 of full
    // scale and then quantised, so 16 bits each is ample; the radius tops out at 4.7 sigma)
    const uint32_t a = lowbias32(idx ^ key0) ^ key1;
    const float u1 = (float)((a >> 16) + 1u) * 1.52587890625e-05f;       // (0, 1]
    const float u2 = (float)(a & 0xFFFFu) * 1.52587890625e-05f;          // [0, 1)
    const float l2 = __builtin_amdgcn_logf(u1);                          // log2
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * l2); // sqrt(-2 ln u1)
    return rad * __builtin_amdgcn_cosf(u2);                              // cos(2 pi u2)
}

// Per-pixel mask values of a7 (triad), a8 (scanline gain) and a9 (vignette), gathered by the
// caller: k_point / the generic kernel load them per pixel, k_phosphor_rr keeps the per-column
// ones in registers and the per-row ones in LDS.
struct PixMasks {
    float m0, m1, m2;   // triad mask RGB at this pixel
    float sl;           // scanline gain
    double vig;         // vignette gain (float64, ref:266-276)
    float z;            // the pixel's N(0,1) grain sample when the caller has already formed it (has_z != 0)
    int has_z;
};

__device__ __forceinline__ double vignette_gain(const KParams& P, double nx2, double ny2) {
    return 1.0 - P.vig_strength * clip01(nx2 + ny2);                    // ref:274-275
}

__device__ __forceinline__ PixMasks load_masks(const KParams& P, const KFrame& F, int y, int x) {
    PixMasks M{1.0f, 1.0f, 1.0f, 1.0f, 1.0};
    if (P.flags & CRTFX_F_TRIAD) {
        const float* m = P.triad_full ? P.triad_full + ((size_t)y * P.W + x) * 3 : P.triad_row + x * 3;
        M.m0 = m[0]; M.m1 = m[1]; M.m2 = m[2];
    }
    if (P.flags & CRTFX_F_SCANLINES) M.sl = F.scan_plane ? F.scan_plane[(size_t)y * P.W + x] : F.scan_row[y];
    if (P.flags & CRTFX_F_VIGNETTE)
        M.vig = P.vig_full ? P.vig_full[(size_t)y * P.W + x] : vignette_gain(P, P.vig_nx2[x], P.vig_ny2[y]);
    return M;
}

// LUT index of ref:250 / :261: clip(trunc(clip(v,0,1) * 1024), 0, 1024).  clip(v) * 1024 lies
// in [0, 1024] exactly, so the integer clip is the identity and is not re-applied.
__device__ __forceinline__ int lut_index(float v) { return (int)(clip01(v) * 1024.0f); }
// the same for a v already known to lie in [0,1] (every stage before the triad ends in a clip)
__device__ __forceinline__ int lut_index_unit(float v) { return (int)(v * 1024.0f); }

// a7..a11 — from the post-bloom image to the pre-warp image.  The reference's image is float32
// up to the scanline multiply and float64 from the vignette / flicker multiply on (NumPy
// promotion); T mirrors that so the values agree before the single final narrowing.
template <typename T, bool PLANES = true>
__device__ __forceinline__ void tail_masks(const KParams& P, const KFrame& F, int y, int x, const PixMasks& M,
                                           float r, float g, float b,
                                           const float* __restrict__ lut_g, const float* __restrict__ lut_inv,
                                           T& o0, T& o1, T& o2) {
    // a7 — _apply_triad_mask (ref:238-263)
    if (P.flags & CRTFX_F_TRIAD) {
        if (P.flags & CRTFX_F_TRIAD_LUT) {
            const float l0 = lut_g[lut_index_unit(r)], l1 = lut_g[lut_index_unit(g)], l2 = lut_g[lut_index_unit(b)];
            float q0 = l0 * M.m0, q1 = l1 * M.m1, q2 = l2 * M.m2;
            if (P.flags & CRTFX_F_TRIAD_LUMA) {
                        const float yb = (0.2126f * l0 + 0.7152f * l1) + 0.0722f * l2;
                const float ya = (0.2126f * q0 + 0.7152f * q1) + 0.0722f * q2;
                float ratio = yb / fmaxf(ya, 1e-6f);
                ratio = fminf(fmaxf(ratio, 0.5f), 2.0f);
                q0 *= ratio; q1 *= ratio; q2 *= ratio;
            }
            // LUT entries are linspace(0,1)^(1/gamma): already inside [0,1], the final clip (ref:263) is the identity
            r = lut_inv[lut_index(q0)]; g = lut_inv[lut_index(q1)]; b = lut_inv[lut_index(q2)];
        } else {
            r = clip01(r * M.m0); g = clip01(g * M.m1); b = clip01(b * M.m2);
        }
    }
    // a8 — scanlines (ref:617-624)
    if (P.flags & CRTFX_F_SCANLINES) { r = clip01(r * M.sl); g = clip01(g * M.sl); b = clip01(b * M.sl); }
    T v0 = (T)r, v1 = (T)g, v2 = (T)b;
    // a9 — vignette (ref:626-628): float64 mask promotes the image
    if (P.flags & CRTFX_F_VIGNETTE) {
        if (P.flags & KF_VIG_UNIT) { v0 = (T)((double)v0 * M.vig); v1 = (T)((double)v1 * M.vig); v2 = (T)((double)v2 * M.vig); }
        else { v0 = (T)clip01((double)v0 * M.vig); v1 = (T)clip01((double)v1 * M.vig); v2 = (T)clip01((double)v2 * M.vig); }
    }
    // a10 — flicker (ref:630-633); np.float64 factor
    if (P.flags & CRTFX_F_FLICKER) {
        v0 = (T)clip01((double)v0 * F.flicker); v1 = (T)clip01((double)v1 * F.flicker); v2 = (T)clip01((double)v2 * F.flicker);
    }
    // a11 — grain (ref:635-647): float32 noise * float32 scale, added in the image dtype
    if (P.flags & CRTFX_F_NOISE) {
        const uint32_t idx = (uint32_t)y * (uint32_t)P.W + (uint32_t)x;
        float z;
        if constexpr (PLANES) {
            if (P.grain > 1) {
                // ref:637-642: N(0,1) drawn at (H//g) x (W//g), cv2.resize INTER_LINEAR up to H x W:
                // horizontal lerp S[sx]*(1-a) + S[sx+1]*a on both rows, then the vertical one
                const int sx = P.gx_ofs[x], sy = P.gy_ofs[y];
                const int sx1 = min(sx + 1, P.gw - 1), sy1 = min(sy + 1, P.gh - 1);
                const float a1 = P.gx_a[x], a0 = 1.0f - a1, b1 = P.gy_a[y], b0 = 1.0f - b1;
                const uint32_t i00 = (uint32_t)sy * P.gw + sx, i01 = (uint32_t)sy * P.gw + sx1;
                const uint32_t i10 = (uint32_t)sy1 * P.gw + sx, i11 = (uint32_t)sy1 * P.gw + sx1;
                float n00, n01, n10, n11;
                if (F.noise_plane) { n00 = F.noise_plane[i00]; n01 = F.noise_plane[i01]; n10 = F.noise_plane[i10]; n11 = F.noise_plane[i11]; }
                else { n00 = grain_normal(F.key0, F.key1, i00); n01 = grain_normal(F.key0, F.key1, i01);
                       n10 = grain_normal(F.key0, F.key1, i10); n11 = grain_normal(F.key0, F.key1, i11); }
                z = (n00 * a0 + n01 * a1) * b0 + (n10 * a0 + n11 * a1) * b1;
            } else {
                z = F.noise_plane ? F.noise_plane[idx] : grain_normal(F.key0, F.key1, idx);
            }
        } else {
            z = M.has_z ? M.z : grain_normal(F.key0, F.key1, idx);
        }
        const float n = z * P.noise_scale;
        v0 = clip01(v0 + (T)n); v1 = clip01(v1 + (T)n); v2 = clip01(v2 + (T)n);
    }
    o0 = v0; o1 = v1; o2 = v2;
}

__device__ __forceinline__ bool promotes(const KParams& P) {
    return (P.flags & (CRTFX_F_VIGNETTE | CRTFX_F_FLICKER)) != 0;
}

// a15 — cv2.convertScaleAbs(alpha=255): saturate(round-half-even(|(float)x * 255|)).  v_cvt_pk_u8_f32 rounds to nearest-even and
// saturates to 0..255: the same function as (int)rintf + clamp for every float32 (tools/ubench/cvt_pk_u8_test.hip, see quant_u8x3).
__device__ __forceinline__ uint32_t quant_u8(float v) {
    uint32_t d = 0;
    const float s = fabsf(v * 255.0f);
    asm("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(d) : "v"(s));
    return d;
}

// a15 of one RGB pixel packed r | g<<8 | b<<16 with three v_cvt_pk_u8_f32: the instruction rounds to nearest-even and
// saturates to 0..255 — the same function as quant_u8 for every float32 (tools/ubench/cvt_pk_u8_test.hip: all 1.07 G
// values of [0, 1.25], the huge / inf / nan range and the negatives, 0 mismatches), in one 3.7-cycle instruction per
// channel instead of multiply-free rint + convert + clamp + shift + or.
__device__ __forceinline__ uint32_t quant_u8x3(float v0, float v1, float v2) {
    uint32_t d = 0;
    const float s0 = fabsf(v0 * 255.0f), s1 = fabsf(v1 * 255.0f), s2 = fabsf(v2 * 255.0f);
    asm("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(d) : "v"(s0));
    asm("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(d) : "v"(s1));
    asm("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(d) : "v"(s2));
    return d;
}

// a15 for half frames: |x*255| narrowed to half (convertScaleAbs without the integer rounding)
// TWO roundings, as numpy's `np.abs(x.astype(float32) * float32(255)).astype(float16)`: the float32 product, then RNE to half.  Written as
// (_Float16)fabsf(v * 255.0f) hipcc is free to emit v_fma_mixlo_f16 (product and narrowing in one rounding) in one kernel and v_mul_f32 +
// v_cvt_f16_f32 in another: the builds then differ by one half ulp wherever the float32 product is an exact tie (round 5: the general
// kernels against the lean ones, ~1 sample in 10^5, always the R channel).  The conversion is therefore pinned as its own instruction.
__device__ __forceinline__ uint32_t quant_f16(float v) {
    const float s = fabsf(v * 255.0f);
    uint32_t h;
    asm("v_cvt_f16_f32 %0, %1" : "=v"(h) : "v"(s));
    return h & 0xFFFFu;
}

struct PackedPix { uint32_t lo, hi; };   // uint8: lo = r | g<<8 | b<<16.  half: lo = r | g<<16, hi = b.

// Store 64 consecutive pixels' RGB bytes of one row from one wavefront.  Lane l holds pixel
// (x0 + l) packed as r | g<<8 | b<<16.  When the row segment is dword-aligned the wavefront
// re-packs through lane shuffles and lanes 0..47 store one dword each (192 contiguous bytes);
// otherwise each lane stores its 3 bytes.
__device__ __forceinline__ void store_row_u8(uint8_t* __restrict__ out, size_t row_byte0, int lane,
                                             int valid_px, uint32_t packed) {
    const bool aligned = ((row_byte0 & 3) == 0);   // wave-uniform
    if (aligned) {
        const int j = lane;                // dword index in the 192-byte segment
        const int a = (4 * j) / 3;         // first contributing pixel
        const int o = (4 * j) - 3 * a;     // byte offset inside that pixel (0..2)
        const uint32_t lo = __shfl(packed, a & 63);
        const uint32_t hi = __shfl(packed, (a + 1) & 63);
        // bytes lo[o..2] then hi[0..2]: (3 - o) + 3 >= 4 bytes, take the first four
        const uint64_t v = (uint64_t)lo | ((uint64_t)hi << 24);
        const uint32_t dw = (uint32_t)(v >> (8 * o));
        const int nbytes = valid_px * 3;
        if (j < 48) {
            if (4 * j + 4 <= nbytes) {
                *reinterpret_cast<uint32_t*>(out + row_byte0 + 4 * j) = dw;
            } else {
                for (int k = 0; k < 4; ++k)
                    if (4 * j + k < nbytes) out[row_byte0 + 4 * j + k] = (uint8_t)(dw >> (8 * k));
            }
        }
    } else if (lane < valid_px) {
        uint8_t* p = out + row_byte0 + (size_t)lane * 3;
        p[0] = (uint8_t)packed; p[1] = (uint8_t)(packed >> 8); p[2] = (uint8_t)(packed >> 16);
    }
}

// The same through a raw buffer resource over the output frame (k_warp_lean; frame bytes < 2^31): 32-bit offsets, and the
// dwords past the row segment's end are given an out-of-range offset, which the hardware drops — no byte tail, no 64-bit
// address arithmetic.  Needs the segment dword-aligned with a whole number of dwords (W % 4 == 0); else the byte form.
// AUX: the store's cache-policy bits (2 = nt: a frame that is written once and never read back by the GPU should not take
// Infinity-Cache lines from the pre-warp images its kernel is still reading — profiles/r04_warp_ablation.txt)
template <int AUX = 0>
__device__ __forceinline__ void store_row_u8_buf(__amdgpu_buffer_rsrc_t rs, uint32_t row_byte0, int lane, int valid_px, uint32_t packed, bool aligned) {
    if (aligned) {                           // wave-uniform
        const int j = lane;                // dword index in the 192-byte segment
        const int a = (4 * j) / 3;         // first contributing pixel
        const int o = (4 * j) - 3 * a;     // byte offset inside that pixel (0..2)
        const uint32_t lo = __shfl(packed, a & 63);
        const uint32_t hi = __shfl(packed, (a + 1) & 63);
        const uint64_t v = (uint64_t)lo | ((uint64_t)hi << 24);
        const uint32_t dw = (uint32_t)(v >> (8 * o));
        const uint32_t off = (4 * j + 4 <= valid_px * 3) ? row_byte0 + 4u * (uint32_t)j : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_buffer_store_b32(dw, rs, off, 0, AUX);
    } else {
        const uint32_t off = lane < valid_px ? row_byte0 + 3u * (uint32_t)lane : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)packed, rs, off, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(packed >> 8), rs, off + 1u, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(packed >> 16), rs, off + 2u, 0, AUX);
    }
}

// One row segment of half pixels: each lane stores its own three halves (6 bytes, 2-byte aligned).
__device__ __forceinline__ void store_row_f16(uint8_t* __restrict__ out, size_t row_px0, int lane, int valid_px, PackedPix pk) {
    if (lane < valid_px) {
        uint16_t* p = reinterpret_cast<uint16_t*>(out) + (row_px0 + (size_t)lane) * 3;
        p[0] = (uint16_t)pk.lo; p[1] = (uint16_t)(pk.lo >> 16); p[2] = (uint16_t)pk.hi;
    }
}
// The same row segment of half pixels as whole dwords through a buffer resource (rows of an even number of pixels: the segment's 6-byte
// pixels then fill dwords exactly).  Lane l holds pixel l as lo = r | g << 16, hi = b; dword d of the segment is
//   d = 3m: lo[2m]      d = 3m + 1: hi[2m] | (lo[2m+1] & 0xffff) << 16      d = 3m + 2: lo[2m+1] >> 16 | hi[2m+1] << 16
// so the wave stores dwords 0..63 with one instruction and 64..95 with a second (three cross-lane reads each) instead of three
// 2-byte stores per lane at a 6-byte stride.  Dwords past the segment's end get an out-of-range offset: dropped by the hardware.
template <int AUX = 0>
__device__ __forceinline__ void store_row_f16_buf(__amdgpu_buffer_rsrc_t rs, uint32_t row_byte0, int lane, int valid_px, PackedPix pk) {
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int d = part * 64 + lane;
        const int m = d / 3, t = d - 3 * m;
        const int sa = (2 * m + (t == 2 ? 1 : 0)) & 63, sb = (2 * m + 1) & 63;
        const uint32_t x = __shfl(pk.lo, sa), y = __shfl(pk.hi, sa), z = __shfl(pk.lo, sb);
        const uint32_t dw = t == 0 ? x : (t == 1 ? ((y & 0xFFFFu) | (z << 16)) : ((x >> 16) | (y << 16)));
        const uint32_t off = (d < 96 && 4 * d + 4 <= valid_px * 6) ? row_byte0 + 4u * (uint32_t)d : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_buffer_store_b32(dw, rs, off, 0, AUX);
    }
}
// ... and as whole QWORDS: ONE store instruction per wave and row instead of two (the texture-address unit is busy per instruction, whatever its
// width, and k_warp_lean<half> keeps it 82 % busy: profiles/r05_c5_vmem.json).  Needs a row segment of a multiple of four pixels (24 bytes =
// three qwords; the launcher checks W % 4 == 0).  Qword q = dwords 2q, 2q + 1 of the segment; with the dword map above the three lanes of a
// period take their two dwords from pixel pairs (4g, 4g + 1), (4g + 1, 4g + 2), (4g + 2, 4g + 3): four cross-lane reads per lane.
template <int AUX = 0>
__device__ __forceinline__ void store_row_f16_buf64(__amdgpu_buffer_rsrc_t rs, uint32_t row_byte0, int lane, int valid_px, PackedPix pk) {
    const int g = lane / 3, t = lane - 3 * g;                  // lanes 0 .. 47: qword = lane
    const int pa = (4 * g + t) & 63, pb = (4 * g + t + 1) & 63;
    const uint32_t la = __shfl(pk.lo, pa), ha = __shfl(pk.hi, pa), lb = __shfl(pk.lo, pb), hb = __shfl(pk.hi, pb);
    // t = 0: dwords 6g, 6g + 1 = lo[4g] , hi[4g] | lo[4g+1] << 16
    // t = 1: dwords 6g + 2, 6g + 3 = lo[4g+1] >> 16 | hi[4g+1] << 16 , lo[4g+2]
    // t = 2: dwords 6g + 4, 6g + 5 = hi[4g+2] | lo[4g+3] << 16 , lo[4g+3] >> 16 | hi[4g+3] << 16
    const uint32_t d0 = t == 0 ? la : (t == 1 ? ((la >> 16) | (ha << 16)) : ((ha & 0xFFFFu) | (lb << 16)));
    const uint32_t d1 = t == 0 ? ((ha & 0xFFFFu) | (lb << 16)) : (t == 1 ? lb : ((lb >> 16) | (hb << 16)));
    const uint32_t off = (lane < 48 && 8 * lane + 8 <= valid_px * 6) ? row_byte0 + 8u * (uint32_t)lane : 0xFFFFFFF0u;
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{d0, d1}, rs, off, 0, AUX);
}
__device__ __forceinline__ void store_row_pix(const KOut& O, size_t row_px0, int lane, int valid_px, PackedPix pk) {
    if (O.pix == CRTFX_PIX_F16) store_row_f16(O.out_u8, row_px0, lane, valid_px, pk);
    else store_row_u8(O.out_u8, row_px0 * 3, lane, valid_px, pk.lo);
}

// a14 + a15 — commit epilogue shared by every kernel that produces final pixels.
// T is the reference's image dtype at this point (double once promoted).
// Returns the packed u8 pixel; stores the float outputs itself.
template <typename T, bool BLEND = true>
__device__ __forceinline__ PackedPix commit_pixel(const KOut& O, uint32_t pix, T v0, T v1, T v2, uint32_t src_pix = 0xFFFFFFFFu) {
    if constexpr (BLEND) {
        // text overlay after the effects (ref:653-663); under a glitch gather it is the overlay of the SOURCE column
        if (O.overlay_after) overlay_blend<T>(O.overlay_after, src_pix == 0xFFFFFFFFu ? pix : src_pix, v0, v1, v2);
    }
    if (O.out_f32) {
        float* p = O.out_f32 + pix * 3u;
        p[0] = (float)v0; p[1] = (float)v1; p[2] = (float)v2;
    }
    if constexpr (!BLEND) {
        // lean kernels: the host routes blended commits through k_commit / k_warp
    } else if (O.blend == CRTFX_BLEND_RENDER) {           // ref:1092
        const float* s = (O.state_in ? O.state_in : O.state) + pix * 3u;
        const T p = (T)O.p, q = (T)O.q;
        v0 = clip01(p * (T)s[0] + q * v0); v1 = clip01(p * (T)s[1] + q * v1); v2 = clip01(p * (T)s[2] + q * v2);
    } else if (O.blend == CRTFX_BLEND_PREVIEW) {   // ref:693 addWeighted = fma(prev, a, img*b)
        const float* s = (O.state_in ? O.state_in : O.state) + pix * 3u;
        const T p = (T)O.p, q = (T)O.q;
        if constexpr (sizeof(T) == 8) {
            v0 = fma((T)s[0], p, v0 * q); v1 = fma((T)s[1], p, v1 * q); v2 = fma((T)s[2], p, v2 * q);
        } else {
            v0 = fmaf(s[0], p, v0 * q); v1 = fmaf(s[1], p, v1 * q); v2 = fmaf(s[2], p, v2 * q);
        }
    }
    const float f0 = (float)v0, f1 = (float)v1, f2 = (float)v2;
    if (O.state) {
        float* s = O.state + pix * 3u;
        s[0] = f0; s[1] = f1; s[2] = f2;
    }
    PackedPix pk;
    if (O.pix == CRTFX_PIX_F16) { pk.lo = quant_f16(f0) | (quant_f16(f1) << 16); pk.hi = quant_f16(f2); }
    else { pk.lo = quant_u8x3(f0, f1, f2); pk.hi = 0; }
    return pk;
}

// One finished pre-warp pixel: either park it for k_warp or commit it.
// Every lane of the wavefront must call this (store_row_u8 shuffles); `live` masks the pixel.
// LEAN: no per-pixel planes in the tail; COMMIT: overlay-after / blend compiled into the commit (the host routes such
// launches elsewhere when it is off).
template <bool LEAN = false, bool COMMIT = !LEAN>
__device__ __forceinline__ void emit_pixel(const KParams& P, const KFrame& F, const KOut& O, int y, int x0, int lane,
                                           bool live, const PixMasks& M, float r, float g, float b,
                                           const float* lut_g, const float* lut_inv) {
    const int x = x0 + lane;
    const uint32_t pix = (uint32_t)y * (uint32_t)P.W + (uint32_t)x;
    PackedPix packed{0, 0};
    if (promotes(P)) {
        double v0 = 0, v1 = 0, v2 = 0;
        if (LEAN || live) tail_masks<double, !LEAN>(P, F, y, x, M, r, g, b, lut_g, lut_inv, v0, v1, v2);   // lean callers pass valid (replicated) pixels in dead lanes: no branch
        if (O.pre) {
            if (live) { float* p = O.pre + pix * 3u; p[0] = (float)v0; p[1] = (float)v1; p[2] = (float)v2; }
            return;
        }
        if (live) packed = commit_pixel<double, COMMIT>(O, pix, v0, v1, v2);
    } else {
        float v0 = 0, v1 = 0, v2 = 0;
        if (LEAN || live) tail_masks<float, !LEAN>(P, F, y, x, M, r, g, b, lut_g, lut_inv, v0, v1, v2);
        if (O.pre) {
            if (live) { float* p = O.pre + pix * 3u; p[0] = v0; p[1] = v1; p[2] = v2; }
            return;
        }
        if (live) packed = commit_pixel<float, COMMIT>(O, pix, v0, v1, v2);
    }
    if (O.out_u8) store_row_pix(O, (size_t)y * P.W + x0, lane, min(64, P.W - x0), packed);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// acc.x += w.h * tp.lo', acc.y += w.h * tp.hi'  with ONE v_pk_fma_f32: h = low / high half of the VGPR pair w (broadcast to
// both lanes of the packed op through op_sel), (lo', hi') = the SGPR pair tp as it is or swapped.  The separable blur's
// 2 x (2R + 1) x 3 fused multiply-adds per pixel are 34 % of the kernel's VALU time (tools/isa_cost.py); one input
// feeds two neighbouring outputs with two neighbouring taps, which is exactly this instruction: measured 3.4 cycles
// against 2 x 2.4 for two v_fmac_f32 with an SGPR tap (profiles/r02_valu_cost.txt).  Each accumulator still receives
// its taps in the oracle's order (left to right / top to bottom), each product fused: the same bits.
__device__ __forceinline__ void pk_fma_bcast(f32x2& acc, f32x2 w, bool whigh, unsigned long long tp, bool swap) {
    if (!whigh && !swap) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "s"(tp));
    else if (!whigh && swap) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "+v"(acc) : "v"(w), "s"(tp));
    else if (whigh && !swap) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "s"(tp));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "s"(tp));
}



}  // namespace crtfx
