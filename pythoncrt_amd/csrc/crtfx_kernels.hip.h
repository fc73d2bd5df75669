// crtfx_kernels.hip.h — gfx950 device code of the CRT effect chain.
//
// Chain order and arithmetic follow crt_filter.py (ref:LINE) stage by stage; every float
// operation is written out in the reference's evaluation order and the translation unit is
// built with -ffp-contract=off, so the only fused multiply-adds are the explicit fmaf() chains
// of the separable blur (the accumulation form OpenCV's filter engine uses).  Tables that the
// reference gets from numpy transcendental calls (LUTs, scanline row gains, Gaussian taps,
// flicker factor) arrive precomputed from the host.
//
// Launch structure per frame:
//   k_phosphor   u8 frame -> [a1 normalise, a2 aberration, a3 pixelate map, a4 grade] ->
//                a5 separable Gaussian bloom through LDS (column strips, H-pass ring) ->
//                a7 triad, a8 scanlines, a9 vignette, a10 flicker, a11 grain
//                -> float32 pre-warp image (or, with no warp, straight to the commit epilogue)
//   k_point      the same chain with bloom off: purely pointwise, no LDS staging
//   k_warp       a12 barrel-warp bilinear gather (+ a14 persistence, a15 quantise)
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
constexpr int MAX_GROUP = 4;
struct KGroup { KFrame f[MAX_GROUP]; KOut o[MAX_GROUP]; };
struct KWarpGroup { const float* pre[MAX_GROUP]; KOut o[MAX_GROUP]; };

// internal gate (set by crtfx_set_params, never by callers): the analytic vignette gain lies in [0,1]
// (0 <= strength <= 1), so clip(x * gain) of an x in [0,1] is the identity and is skipped.
constexpr uint32_t KF_VIG_UNIT = 1u << 24;
// the full-chain gate set of BASELINE configs 2-5 (everything but the bloom flavour, warp and pixelate)
constexpr uint32_t SF_FULL_GATES = CRTFX_F_BLOOM | CRTFX_F_TRIAD | CRTFX_F_TRIAD_LUT | CRTFX_F_SCANLINES | CRTFX_F_VIGNETTE | CRTFX_F_NOISE | KF_VIG_UNIT;
// ... with the fast half-res bloom (the reference CLI's default), without / with pixelate
constexpr uint32_t SF_FAST = SF_FULL_GATES | CRTFX_F_BLOOM_FAST;
constexpr uint32_t SF_FAST_PIX = SF_FAST | CRTFX_F_PIXELATE;

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

// a15 — cv2.convertScaleAbs(alpha=255): saturate(round-half-even(|(float)x * 255|))
__device__ __forceinline__ uint32_t quant_u8(float v) {
    const float s = fabsf(v * 255.0f);
    const int r = (int)rintf(s);
    return (uint32_t)min(max(r, 0), 255);
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
__device__ __forceinline__ uint32_t quant_f16(float v) { return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)fabsf(v * 255.0f)); }   // RNE narrowing

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
        __builtin_amdgcn_raw_buffer_store_b32(dw, rs, off, 0, 0);
    } else {
        const uint32_t off = lane < valid_px ? row_byte0 + 3u * (uint32_t)lane : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)packed, rs, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(packed >> 8), rs, off + 1u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(packed >> 16), rs, off + 2u, 0, 0);
    }
}

// One row segment of half pixels: each lane stores its own three halves (6 bytes, 2-byte aligned).
__device__ __forceinline__ void store_row_f16(uint8_t* __restrict__ out, size_t row_px0, int lane, int valid_px, PackedPix pk) {
    if (lane < valid_px) {
        uint16_t* p = reinterpret_cast<uint16_t*>(out) + (row_px0 + (size_t)lane) * 3;
        p[0] = (uint16_t)pk.lo; p[1] = (uint16_t)(pk.lo >> 16); p[2] = (uint16_t)pk.hi;
    }
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
    else { pk.lo = quant_u8(f0) | (quant_u8(f1) << 8) | (quant_u8(f2) << 16); pk.hi = 0; }
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
template <uint32_t SF, int PIX>
__device__ __forceinline__ void half_body(const KParams& Pin, const KFrame& Fin, float* __restrict__ ds) {
    KParams P = Pin;
    KFrame F = Fin;
    if constexpr (SF != 0xFFFFFFFFu) { P.flags = SF; P.pix = PIX; F.overlay_before = nullptr; }
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= P.hw || j >= P.hh) return;
    float o[3];
    if (!P.dx_ofs) {
        float a[3], b[3], c[3], d[3];
        fetch_graded(P, F, 2 * j, 2 * i, a[0], a[1], a[2]);
        fetch_graded(P, F, 2 * j, 2 * i + 1, b[0], b[1], b[2]);
        fetch_graded(P, F, 2 * j + 1, 2 * i, c[0], c[1], c[2]);
        fetch_graded(P, F, 2 * j + 1, 2 * i + 1, d[0], d[1], d[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            o[k] = (((bloom_src(P, a[k]) + bloom_src(P, b[k])) + bloom_src(P, c[k])) + bloom_src(P, d[k])) * 0.25f;
    } else {
        const int sx = P.dx_ofs[i], sy = P.dy_ofs[j];
        const int sx1 = min(sx + 1, P.W - 1), sy1 = min(sy + 1, P.H - 1);
        const float a1 = P.dx_a[i], a0 = 1.0f - a1, b1 = P.dy_a[j], b0 = 1.0f - b1;
        float a[3], b[3], c[3], d[3];
        fetch_graded(P, F, sy, sx, a[0], a[1], a[2]);
        fetch_graded(P, F, sy, sx1, b[0], b[1], b[2]);
        fetch_graded(P, F, sy1, sx, c[0], c[1], c[2]);
        fetch_graded(P, F, sy1, sx1, d[0], d[1], d[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            o[k] = (bloom_src(P, a[k]) * a0 + bloom_src(P, b[k]) * a1) * b0 + (bloom_src(P, c[k]) * a0 + bloom_src(P, d[k]) * a1) * b1;
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
        else { pk.lo = quant_u8(f0) | (quant_u8(f1) << 8) | (quant_u8(f2) << 16); pk.hi = 0; }
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
template <uint32_t SF, int PIX, int BLENDM>
__global__ __launch_bounds__(1024) void k_point_lean_seq(KParams Pin, KGroup G, int nseq) {
    __shared__ float lut[2 * LUT_STRIDE];
    constexpr int ROWS = CRTFX_POINT_ROWS;
    KParams P = Pin;
    P.flags = SF; P.pix = PIX; P.triad_full = nullptr; P.vig_full = nullptr; P.grain = 1;
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
        if constexpr ((SF & CRTFX_F_BLOOM_FAST) != 0) {
            const int sx = P.ux_ofs[x], sy = P.uy_ofs[y];
            const int sx1 = min(sx + 1, P.hw - 1), sy1 = min(sy + 1, P.hh - 1);
            a1[k] = P.ux_a[x]; a0[k] = 1.0f - a1[k]; b1[k] = P.uy_a[y]; b0[k] = 1.0f - b1[k];
            o00[k] = (uint32_t)(sy * P.hw + sx) * 12u; o01[k] = (uint32_t)(sy * P.hw + sx1) * 12u;
            o10[k] = (uint32_t)(sy1 * P.hw + sx) * 12u; o11[k] = (uint32_t)(sy1 * P.hw + sx1) * 12u;
        }
        {   // = fetch_raw's addressing (ref:573-583), frame-invariant
            int xs = x, ys = y;
            if constexpr ((SF & CRTFX_F_PIXELATE) != 0) { xs = P.xmap[x]; ys = P.ymap[y]; }
            const uint32_t row = (uint32_t)ys * (uint32_t)P.W * 3u;
            int xr = xs, xb = xs;
            if (P.ab != 0) { xr = wrap(xs - P.ab, P.W); xb = wrap(xs + P.ab, P.W); }
            er[k] = row + (uint32_t)xr * 3u; eg[k] = row + (uint32_t)xs * 3u + 1u; eb[k] = row + (uint32_t)xb * 3u + 2u;
        }
        if constexpr (BLENDM == CRTFX_BLEND_RENDER) st[k] = *reinterpret_cast<const F3*>(state_in + ((uint32_t)y * (uint32_t)P.W + (uint32_t)x) * 3u);
    }
    const size_t slot = (size_t)P.hh * P.hw * 3;
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
            if constexpr ((SF & CRTFX_F_SCANLINES) != 0) M.sl = F.scan_row[y];
            float r, g, b;
            {   // = fetch_graded (no overlay in the lean build)
                const RawRGB raw = load_raw(PIX, F.in, er[k], eg[k], eb[k]);
                if (P.grade_lut && (P.flags & CRTFX_F_GAMMA)) { r = P.grade_lut[raw.r]; g = P.grade_lut[256 + raw.g]; b = P.grade_lut[512 + raw.b]; }
                else { r = norm_px(PIX, raw.r); g = norm_px(PIX, raw.g); b = norm_px(PIX, raw.b); grade(P, r, g, b); }
            }
            if constexpr ((SF & CRTFX_F_BLOOM_FAST) != 0) {
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
                    else { pk.lo = quant_u8(f0) | (quant_u8(f1) << 8) | (quant_u8(f2) << 16); pk.hi = 0; }
                    store_row_pix(O, (size_t)yr[k] * P.W + x0, lane, min(64, P.W - x0), pk);
                }
            }
        }
    }
}
#endif  // CRTFX_MAIN_TU

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
    float* nlut = reinterpret_cast<float*>(ytab);                          // (that build never pixelates: ytab is empty)
    float* glut = reinterpret_cast<float*>(ytab + ((Pin.flags & CRTFX_F_PIXELATE) ? seg_rows + 2 * R : 0));   // [3][256] grade table (runtime-gate build)                   // [seg_rows + 2R]: source row of halo row (pixelate)

    // The four waves of a block have unequal roles (the V-pass has 192 columns for 256 threads, wave 0 carries the
    // prefetches).  Rotating the roles by the block's dispatch number spreads them over the SIMDs of a CU: measured
    // 4K 165.1 us per 2-frame launch without, 161.6 with the low bits (>>3: 161.5, >>5: 163.1, >>8: 171.9).
    const int wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int tid = (threadIdx.x + ((wg_lin & 3) << 6)) & (RR_THREADS - 1);
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int H = P.H, W = P.W;
    const int x0 = blockIdx.x * TW;
    const int y_begin = blockIdx.y * seg_rows;
    const int y_end = min(H, y_begin + seg_rows);
    if (y_begin >= H) return;
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
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
#define LDS_AT(T, off) (*(T*)(uintptr_t)(uint32_t)(off))

// staging plane stride: >= the staged width and == 4 (mod 8) dwords, so that the two (row, channel) planes one 16-lane
// ds_read_b128 group covers in the H pass (8 lanes each, 32 bytes apart) land on disjoint banks
__host__ __device__ constexpr int cc_sws(int R) { return ((rr_swp(R) + 3) & ~7) + 4; }
constexpr int CC_HROW = 3 * TW + 4;       // H-row tile row stride in floats: == 4 (mod 32), the H pass's column-strided stores stay 2-way
__host__ __device__ constexpr int cc_cring_words(int R, int pix) { return pix ? (rr_cring(R) * TW * 3 + 1) / 2 : rr_cring(R) * TW; }   // half: [CR][TW][3] uint16; uint8: [CR][TW] packed r | g<<8 | b<<16
// LDS words: staging, ONE H-row tile, LUTs, centre ring, a1 table (uint8), vignette tile (f64), two grain tiles (f32), row table
#ifdef CC_EXP_FUSEWARP
// TIMING EXPERIMENT, never shipped ("Fusing the warp"): the tail's pixels go to a 16-row LDS ring instead of
// HBM and every consumer thread also does k_warp_lean's work for its share of the block's 8 x 64 output pixels — map
// coordinates, four 12-byte taps (from the ring, at pseudo-addresses: the pixels are WRONG, the instruction stream, LDS
// traffic and stores are those of a fused kernel without its ownership search and halo), float64 interpolation, quantise,
// uint8 row store.  A lower bound on what a real fused kernel would cost.
constexpr int CC_WRING_WORDS = 2 * NB * 3 * TW;
struct WarpTapsE { F3 A, B, C, D; float u00, u01, u10, u11; };
__device__ __forceinline__ void warp_coords_e(const KParams& P, int y, int x, int& ix, int& iy, int& fx, int& fy) {      // = warp_coords
    const float xv = P.xhat[x], yv = P.yhat[y];
    const float r2 = xv * xv + yv * yv;
    const float factor = 1.0f + P.warp_k * r2;
    const float mx = (xv * factor) * P.cx + P.cx;
    const float my = (yv * factor) * P.cy + P.cy;
    const int sx = (int)rintf(mx * 32.0f);
    const int sy = (int)rintf(my * 32.0f);
    ix = min(max(sx >> 5, -32768), 32767);
    iy = min(max(sy >> 5, -32768), 32767);
    fx = sx & 31; fy = sy & 31;
}
__device__ __forceinline__ void warp_combine_e(const WarpTapsE& t, double& o0, double& o1, double& o2) {                   // = warp_combine<double>
    o0 = (((double)t.A.x * (double)t.u00 + (double)t.B.x * (double)t.u01) + (double)t.C.x * (double)t.u10) + (double)t.D.x * (double)t.u11;
    o1 = (((double)t.A.y * (double)t.u00 + (double)t.B.y * (double)t.u01) + (double)t.C.y * (double)t.u10) + (double)t.D.y * (double)t.u11;
    o2 = (((double)t.A.z * (double)t.u00 + (double)t.B.z * (double)t.u01) + (double)t.C.z * (double)t.u10) + (double)t.D.z * (double)t.u11;
}
#else
constexpr int CC_WRING_WORDS = 0;
#endif
__host__ __device__ constexpr int cc_lds_words(int R, int pix) {
    return NB * 3 * cc_sws(R) + NB * CC_HROW + 2 * LUT_STRIDE + cc_cring_words(R, pix) + (pix ? 0 : 256) + NB * TW * 2 + 2 * NB * TW + 16 * 4 + CC_WRING_WORDS;
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
    constexpr uint32_t WRING_B = ROWTAB_B + 16 * 4 * 4;                      // CC_EXP_FUSEWARP only: [2][NB][TW][3] float
    static_assert(WRING_B + CC_WRING_WORDS * 4 == (uint32_t)cc_lds_words(R, PIX) * 4, "LDS map and cc_lds_words disagree");
    (void)WRING_B;
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
    const int y_begin = blockIdx.y * seg_rows;
    const int y_end = min(H, y_begin + seg_rows);
    if (y_begin >= H) return;

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
#ifdef CC_EXP_FUSEWARP
            {
                const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(O.out_u8, 0, (int)((uint32_t)H * (uint32_t)W * 3u), 0x00020000);
                const int yb2 = hb - 2 * NB - R;                     // rows the previous trip's C2 left in the ring
                const int xw = min(x0 + lane, W - 1);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int jj = 3 * u + wave;                     // this wave's output row of the block (8 of the 9 slots exist)
                    if (jj < NB) {
                        const int yy = yb2 + jj;
                        const int yc = min(max(yy, 0), H - 1);
                        int ix, iy, fx, fy;
                        warp_coords_e(P, yc, xw, ix, iy, fx, fy);
                        WarpTapsE t;
                        const float wx1 = (float)fx * 0.03125f, wx0 = 1.0f - wx1;
                        const float wy1 = (float)fy * 0.03125f, wy0 = 1.0f - wy1;
                        const float mx0 = (unsigned)ix < (unsigned)W ? wx0 : 0.0f, mx1 = (unsigned)(ix + 1) < (unsigned)W ? wx1 : 0.0f;
                        t.u00 = wy0 * mx0; t.u01 = wy0 * mx1; t.u10 = wy1 * mx0; t.u11 = wy1 * mx1;
                        const uint32_t ra = (uint32_t)(iy & 15) * (3 * TW * 4), rb = (uint32_t)((iy + 1) & 15) * (3 * TW * 4);
                        const uint32_t ca = (uint32_t)(ix & 63) * 12u, cb = (uint32_t)((ix + 1) & 63) * 12u;
                        auto px = [&](uint32_t o) { return F3{LDS_AT(lds_f32_t, WRING_B + o), LDS_AT(lds_f32_t, WRING_B + o + 4), LDS_AT(lds_f32_t, WRING_B + o + 8)}; };
                        t.A = px(ra + ca); t.B = px(ra + cb); t.C = px(rb + ca); t.D = px(rb + cb);
                        double w0, w1, w2;
                        warp_combine_e(t, w0, w1, w2);
                        const bool rowok = yy >= y_begin && yy < y_end;     // wave-uniform
                        store_row_u8_buf(out_rs, rowok ? ((uint32_t)yc * (uint32_t)W + (uint32_t)x0) * 3u : 0xFFFFFF00u, lane, min(64, W - x0),
                                         quant_u8x3((float)w0, (float)w1, (float)w2), (W & 3) == 0);
                    }
                }
            }
#endif
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
#ifdef CC_EXP_ANLUT_TA      // A/B: the A items' a1 lookups as L1 gathers from a global copy of the table, issued before the V pass
            float nv[AO][3];
#pragma unroll
            for (int u = 0; u < AO; ++u) { nv[u][0] = P.consts[32 + raw[u].r]; nv[u][1] = P.consts[32 + raw[u].g]; nv[u][2] = P.consts[32 + raw[u].b]; }
#endif
            CC_PRIO(CC_P_VH);
            v_pass(blur);
            CC_PRIO(CC_P_A);
            STAMP(4);
            {
#ifndef CC_EXP_ANLUT_TA
                float nv[AO][3];
#pragma unroll
                for (int u = 0; u < AO; ++u) a_lookup(raw[u], nv[u]);
#endif
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
#ifdef CC_EXP_FUSEWARP
                    LDS_AT(lds_f32_t, WRING_B + (uint32_t)((((n & 1) * NB + j) * 3 * TW + f) * 4)) = v[j];
                    (void)oob;
#else
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j]), pre_rsrc, boff | oob, 0, 0);
#endif
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

// ---------------------------------------------------------------------------------------
// k_warp — barrel warp gather (ref:331-348 + cv2.remap INTER_LINEAR / BORDER_CONSTANT 0),
// then the commit epilogue.  One thread per output pixel; taps come straight from the
// float32 pre-warp image (L2 / Infinity-Cache resident: written by the preceding k_phosphor).
// identity != 0: no warp, read the pre-warp pixel itself (used when only the commit is wanted).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void warp_coords(const KParams& P, int y, int x, int& ix, int& iy, int& fx, int& fy) {
    const float xv = P.xhat[x], yv = P.yhat[y];
    const float r2 = xv * xv + yv * yv;
    const float factor = 1.0f + P.warp_k * r2;
    const float mx = (xv * factor) * P.cx + P.cx;
    const float my = (yv * factor) * P.cy + P.cy;
    const int sx = (int)rintf(mx * 32.0f);   // cvRound: ties to even
    const int sy = (int)rintf(my * 32.0f);
    ix = min(max(sx >> 5, -32768), 32767);   // saturate_cast<short>
    iy = min(max(sy >> 5, -32768), 32767);
    fx = sx & 31; fy = sy & 31;
}


// The four taps are loaded unconditionally from CLAMPED addresses (always inside the image) as
// 12-byte vectors, all four in flight together; a tap that lies outside the image is then
// replaced by the border value 0 (cv2.remap BORDER_CONSTANT), exactly what OpenCV's border
// branch feeds into the same weighted sum.
template <typename T>
__device__ __forceinline__ void warp_sample(const KParams& P, const float* __restrict__ pre, int ix, int iy, int fx, int fy,
                                            T& o0, T& o1, T& o2) {
    const float wx1 = (float)fx * 0.03125f, wx0 = 1.0f - wx1;
    const float wy1 = (float)fy * 0.03125f, wy0 = 1.0f - wy1;
    const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
    const bool xin0 = (unsigned)ix < (unsigned)P.W, xin1 = (unsigned)(ix + 1) < (unsigned)P.W;
    const bool yin0 = (unsigned)iy < (unsigned)P.H, yin1 = (unsigned)(iy + 1) < (unsigned)P.H;
    const int xa = min(max(ix, 0), P.W - 1), xb = min(max(ix + 1, 0), P.W - 1);
    const int ya = min(max(iy, 0), P.H - 1), yb = min(max(iy + 1, 0), P.H - 1);
    // 32-bit element offsets (H, W <= 32767 at 3 floats per pixel stay below 2^32): one 64-bit
    // add per tap instead of 64-bit multiplies
    const uint32_t rowa = (uint32_t)ya * (uint32_t)P.W, rowb = (uint32_t)yb * (uint32_t)P.W;
    const F3 A = *reinterpret_cast<const F3*>(pre + (rowa + (uint32_t)xa) * 3u);
    const F3 B = *reinterpret_cast<const F3*>(pre + (rowa + (uint32_t)xb) * 3u);
    const F3 C = *reinterpret_cast<const F3*>(pre + (rowb + (uint32_t)xa) * 3u);
    const F3 D = *reinterpret_cast<const F3*>(pre + (rowb + (uint32_t)xb) * 3u);
    // a tap outside the image contributes borderValue 0: 0 * w == v * 0 for finite v, so the tap's WEIGHT is
    // zeroed (4 selects) instead of its three channel values (12)
    const float u00 = (xin0 && yin0) ? w00 : 0.0f, u01 = (xin1 && yin0) ? w01 : 0.0f;
    const float u10 = (xin0 && yin1) ? w10 : 0.0f, u11 = (xin1 && yin1) ? w11 : 0.0f;
    o0 = (((T)A.x * (T)u00 + (T)B.x * (T)u01) + (T)C.x * (T)u10) + (T)D.x * (T)u11;
    o1 = (((T)A.y * (T)u00 + (T)B.y * (T)u01) + (T)C.y * (T)u10) + (T)D.y * (T)u11;
    o2 = (((T)A.z * (T)u00 + (T)B.z * (T)u01) + (T)C.z * (T)u10) + (T)D.z * (T)u11;
}

#ifdef CRTFX_MAIN_TU
__global__ __launch_bounds__(256) void k_warp(KParams P, KWarpGroup G, int identity) {
    const float* __restrict__ pre = G.pre[blockIdx.z];
    const KOut O = G.o[blockIdx.z];
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TW;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= P.H) return;
    const int x = x0 + lane;
    const bool live = x < P.W;
    const uint32_t pix = (uint32_t)y * (uint32_t)P.W + (uint32_t)x;
    PackedPix packed{0, 0};
    if (live) {
        // a13 glitch (ref:680-685 / 852-858): out[y, x] = post[y, (x + offs) mod W] for the rows of the bottom
        // band, post being the warped + overlaid image — so everything upstream is evaluated at column xs.
        int xs = x;
        if (O.glitch_offs && y >= O.glitch_y0) {
            const int col = O.glitch_seg_len > 0 ? x / O.glitch_seg_len : (O.glitch_cols == 1 ? 0 : x);
            const int off = O.glitch_offs[(size_t)(y - O.glitch_y0) * O.glitch_cols + col];
            xs = (x + off) % P.W;
            if (xs < 0) xs += P.W;
        }
        const uint32_t spix = (uint32_t)y * (uint32_t)P.W + (uint32_t)xs;
        if (promotes(P)) {
            double v0, v1, v2;
            if (identity) { const float* p = pre + spix * 3u; v0 = p[0]; v1 = p[1]; v2 = p[2]; }
            else {
                int ix, iy, fx, fy; warp_coords(P, y, xs, ix, iy, fx, fy);
#ifdef CRTFX_WARP_F32
                float f0, f1, f2; warp_sample<float>(P, pre, ix, iy, fx, fy, f0, f1, f2); v0 = f0; v1 = f1; v2 = f2;
#else
                warp_sample<double>(P, pre, ix, iy, fx, fy, v0, v1, v2);
#endif
            }
            packed = commit_pixel<double>(O, pix, v0, v1, v2, spix);
        } else {
            float v0, v1, v2;
            if (identity) { const float* p = pre + spix * 3u; v0 = p[0]; v1 = p[1]; v2 = p[2]; }
            else { int ix, iy, fx, fy; warp_coords(P, y, xs, ix, iy, fx, fy); warp_sample<float>(P, pre, ix, iy, fx, fy, v0, v1, v2); }
            packed = commit_pixel<float>(O, pix, v0, v1, v2, spix);
        }
    }
    if (O.out_u8) store_row_pix(O, (size_t)y * P.W + x0, lane, min(64, P.W - x0), packed);
}
#endif  // CRTFX_MAIN_TU

#ifdef CRTFX_MAIN_TU
// k_warp_lean — k_warp for the frames of a plain render: warp on, no glitch band, no overlay, no float output,
// blend NONE or RENDER.  Image dtype, blend mode and pixel format are compile-time, so the body is straight-line
// code: the four tap loads issue back to back and nothing waits on a branch (the general k_warp carries
// eight runtime paths; hipcc puts an s_waitcnt vmcnt(0) in front of every branch that contains a load).
// ROWS output rows per thread (y, y + 4, ...): the gathers of all of them are issued before the first is used.
struct WarpTaps { F3 A, B, C, D; float u00, u01, u10, u11; };
__device__ __forceinline__ WarpTaps warp_load(const KParams& P, const float* __restrict__ pre, int ix, int iy, int fx, int fy) {
    WarpTaps t;
    const float wx1 = (float)fx * 0.03125f, wx0 = 1.0f - wx1;
    const float wy1 = (float)fy * 0.03125f, wy0 = 1.0f - wy1;
    const bool xin0 = (unsigned)ix < (unsigned)P.W, xin1 = (unsigned)(ix + 1) < (unsigned)P.W;
    const bool yin0 = (unsigned)iy < (unsigned)P.H, yin1 = (unsigned)(iy + 1) < (unsigned)P.H;
    const int xa = min(max(ix, 0), P.W - 1), xb = min(max(ix + 1, 0), P.W - 1);
    const int ya = min(max(iy, 0), P.H - 1), yb = min(max(iy + 1, 0), P.H - 1);
    const uint32_t rowa = (uint32_t)ya * (uint32_t)P.W, rowb = (uint32_t)yb * (uint32_t)P.W;
    t.A = *reinterpret_cast<const F3*>(pre + (rowa + (uint32_t)xa) * 3u);
    t.B = *reinterpret_cast<const F3*>(pre + (rowa + (uint32_t)xb) * 3u);
    t.C = *reinterpret_cast<const F3*>(pre + (rowb + (uint32_t)xa) * 3u);
    t.D = *reinterpret_cast<const F3*>(pre + (rowb + (uint32_t)xb) * 3u);
    t.u00 = (xin0 && yin0) ? wy0 * wx0 : 0.0f; t.u01 = (xin1 && yin0) ? wy0 * wx1 : 0.0f;      // see warp_sample
    t.u10 = (xin0 && yin1) ? wy1 * wx0 : 0.0f; t.u11 = (xin1 && yin1) ? wy1 * wx1 : 0.0f;
    return t;
}
// The same four taps through a raw buffer resource over the pre-warp image (H * W * 12 bytes < 2^31: the host routes
// larger frames to the general k_warp).  One 32-bit byte offset per row pair, the right-hand tap in the instruction's
// immediate; a tap ABOVE or BELOW the image is an offset outside the buffer, for which the hardware's range check returns
// 0 — cv2.remap's border value — so only the x range needs masking (the linear offset of a column left / right of the
// image lands in a neighbouring row): the mask zeroes wx0 / wx1 before the four weights are formed.  Same products, same
// sums as warp_load + warp_combine (a zeroed weight times a finite tap and a finite weight times a zero tap are both +0);
// what goes is the 64-bit address arithmetic and the clamp / compare / select ladder: k_warp_lean was 75 % VALU-bound
// by cost (tools/isa_cost.py: 932 cycles per thread, 52 % of it integer).
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ F3 buf_load_px(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, off, 0, 0);
    return F3{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2])};
}
template <typename T>
__device__ __forceinline__ void warp_combine(const WarpTaps& t, T& o0, T& o1, T& o2) {
    o0 = (((T)t.A.x * (T)t.u00 + (T)t.B.x * (T)t.u01) + (T)t.C.x * (T)t.u10) + (T)t.D.x * (T)t.u11;
    o1 = (((T)t.A.y * (T)t.u00 + (T)t.B.y * (T)t.u01) + (T)t.C.y * (T)t.u10) + (T)t.D.y * (T)t.u11;
    o2 = (((T)t.A.z * (T)t.u00 + (T)t.B.z * (T)t.u01) + (T)t.C.z * (T)t.u10) + (T)t.D.z * (T)t.u11;
}

// nseq (BLEND_RENDER only): the frames of G that each thread takes ONE AFTER THE OTHER — the persistence recurrence
// ref:1092 is per pixel (state_n = clip(p * state_{n-1} + q * img_n) of the same pixel; the warp's gather reads the frame's
// own pre-warp image, not the state), so a thread keeps its pixels' state in registers across the frames of a group: the
// map coordinates and weights are computed once, and the float32 state (12 + 12 bytes per pixel and frame, more than
// the frame's own 12 + 3) is read for the first frame and written behind the last one only (or behind every frame whose
// record names a state buffer of its own: crtfx_process_batch's local_states).  Same operations in the same order per
// pixel as one launch per frame: the same bits.  Other blends: nseq = 1, blockIdx.z = frame.
// IDENT: no warp — the commit alone (a persistence blend behind the Gaussian chain with warp off): the tap is the pixel itself.
template <bool PROMOTE, int BLEND, int PIX, int ROWS, bool IDENT = false>
__global__ __launch_bounds__(256) void k_warp_lean(KParams P, KWarpGroup G, int nseq) {
    using T = typename std::conditional<PROMOTE, double, float>::type;
    const int z0 = BLEND == CRTFX_BLEND_RENDER ? 0 : (int)blockIdx.z;
    const int nf = BLEND == CRTFX_BLEND_RENDER ? nseq : 1;
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TW;
    const int ybase = blockIdx.y * (4 * ROWS) + (threadIdx.x >> 6);
    if (ybase >= P.H) return;
    // (s_setprio 1 / 3 once the taps have been requested — a wave whose taps have arrived drains ahead of the waves still
    // issuing loads — measured slower: 59.2 / 60.8 vs 56.6 us per 2-frame 4K launch.)
    const int x = min(x0 + lane, P.W - 1);
    const bool live = x0 + lane < P.W;
    // geometry of this thread's ROWS pixels: frame-invariant
    float u00[ROWS], u01[ROWS], u10[ROWS], u11[ROWS];
    uint32_t off_a[ROWS], off_b[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int y = min(ybase + 4 * r, P.H - 1);           // a row past the bottom redoes the last one; its stores are skipped
        if constexpr (IDENT) {
            u00[r] = 1.0f; u01[r] = u10[r] = u11[r] = 0.0f;
            off_a[r] = off_b[r] = ((uint32_t)y * (uint32_t)P.W + (uint32_t)x) * 12u;
            continue;
        }
        int ix, iy, fx, fy;
        warp_coords(P, y, x, ix, iy, fx, fy);
        const float wx1 = (float)fx * 0.03125f, wx0 = 1.0f - wx1;
        const float wy1 = (float)fy * 0.03125f, wy0 = 1.0f - wy1;
        const float mx0 = (unsigned)ix < (unsigned)P.W ? wx0 : 0.0f, mx1 = (unsigned)(ix + 1) < (unsigned)P.W ? wx1 : 0.0f;
        u00[r] = wy0 * mx0; u01[r] = wy0 * mx1; u10[r] = wy1 * mx0; u11[r] = wy1 * mx1;
        // clamps keep the offset arithmetic inside 32 bits: iy to [-2, H] (both rows of the pair stay outside when iy is), ix to
        // [-1, W] (the masks above come from the unclamped ix)
        const int ixc = min(max(ix, -1), P.W), iyc = min(max(iy, -2), P.H);
        off_a[r] = (uint32_t)(iyc * P.W + ixc) * 12u;        // a negative offset (rows -2, -1) wraps far past the buffer's end
        off_b[r] = off_a[r] + (uint32_t)P.W * 12u;
    }
    F3 st[ROWS];
    if constexpr (BLEND == CRTFX_BLEND_RENDER) {
        const float* state_in = G.o[0].state_in ? G.o[0].state_in : G.o[0].state;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int y = min(ybase + 4 * r, P.H - 1);
            st[r] = *reinterpret_cast<const F3*>(state_in + ((uint32_t)y * (uint32_t)P.W + (uint32_t)x) * 3u);
        }
    }
    for (int jf = 0; jf < nf; ++jf) {
        const float* __restrict__ pre = G.pre[z0 + jf];      // wave-uniform index: scalar loads
        const KOut O = G.o[z0 + jf];
        const __amdgpu_buffer_rsrc_t pre_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pre), 0, (int)((uint32_t)P.H * (uint32_t)P.W * 12u), 0x00020000);
        const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(O.out_u8, 0, O.out_u8 ? (int)((uint32_t)P.H * (uint32_t)P.W * 3u) : 0, 0x00020000);
        // the state is stored behind this frame when nobody keeps it in registers for the next one: the group's last frame,
        // or a frame whose record names its own state buffer
        const bool keep_state = BLEND != CRTFX_BLEND_RENDER || jf == nf - 1 || G.o[z0 + jf + 1].state != O.state;
        WarpTaps taps[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            taps[r].u00 = u00[r]; taps[r].u01 = u01[r]; taps[r].u10 = u10[r]; taps[r].u11 = u11[r];
            taps[r].A = buf_load_px(pre_rs, off_a[r]);
            if constexpr (!IDENT) {
                taps[r].B = buf_load_px(pre_rs, off_a[r] + 12u);
                taps[r].C = buf_load_px(pre_rs, off_b[r]); taps[r].D = buf_load_px(pre_rs, off_b[r] + 12u);
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int y = ybase + 4 * r;
            if (y >= P.H) break;                                  // wave-uniform
            const uint32_t pix = (uint32_t)y * (uint32_t)P.W + (uint32_t)x;
            T v0, v1, v2;
            if constexpr (IDENT) { v0 = (T)taps[r].A.x; v1 = (T)taps[r].A.y; v2 = (T)taps[r].A.z; }
            else warp_combine<T>(taps[r], v0, v1, v2);
            if constexpr (BLEND == CRTFX_BLEND_RENDER) {           // ref:1092
                const T p = (T)O.p, q = (T)O.q;
                v0 = clip01(p * (T)st[r].x + q * v0); v1 = clip01(p * (T)st[r].y + q * v1); v2 = clip01(p * (T)st[r].z + q * v2);
            }
            const float f0 = (float)v0, f1 = (float)v1, f2 = (float)v2;
            if constexpr (BLEND == CRTFX_BLEND_RENDER) st[r] = F3{f0, f1, f2};
            if (O.state && live && keep_state) *reinterpret_cast<F3*>(O.state + pix * 3u) = F3{f0, f1, f2};
            if (O.out_u8) {
                if constexpr (PIX == CRTFX_PIX_F16) {
                    store_row_f16(O.out_u8, (size_t)y * P.W + x0, lane, min(64, P.W - x0), PackedPix{quant_f16(f0) | (quant_f16(f1) << 16), quant_f16(f2)});
                } else {
                    store_row_u8_buf(out_rs, ((uint32_t)y * (uint32_t)P.W + (uint32_t)x0) * 3u, lane, min(64, P.W - x0), quant_u8x3(f0, f1, f2), (P.W & 3) == 0);
                }
            }
        }
    }
}
#endif  // CRTFX_MAIN_TU

#ifdef CRTFX_MAIN_TU
// crtfx_scanline_plane — make_scanline_mask_2d (ref:308-328) on the device: the slanted / thickness-shaped scanline
// gain the reference rebuilds on the CPU for every frame (float64 sin and pow per pixel, then cast to float32).
// Same expression tree in double; the device's sin/pow are not numpy's, so a value can come out one float32 ulp
// away from the host table when the double results straddle a float32 rounding boundary (rare: see
// tests/test_parity_gpu.py::test_scanline_plane_on_device).
__global__ void k_scan_plane(int H, int W, double strength, double omega, double phase, double tan_theta, double inv_sharp,
                             float* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W || y >= H) return;
    const double slanted = (double)y + tan_theta * (double)x;
    const double s = 0.5 * (1.0 + sin(omega * (slanted + phase)));
    out[(size_t)y * W + x] = (float)(1.0 - strength * pow(s, inv_sharp));
}
#endif  // CRTFX_MAIN_TU

#ifdef CRTFX_MAIN_TU
// crtfx_resize_state — cv2.resize(state_prev, (W, H), INTER_LINEAR) of ref:690: the previous persistence state
// arrives with another size (the preview window was resized between ticks).  OpenCV: source offset and FLOAT
// coefficient per axis from fx = (float)((d + 0.5) * scale - 0.5) (clamped to the edges with coefficient 0), the
// horizontal lerp of the two source rows first, then the vertical one, in the work type T (float for a float32
// state; double for the float64 state of a promoted chain, whose values the GPU holds rounded to float32);
// exact 2x decimation is OpenCV's area fast path, (a + b + c + d) * 0.25.
template <typename T>
__device__ __forceinline__ void resize_axis(int d, double scale, int n, int& s0, int& s1, T& c0, T& c1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.0f; s = 0; }
    if (s >= n - 1) { f = 0.0f; s = n - 1; }
    s0 = s; s1 = min(s + 1, n - 1);
    c1 = (T)f; c0 = (T)(1.0f - f);
}

template <typename T>
__global__ void k_resize_state(const float* __restrict__ src, int sh, int sw, float* __restrict__ dst, int dh, int dw,
                               double scale_x, double scale_y) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    float* o = dst + ((size_t)y * dw + x) * 3;
    if (dw * 2 == sw && dh * 2 == sh) {
        const float* p = src + ((size_t)(2 * y) * sw + 2 * x) * 3;
        const float* q = p + (size_t)sw * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (float)(((((T)p[c] + (T)p[3 + c]) + (T)q[c]) + (T)q[3 + c]) * (T)0.25);
        return;
    }
    int x0, x1, y0, y1;
    T a0, a1, b0, b1;
    resize_axis<T>(x, scale_x, sw, x0, x1, a0, a1);
    resize_axis<T>(y, scale_y, sh, y0, y1, b0, b1);
    const float* r0 = src + (size_t)y0 * sw * 3;
    const float* r1 = src + (size_t)y1 * sw * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const T h0 = (T)r0[x0 * 3 + c] * a0 + (T)r0[x1 * 3 + c] * a1;
        const T h1 = (T)r1[x0 * 3 + c] * a0 + (T)r1[x1 * 3 + c] * a1;
        o[c] = (float)(h0 * b0 + h1 * b1);
    }
}

#endif  // CRTFX_MAIN_TU

// crtfx_warp_map — the integer sampling map alone (parity: bit-exact against the oracle).
#ifdef CRTFX_MAIN_TU
__global__ void k_warp_map(KParams P, int* __restrict__ ix_out, int* __restrict__ iy_out, int* __restrict__ fxy_out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= P.W || y >= P.H) return;
    int ix, iy, fx, fy;
    warp_coords(P, y, x, ix, iy, fx, fy);
    const size_t i = (size_t)y * P.W + x;
    ix_out[i] = ix; iy_out[i] = iy; fxy_out[i] = (fy << 5) | fx;
}
#endif  // CRTFX_MAIN_TU

// crtfx_noise_plane — the RNG's N(0,1) draw for every pixel of a frame.
#ifdef CRTFX_MAIN_TU
__global__ void k_noise_plane(int n, uint32_t key0, uint32_t key1, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = grain_normal(key0, key1, (uint32_t)i);
}
#endif  // CRTFX_MAIN_TU

// crtfx_blend_quantise / crtfx_halo_correct_quantise — commit step on an existing float image.
// mode 0: blend per O.blend.  mode 1: v = clip(local + coeff*carry) (frame-sharded halo fix-up).
#ifdef CRTFX_MAIN_TU
__global__ __launch_bounds__(256) void k_commit(int H, int W, const float* __restrict__ src, const float* __restrict__ carry,
                                                double coeff, KOut O, int mode) {
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TW;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= H) return;
    const int x = x0 + lane;
    const bool live = x < W;
    const uint32_t pix = (uint32_t)y * (uint32_t)W + (uint32_t)x;
    PackedPix packed{0, 0};
    if (live) {
        const float* p = src + pix * 3u;
        if (mode == 1) {
            const float* c = carry + pix * 3u;
            const float cf = (float)coeff;
            const float v0 = clip01(p[0] + cf * c[0]), v1 = clip01(p[1] + cf * c[1]), v2 = clip01(p[2] + cf * c[2]);
            packed = commit_pixel<float>(O, pix, v0, v1, v2);
        } else {
            packed = commit_pixel<float>(O, pix, p[0], p[1], p[2]);
        }
    }
    if (O.out_u8) store_row_pix(O, (size_t)y * W + x0, lane, min(64, W - x0), packed);
}
#endif  // CRTFX_MAIN_TU

// crtfx_halo_correct_batch — the fix-up pass of a frame-sharded chunk for n frames in one launch:
// out_j = quantise(clip(local_j + coeff_j * carry)), coeff_j = p^(j+1).  A thread keeps its pixel of the carry in
// registers and walks the chunk's frames, so the carry is read once instead of once per frame.
constexpr int HALO_MAX_FRAMES = 64;
struct HaloCoeffs { float c[HALO_MAX_FRAMES]; };
#ifdef CRTFX_MAIN_TU
__global__ __launch_bounds__(256) void k_halo_batch(int H, int W, const float* __restrict__ local_base, size_t frame_elems,
                                                    const float* __restrict__ carry, HaloCoeffs K, int n, uint8_t* __restrict__ out_base,
                                                    size_t out_stride_bytes, int pix_fmt) {
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TW;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= H) return;
    const int x = min(x0 + lane, W - 1);
    const uint32_t pix = (uint32_t)y * (uint32_t)W + (uint32_t)x;
    const F3 c = *reinterpret_cast<const F3*>(carry + pix * 3u);
    KOut O{};
    O.pix = pix_fmt;
    for (int j = 0; j < n; ++j) {
        const F3 l = *reinterpret_cast<const F3*>(local_base + (size_t)j * frame_elems + pix * 3u);
        const float cf = K.c[j];
        const float v0 = clip01(l.x + cf * c.x), v1 = clip01(l.y + cf * c.y), v2 = clip01(l.z + cf * c.z);
        PackedPix pk;
        if (pix_fmt == CRTFX_PIX_F16) { pk.lo = quant_f16(v0) | (quant_f16(v1) << 16); pk.hi = quant_f16(v2); }
        else { pk.lo = quant_u8(v0) | (quant_u8(v1) << 8) | (quant_u8(v2) << 16); pk.hi = 0; }
        O.out_u8 = out_base + (size_t)j * out_stride_bytes;
        store_row_pix(O, (size_t)y * W + x0, lane, min(64, W - x0), pk);
    }
}
#endif  // CRTFX_MAIN_TU

}  // namespace crtfx
