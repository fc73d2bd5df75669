// crtfx_warp.hip.h — k_warp / k_warp_lean (a12-a15) and the small utility kernels (scanline plane, state resize, warp map, noise plane, commit, halo fix-up)
// (one of the parts of crtfx_kernels.hip.h; see the chain overview there and DESIGN.md §3)
#pragma once
#include "crtfx_common.hip.h"

namespace crtfx {

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


// a*wa + b*wb + c*wc + d*wd, left to right as cv2.remap sums it.  In double every product of a float32 tap and a float32 weight is EXACT
// (24 + 24 significant bits <= 53), so fma(b, wb, a * wa) rounds the same real number as a * wa + b * wb does: one v_mul_f64 + three
// v_fma_f64 per channel instead of four multiplies and three adds (36 of 84 float64 instructions per thread of four pixels), same bits.
// In float the products round, so the float image keeps the multiply-then-add form.
__device__ __forceinline__ double wsum4(double a, double wa, double b, double wb, double c, double wc, double d, double wd) {
    return fma(d, wd, fma(c, wc, fma(b, wb, a * wa)));
}
__device__ __forceinline__ float wsum4(float a, float wa, float b, float wb, float c, float wc, float d, float wd) {
    return ((a * wa + b * wb) + c * wc) + d * wd;
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
    const T t00 = (T)u00, t01 = (T)u01, t10 = (T)u10, t11 = (T)u11;
    o0 = wsum4((T)A.x, t00, (T)B.x, t01, (T)C.x, t10, (T)D.x, t11);
    o1 = wsum4((T)A.y, t00, (T)B.y, t01, (T)C.y, t10, (T)D.y, t11);
    o2 = wsum4((T)A.z, t00, (T)B.z, t01, (T)C.z, t10, (T)D.z, t11);
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
// ROWS output rows per thread (y, y + 4 / WX, ...): the gathers of all of them are issued before the first is used —
// the kernel is bound by the fabric traffic of its float32 source, and what it needs from its shape is loads in flight:
// without a persistence state in registers 4 rows (16 pixel loads per thread) in a 128 x 8 tile measure 51.4 - 52.3 us per
// 2-frame 4K launch against 56.4 - 56.6 for 2 rows in 64 x 8 (64 x 16: 52.7 - 53.7; 8 rows: 66 - 68, the occupancy goes).
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
    const T w00 = (T)t.u00, w01 = (T)t.u01, w10 = (T)t.u10, w11 = (T)t.u11;
    o0 = wsum4((T)t.A.x, w00, (T)t.B.x, w01, (T)t.C.x, w10, (T)t.D.x, w11);
    o1 = wsum4((T)t.A.y, w00, (T)t.B.y, w01, (T)t.C.y, w10, (T)t.D.y, w11);
    o2 = wsum4((T)t.A.z, w00, (T)t.B.z, w01, (T)t.C.z, w10, (T)t.D.z, w11);
}

// nseq: the frames of G that each thread takes ONE AFTER THE OTHER (slice blockIdx.z of the group's ntot frames).  The persistence recurrence
// ref:1092 is per pixel (state_n = clip(p * state_{n-1} + q * img_n) of the same pixel; the warp's gather reads the frame's
// own pre-warp image, not the state), so a thread keeps its pixels' state in registers across the frames of a group: the
// map coordinates and weights are computed once, and the float32 state (12 + 12 bytes per pixel and frame, more than
// the frame's own 12 + 3) is read for the first frame and written behind the last one only (or behind every frame whose
// record names a state buffer of its own: crtfx_process_batch's local_states).  Same operations in the same order per
// pixel as one launch per frame: the same bits.  SEQ = false (frames without a blend): one frame per thread, nf = 1 at compile time —
// the geometry is 60 % of a pixel's instructions, but keeping it in registers across a frame loop costs more occupancy than it saves
// issue slots (profiles/r03_ct_ablation.txt, E).
// IDENT: no warp — the commit alone (a persistence blend behind the Gaussian chain with warp off): the tap is the pixel itself.
// WX: waves side by side in a block's tile — (64 * WX) pixels x (4 / WX * ROWS) rows, a thread's rows 4 / WX apart.
// PLAIN: what the launcher has checked for the whole group — frames out (never null; uint8 with W % 4 == 0, or half with W % 2 == 0 and a
// dword-aligned base: whole dwords per row segment), and either no state to keep (unblended frames) or ONE state buffer shared by every frame of
// a persistence chain (stored once, behind the chain's last frame: 39.6 -> 37.2 us per 5-frame 1080p launch) — so the body has no
// branch at all: a row past the bottom redoes the last one and its dword stores land beyond the output buffer's range (dropped by the
// hardware), and the sixteen tap loads of a thread issue before the first interpolation.
template <bool PROMOTE, int BLEND, int PIX, int ROWS, bool IDENT = false, int WX = 1, bool SEQ = true, bool PLAIN = false>
__global__ __launch_bounds__(256) void k_warp_lean(KParams P, KWarpGroup G, int nseq, int ntot) {
    static_assert(!PLAIN || (!IDENT && ((BLEND == CRTFX_BLEND_NONE && !SEQ) || (BLEND == CRTFX_BLEND_RENDER && SEQ && PIX == CRTFX_PIX_U8))),
                  "PLAIN: unblended frames behind a warp, or a persistence chain of uint8 frames that shares ONE state buffer");
    constexpr int WY = 4 / WX;
    using T = typename std::conditional<PROMOTE, double, float>::type;
    const int z0 = (int)blockIdx.z * nseq;                   // BLEND_RENDER: one z slice
    const int nf = SEQ ? min(nseq, ntot - z0) : 1;          // !SEQ: launched with nseq = 1
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int x0 = (blockIdx.x * WX + wv % WX) * TW;
    const int ybase = G.y0 + blockIdx.y * (WY * ROWS) + wv / WX;
    if (ybase >= P.H || x0 >= P.W) return;
    // (s_setprio 1 / 3 once the taps have been requested — a wave whose taps have arrived drains ahead of the waves still
    // issuing loads — measured slower: 59.2 / 60.8 vs 56.6 us per 2-frame 4K launch.)
    const int x = min(x0 + lane, P.W - 1);
    const bool live = x0 + lane < P.W;
    // geometry of this thread's ROWS pixels: frame-invariant
    float u00[ROWS], u01[ROWS], u10[ROWS], u11[ROWS];
    uint32_t off_a[ROWS], off_b[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int y = min(ybase + WY * r, P.H - 1);           // a row past the bottom redoes the last one; its stores are skipped
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
        // 24-bit multiplies (|iyc| <= 32767, W * 12 < 2^19, |ixc| <= 32767): v_mad_i32_i24 at half the cost of the 64-bit multiply-add hipcc forms for the 32-bit product
        off_a[r] = (uint32_t)(__mul24(iyc, P.W * 12) + __mul24(ixc, 12));        // a negative offset (rows -2, -1) wraps far past the buffer's end
        off_b[r] = off_a[r] + (uint32_t)P.W * 12u;
    }
    F3 st[ROWS];
    if constexpr (BLEND == CRTFX_BLEND_RENDER) {
        const float* state_in = G.o[0].state_in ? G.o[0].state_in : G.o[0].state;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int y = min(ybase + WY * r, P.H - 1);
            st[r] = *reinterpret_cast<const F3*>(state_in + ((uint32_t)y * (uint32_t)P.W + (uint32_t)x) * 3u);
        }
    }
    for (int jf = 0; jf < nf; ++jf) {
        const float* __restrict__ pre = G.pre[z0 + jf];      // wave-uniform index: scalar loads
        const KOut O = G.o[z0 + jf];
        const __amdgpu_buffer_rsrc_t pre_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pre), 0, (int)((uint32_t)P.H * (uint32_t)P.W * 12u), 0x00020000);
        const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(O.out_u8, 0, O.out_u8 ? (int)((uint32_t)P.H * (uint32_t)P.W * (PIX == CRTFX_PIX_F16 ? 6u : 3u)) : 0, 0x00020000);
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
            const int y = ybase + WY * r;
            if constexpr (!PLAIN) { if (y >= P.H) break; }        // wave-uniform
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
            if constexpr (PLAIN) {
                // row y >= H: its byte offset is >= the buffer's size: every dword of the row is dropped
                if constexpr (PIX == CRTFX_PIX_F16) {
                    const PackedPix pk{quant_f16(f0) | (quant_f16(f1) << 16), quant_f16(f2)};
                    const uint32_t rb = __umul24((uint32_t)y, (uint32_t)P.W * 6u) + (uint32_t)x0 * 6u;
                    if ((P.W & 3) == 0) store_row_f16_buf64<2>(out_rs, rb, lane, min(64, P.W - x0), pk);      // kernel-argument-uniform: whole qwords, one store per row
                    else store_row_f16_buf<2>(out_rs, rb, lane, min(64, P.W - x0), pk);
                }
                else
                    store_row_u8_buf<2>(out_rs, __umul24((uint32_t)y, (uint32_t)P.W * 3u) + (uint32_t)x0 * 3u, lane, min(64, P.W - x0), quant_u8x3(f0, f1, f2), true);
                continue;
            }
            if (O.state && live && keep_state) *reinterpret_cast<F3*>(O.state + pix * 3u) = F3{f0, f1, f2};
            if (O.out_u8) {
                if constexpr (PIX == CRTFX_PIX_F16) {
                    store_row_f16(O.out_u8, (size_t)y * P.W + x0, lane, min(64, P.W - x0), PackedPix{quant_f16(f0) | (quant_f16(f1) << 16), quant_f16(f2)});
                } else {
                    store_row_u8_buf<2>(out_rs, ((uint32_t)y * (uint32_t)P.W + (uint32_t)x0) * 3u, lane, min(64, P.W - x0), quant_u8x3(f0, f1, f2), (P.W & 3) == 0);
                }
            }
        }
    }
    if constexpr (PLAIN && BLEND == CRTFX_BLEND_RENDER) {
        // the chain's one state buffer takes the state behind its last frame: lanes right of the frame and rows past its bottom get an
        // offset outside the resource (dropped), so this too is branch-free
        const __amdgpu_buffer_rsrc_t st_rs = __builtin_amdgcn_make_buffer_rsrc(G.o[z0].state, 0, (int)((uint32_t)P.H * (uint32_t)P.W * 12u), 0x00020000);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int y = ybase + WY * r;
            const uint32_t off = (live && y < P.H) ? (__umul24((uint32_t)y, (uint32_t)P.W) + (uint32_t)x) * 12u : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b96(u32x3{__float_as_uint(st[r].x), __float_as_uint(st[r].y), __float_as_uint(st[r].z)}, st_rs, off, 0, 0);
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
    // thickness 1 (--scanline-angle alone): np.power(s, 1.0) is s itself — the float64 pow is half of this kernel's 21 us per 1080p plane
    const double shaped = inv_sharp == 1.0 ? s : pow(s, inv_sharp);
    out[(size_t)y * W + x] = (float)(1.0 - strength * shaped);
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
