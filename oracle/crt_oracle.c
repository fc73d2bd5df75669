/*
 * ORACLE (test infrastructure only) — C half of the CPU restatement of the PythonCRT
 * per-frame effect chain.  Nothing under pythoncrt_amd/ may link or load this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * These are the stages whose arithmetic the reference delegates to opencv-python-headless
 * (requirements.txt:6, ">=4.8.0", not vendored under /root/reference).  They restate the
 * published OpenCV 4.x algorithms; PARITY UNPINNED at this boundary (the reference holds no
 * tests or golden vectors, and cv2 is not installable here).
 *
 *   orc_sepblur_f32      cv2.GaussianBlur call sites crt_filter.py:234 (triad softening, ksize (k,1))
 *                        and :610 / :780 (bloom, ksize (k,k)), BORDER_REPLICATE.
 *                        OpenCV: sepFilter2D with CV_32F intermediates; RowFilter / ColumnFilter
 *                        accumulate tap 0 .. tap k-1 in order with fused multiply-add (the AVX2
 *                        build's v_muladd).  OpenCV's symmetric-column variant and its IPP path
 *                        reassociate the same sum; they differ from this by a few ulp.
 *   orc_remap_bilinear_* cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) crt_filter.py:347.
 *                        OpenCV: float maps are quantised to 1/32 px with cvRound
 *                        (ties-to-even), integer part = >>5 saturated to short, weights from the
 *                        32x32 bilinear table ((1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy fx in float),
 *                        v = S00*w0 + S01*w1 + S10*w2 + S11*w3 evaluated left to right in the
 *                        work type (float for CV_32F pixels, double for CV_64F pixels with float
 *                        weights); taps outside the image contribute borderValue 0.
 *   orc_resize_linear_f32 / orc_resize_nearest_f32
 *                        cv2.resize call sites :582-583/:751-752 (INTER_NEAREST pixelate),
 *                        :606-607/:776-777 (INTER_LINEAR fast bloom), :642/:812 (grain upsample).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off [-mfma]); fmaf() is exact either way.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Separable correlation, BORDER_REPLICATE, interleaved channels.
 * tmp must hold h*w*cn floats.  kx has nkx taps (anchor nkx/2), ky has nky taps. */
int orc_sepblur_f32(const float* src, float* dst, float* tmp, int h, int w, int cn,
                    const float* kx, int nkx, const float* ky, int nky)
{
    const int rx = nkx / 2, ry = nky / 2;
    for (int y = 0; y < h; ++y) {
        const float* srow = src + (size_t)y * w * cn;
        float* trow = tmp + (size_t)y * w * cn;
        for (int x = 0; x < w; ++x) {
            for (int c = 0; c < cn; ++c) {
                float s = 0.0f;
                for (int k = 0; k < nkx; ++k) {
                    int xx = clampi(x + k - rx, 0, w - 1);
                    s = fmaf(srow[(size_t)xx * cn + c], kx[k], s);
                }
                trow[(size_t)x * cn + c] = s;
            }
        }
    }
    const size_t rowlen = (size_t)w * cn;
    for (int y = 0; y < h; ++y) {
        float* drow = dst + (size_t)y * rowlen;
        for (size_t i = 0; i < rowlen; ++i) drow[i] = 0.0f;
        for (int k = 0; k < nky; ++k) {
            int yy = clampi(y + k - ry, 0, h - 1);
            const float* trow = tmp + (size_t)yy * rowlen;
            const float f = ky[k];
            for (size_t i = 0; i < rowlen; ++i) drow[i] = fmaf(trow[i], f, drow[i]);
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * ALTERNATIVE ACCUMULATION FORMS of the same separable blur — not the oracle, the SPREAD around it.
 * cv2.GaussianBlur on CV_32F is "parity unpinned" (no reference-held vector, cv2 not installable here), and OpenCV's
 * own code paths do not all add the taps in the order orc_sepblur_f32 does.  The forms OpenCV 4.x's C++ filter
 * engine (modules/imgproc/src/filter.simd.hpp) is known to contain are restated here so that tests can measure how
 * far each lies from the oracle and hold the GPU frame against EVERY one of them (tests/test_oracle_variants.py,
 * tests/test_parity_gpu.py::test_gpu_within_every_opencv_variant):
 *   row_mode 0  RowFilter, taps left to right, fused multiply-add (AVX2/FMA3 dispatch: v_muladd)      = the oracle
 *            1  RowFilter, taps left to right, multiply then add (SSE baseline)
 *            2  SymmRowSmallFilter for ksize <= 5 (x0*k0 + (x-1 + x+1)*k1 [+ (x-2 + x+2)*k2]), FMA; else as mode 0
 *            3  the same without FMA; else as mode 1
 *   col_mode 0  ColumnFilter, taps top to bottom, FMA                                                  = the oracle
 *            1  SymmColumnFilter: centre tap first, then (S[+k] + S[-k]) * f[k] outwards, FMA
 *            2  SymmColumnFilter without FMA
 *            3  ColumnFilter, taps top to bottom, multiply then add
 * (The IPP path of the pip wheels, ippiFilterGaussianBorder, is closed arithmetic and cannot be restated.) */
int orc_sepblur_f32_variant(const float* src, float* dst, float* tmp, int h, int w, int cn,
                            const float* kx, int nkx, const float* ky, int nky, int row_mode, int col_mode)
{
    const int rx = nkx / 2, ry = nky / 2;
    const int row_small = (row_mode >= 2) && nkx <= 5 && nkx >= 3;
    const int row_fma = (row_mode == 0 || row_mode == 2);
    for (int y = 0; y < h; ++y) {
        const float* srow = src + (size_t)y * w * cn;
        float* trow = tmp + (size_t)y * w * cn;
        for (int x = 0; x < w; ++x) {
            for (int c = 0; c < cn; ++c) {
                float s;
#define SX(d) srow[(size_t)clampi(x + (d), 0, w - 1) * cn + c]
                if (row_small) {
                    if (row_fma) {
                        s = SX(0) * kx[rx];
                        for (int k = 1; k <= rx; ++k) s = fmaf(SX(-k) + SX(k), kx[rx + k], s);
                    } else {
                        s = SX(0) * kx[rx];
                        for (int k = 1; k <= rx; ++k) { const float t = (SX(-k) + SX(k)) * kx[rx + k]; s = s + t; }
                    }
                } else if (row_fma) {
                    s = 0.0f;
                    for (int k = 0; k < nkx; ++k) s = fmaf(SX(k - rx), kx[k], s);
                } else {
                    s = SX(-rx) * kx[0];
                    for (int k = 1; k < nkx; ++k) { const float t = SX(k - rx) * kx[k]; s = s + t; }
                }
#undef SX
                trow[(size_t)x * cn + c] = s;
            }
        }
    }
    const size_t rowlen = (size_t)w * cn;
    for (int y = 0; y < h; ++y) {
        float* drow = dst + (size_t)y * rowlen;
#define TR(d) (tmp + (size_t)clampi(y + (d), 0, h - 1) * rowlen)
        if (col_mode == 0) {
            for (size_t i = 0; i < rowlen; ++i) drow[i] = 0.0f;
            for (int k = 0; k < nky; ++k) { const float* t = TR(k - ry); const float f = ky[k]; for (size_t i = 0; i < rowlen; ++i) drow[i] = fmaf(t[i], f, drow[i]); }
        } else if (col_mode == 3) {
            { const float* t = TR(-ry); const float f = ky[0]; for (size_t i = 0; i < rowlen; ++i) drow[i] = t[i] * f; }
            for (int k = 1; k < nky; ++k) { const float* t = TR(k - ry); const float f = ky[k]; for (size_t i = 0; i < rowlen; ++i) { const float p = t[i] * f; drow[i] = drow[i] + p; } }
        } else {
            { const float* t = TR(0); const float f = ky[ry]; for (size_t i = 0; i < rowlen; ++i) drow[i] = t[i] * f; }
            for (int k = 1; k <= ry; ++k) {
                const float* ta = TR(k); const float* tb = TR(-k); const float f = ky[ry + k];
                if (col_mode == 1) for (size_t i = 0; i < rowlen; ++i) drow[i] = fmaf(ta[i] + tb[i], f, drow[i]);
                else for (size_t i = 0; i < rowlen; ++i) { const float p = (ta[i] + tb[i]) * f; drow[i] = drow[i] + p; }
            }
        }
#undef TR
    }
    return 0;
}

/* cvRound for float: round half to even (SSE cvtss2si under the default MXCSR). */
static inline int cv_round_f(float v) { return (int)lrintf(v); }

static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

/* Integer part / 5-bit fractions of the quantised map — exported so tests can check the
 * GPU's tap indices bit-for-bit ("bit-exact for integer mask indexing"). */
int orc_remap_quantise(const float* mapx, const float* mapy, int n, int32_t* ix, int32_t* iy, int32_t* fxy)
{
    for (int i = 0; i < n; ++i) {
        int sx = cv_round_f(mapx[i] * 32.0f);
        int sy = cv_round_f(mapy[i] * 32.0f);
        ix[i] = sat_short(sx >> 5);
        iy[i] = sat_short(sy >> 5);
        fxy[i] = (sy & 31) * 32 + (sx & 31);
    }
    return 0;
}

#define REMAP_BODY(T, WT)                                                                      \
    for (int y = 0; y < h; ++y) {                                                              \
        for (int x = 0; x < w; ++x) {                                                          \
            size_t m = (size_t)y * w + x;                                                      \
            int sxq = cv_round_f(mapx[m] * 32.0f);                                             \
            int syq = cv_round_f(mapy[m] * 32.0f);                                             \
            int sx = sat_short(sxq >> 5), sy = sat_short(syq >> 5);                            \
            float fx = (float)(sxq & 31) * (1.0f / 32.0f);                                     \
            float fy = (float)(syq & 31) * (1.0f / 32.0f);                                     \
            float wgt[4];                                                                      \
            wgt[0] = (1.0f - fy) * (1.0f - fx);                                                \
            wgt[1] = (1.0f - fy) * fx;                                                         \
            wgt[2] = fy * (1.0f - fx);                                                         \
            wgt[3] = fy * fx;                                                                  \
            int in00 = (sx >= 0 && sx < w && sy >= 0 && sy < h);                               \
            int in01 = (sx + 1 >= 0 && sx + 1 < w && sy >= 0 && sy < h);                       \
            int in10 = (sx >= 0 && sx < w && sy + 1 >= 0 && sy + 1 < h);                       \
            int in11 = (sx + 1 >= 0 && sx + 1 < w && sy + 1 >= 0 && sy + 1 < h);               \
            for (int c = 0; c < cn; ++c) {                                                     \
                WT v0 = in00 ? (WT)src[((size_t)sy * w + sx) * cn + c] : (WT)0;                \
                WT v1 = in01 ? (WT)src[((size_t)sy * w + sx + 1) * cn + c] : (WT)0;            \
                WT v2 = in10 ? (WT)src[((size_t)(sy + 1) * w + sx) * cn + c] : (WT)0;          \
                WT v3 = in11 ? (WT)src[((size_t)(sy + 1) * w + sx + 1) * cn + c] : (WT)0;      \
                WT r = v0 * wgt[0] + v1 * wgt[1] + v2 * wgt[2] + v3 * wgt[3];                  \
                dst[m * cn + c] = (T)r;                                                        \
            }                                                                                  \
        }                                                                                      \
    }

int orc_remap_bilinear_f32(const float* src, float* dst, int h, int w, int cn,
                           const float* mapx, const float* mapy)
{
    REMAP_BODY(float, float)
    return 0;
}

int orc_remap_bilinear_f64(const double* src, double* dst, int h, int w, int cn,
                           const float* mapx, const float* mapy)
{
    REMAP_BODY(double, double)
    return 0;
}

/* Spread form of the same interpolation: the four products contracted into fused multiply-adds, as a compiler does
 * with -ffp-contract=fast on a baseline that has FMA (aarch64 wheels; x86-64 wheels build imgwarp.cpp for SSE3 and do
 * not).  r = fma(v3, w3, fma(v2, w2, fma(v1, w1, v0 * w0))). */
#undef REMAP_COMBINE
int orc_remap_bilinear_f64_fma(const double* src, double* dst, int h, int w, int cn, const float* mapx, const float* mapy)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            size_t m = (size_t)y * w + x;
            int sxq = cv_round_f(mapx[m] * 32.0f), syq = cv_round_f(mapy[m] * 32.0f);
            int sx = sat_short(sxq >> 5), sy = sat_short(syq >> 5);
            float fx = (float)(sxq & 31) * (1.0f / 32.0f), fy = (float)(syq & 31) * (1.0f / 32.0f);
            const double w0 = (1.0f - fy) * (1.0f - fx), w1 = (1.0f - fy) * fx, w2 = fy * (1.0f - fx), w3 = fy * fx;
            for (int c = 0; c < cn; ++c) {
#define TAPV(yy, xx) (((xx) >= 0 && (xx) < w && (yy) >= 0 && (yy) < h) ? src[((size_t)(yy) * w + (xx)) * cn + c] : 0.0)
                const double v0 = TAPV(sy, sx), v1 = TAPV(sy, sx + 1), v2 = TAPV(sy + 1, sx), v3 = TAPV(sy + 1, sx + 1);
#undef TAPV
                dst[m * cn + c] = fma(v3, w3, fma(v2, w2, fma(v1, w1, v0 * w0)));
            }
        }
    return 0;
}

/* cv2.resize INTER_NEAREST (resizeNN): fx = dw/sw in double, ifx = 1/fx, sx = min(cvFloor(dx*ifx), sw-1). */
int orc_resize_nearest_f32(const float* src, int sh, int sw, float* dst, int dh, int dw, int cn)
{
    const double ifx = 1.0 / ((double)dw / sw), ify = 1.0 / ((double)dh / sh);
    for (int y = 0; y < dh; ++y) {
        int sy = (int)floor(y * ify);
        if (sy > sh - 1) sy = sh - 1;
        for (int x = 0; x < dw; ++x) {
            int sx = (int)floor(x * ifx);
            if (sx > sw - 1) sx = sw - 1;
            memcpy(dst + ((size_t)y * dw + x) * cn, src + ((size_t)sy * sw + sx) * cn, sizeof(float) * cn);
        }
    }
    return 0;
}

/* cv2.resize INTER_LINEAR for CV_32F.
 * Exact 2x decimation (dw*2==sw && dh*2==sh) takes OpenCV's INTER_AREA fast path
 * (resizeAreaFast: (a+b+c+d)*0.25f).  Otherwise: fx = (dx+0.5)*scale-0.5, sx=floor(fx),
 * fx-=sx; sx<0 -> (0, fx=0); sx>=sw-1 -> (sw-1, fx=0); horizontal pass
 * S[sx]*(1-fx) + S[sx+1]*fx, then vertical pass r0*(1-fy) + r1*fy, all float. */
int orc_resize_linear_f32(const float* src, int sh, int sw, float* dst, int dh, int dw, int cn)
{
    if (dw * 2 == sw && dh * 2 == sh) {
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < cn; ++c) {
                    const float* p = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
                    const float* q = p + (size_t)sw * cn;
                    dst[((size_t)y * dw + x) * cn + c] = (p[0] + p[cn] + q[0] + q[cn]) * 0.25f;
                }
        return 0;
    }
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int* xofs = (int*)malloc(sizeof(int) * dw);
    float* xa = (float*)malloc(sizeof(float) * dw);
    float* rows = (float*)malloc(sizeof(float) * 2 * (size_t)dw * cn);
    if (!xofs || !xa || !rows) { free(xofs); free(xa); free(rows); return -1; }
    for (int x = 0; x < dw; ++x) {
        float fx = (float)((x + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[x] = sx; xa[x] = fx;
    }
    for (int y = 0; y < dh; ++y) {
        float fy = (float)((y + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        int sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
        for (int r = 0; r < 2; ++r) {
            const float* srow = src + (size_t)(r ? sy1 : sy) * sw * cn;
            float* rr = rows + (size_t)r * dw * cn;
            for (int x = 0; x < dw; ++x) {
                int sx = xofs[x];
                int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
                float a1 = xa[x], a0 = 1.0f - a1;
                for (int c = 0; c < cn; ++c)
                    rr[(size_t)x * cn + c] = srow[(size_t)sx * cn + c] * a0 + srow[(size_t)sx1 * cn + c] * a1;
            }
        }
        const float b1 = fy, b0 = 1.0f - fy;
        float* drow = dst + (size_t)y * dw * cn;
        for (size_t i = 0; i < (size_t)dw * cn; ++i) drow[i] = rows[i] * b0 + rows[(size_t)dw * cn + i] * b1;
    }
    free(xofs); free(xa); free(rows);
    return 0;
}

/* ALTERNATIVE FORM of the same resize (not the oracle: the spread around it, see orc_sepblur_f32_variant): mode 1 = the two
 * linear passes with a fused multiply-add (an FMA build contracts S0*a0 + S1*a1), and the 2x2 mean of the exact-decimation
 * fast path summed pairwise, (p00 + p01) + (p10 + p11), as a SIMD resizeAreaFast adds its row vectors. */
int orc_resize_linear_f32_variant(const float* src, int sh, int sw, float* dst, int dh, int dw, int cn, int mode)
{
    if (dw * 2 == sw && dh * 2 == sh) {
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < cn; ++c) {
                    const float* p = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
                    const float* q = p + (size_t)sw * cn;
                    dst[((size_t)y * dw + x) * cn + c] = mode ? ((p[0] + p[cn]) + (q[0] + q[cn])) * 0.25f : (p[0] + p[cn] + q[0] + q[cn]) * 0.25f;
                }
        return 0;
    }
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int* xofs = (int*)malloc(sizeof(int) * dw);
    float* xa = (float*)malloc(sizeof(float) * dw);
    float* rows = (float*)malloc(sizeof(float) * 2 * (size_t)dw * cn);
    if (!xofs || !xa || !rows) { free(xofs); free(xa); free(rows); return -1; }
    for (int x = 0; x < dw; ++x) {
        float fx = (float)((x + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[x] = sx; xa[x] = fx;
    }
    for (int y = 0; y < dh; ++y) {
        float fy = (float)((y + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        int sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
        for (int r = 0; r < 2; ++r) {
            const float* srow = src + (size_t)(r ? sy1 : sy) * sw * cn;
            float* rr = rows + (size_t)r * dw * cn;
            for (int x = 0; x < dw; ++x) {
                int sx = xofs[x];
                int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
                float a1 = xa[x], a0 = 1.0f - a1;
                for (int c = 0; c < cn; ++c)
                    rr[(size_t)x * cn + c] = mode ? fmaf(srow[(size_t)sx1 * cn + c], a1, srow[(size_t)sx * cn + c] * a0) : srow[(size_t)sx * cn + c] * a0 + srow[(size_t)sx1 * cn + c] * a1;
            }
        }
        const float b1 = fy, b0 = 1.0f - fy;
        float* drow = dst + (size_t)y * dw * cn;
        for (size_t i = 0; i < (size_t)dw * cn; ++i) drow[i] = mode ? fmaf(rows[(size_t)dw * cn + i], b1, rows[i] * b0) : rows[i] * b0 + rows[(size_t)dw * cn + i] * b1;
    }
    free(xofs); free(xa); free(rows);
    return 0;
}

/* cv2.resize INTER_LINEAR for CV_64F (the persistence state of a promoted chain, ref:690): same
 * offsets and FLOAT coefficients as the 32F path (the alpha/beta tables are float for every depth),
 * arithmetic in double (HResizeLinear<double,double,float>, VResizeLinear<double,double,float>);
 * exact 2x decimation goes through resizeAreaFast with a double work type. */
int orc_resize_linear_f64(const double* src, int sh, int sw, double* dst, int dh, int dw, int cn)
{
    if (dw * 2 == sw && dh * 2 == sh) {
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < cn; ++c) {
                    const double* p = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
                    const double* q = p + (size_t)sw * cn;
                    dst[((size_t)y * dw + x) * cn + c] = (p[0] + p[cn] + q[0] + q[cn]) * 0.25;
                }
        return 0;
    }
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    for (int y = 0; y < dh; ++y) {
        float fy = (float)((y + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        const int sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
        const float b1 = fy, b0 = 1.0f - fy;
        for (int x = 0; x < dw; ++x) {
            float fx = (float)((x + 0.5) * scale_x - 0.5);
            int sx = (int)floorf(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            const float a1 = fx, a0 = 1.0f - fx;
            for (int c = 0; c < cn; ++c) {
                const double r0 = src[((size_t)sy * sw + sx) * cn + c] * (double)a0 + src[((size_t)sy * sw + sx1) * cn + c] * (double)a1;
                const double r1 = src[((size_t)sy1 * sw + sx) * cn + c] * (double)a0 + src[((size_t)sy1 * sw + sx1) * cn + c] * (double)a1;
                dst[((size_t)y * dw + x) * cn + c] = r0 * (double)b0 + r1 * (double)b1;
            }
        }
    }
    return 0;
}

/* cv2.convertScaleAbs(alpha=255, beta=0): the 32f and 64f sources both go through the float
 * work type (cvtabs_32f): u8 = saturate(cvRound(|(float)x * 255.0f + 0|)). */
int orc_convert_scale_abs_f32(const float* src, uint8_t* dst, size_t n, float alpha)
{
    for (size_t i = 0; i < n; ++i) {
        float v = fabsf(fmaf(src[i], alpha, 0.0f));
        int r = cv_round_f(v);
        dst[i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
    return 0;
}

int orc_convert_scale_abs_f64(const double* src, uint8_t* dst, size_t n, float alpha)
{
    for (size_t i = 0; i < n; ++i) {
        float v = fabsf(fmaf((float)src[i], alpha, 0.0f));
        int r = cv_round_f(v);
        dst[i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
    return 0;
}

/* Spread form: the scalar tail of cvtabs_32f for a CV_64F source multiplies in double (src[j] * a with a float a,
 * no narrowing first): u8 = saturate(cvRound(|x * 255.0|)) with the double rounded to nearest-even by cvRound(double). */
int orc_convert_scale_abs_f64_dbl(const double* src, uint8_t* dst, size_t n, float alpha)
{
    for (size_t i = 0; i < n; ++i) {
        double v = fabs(src[i] * (double)alpha);
        long r = lrint(v);
        dst[i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
    return 0;
}

/* cv2.addWeighted(a, alpha, b, beta, 0): dst = fma(a, alpha, fma(b, beta, gamma)) in the
 * array's own type (scalars narrowed to float for CV_32F). crt_filter.py:693. */
int orc_add_weighted_f32(const float* a, float alpha, const float* b, float beta, float* dst, size_t n)
{
    for (size_t i = 0; i < n; ++i) dst[i] = fmaf(a[i], alpha, fmaf(b[i], beta, 0.0f));
    return 0;
}

int orc_add_weighted_f64(const double* a, double alpha, const double* b, double beta, double* dst, size_t n)
{
    for (size_t i = 0; i < n; ++i) dst[i] = fma(a[i], alpha, fma(b[i], beta, 0.0));
    return 0;
}

int orc_abi_version(void) { return 3; }
