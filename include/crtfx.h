/*
 * crtfx.h — C ABI of libcrtfx.so: the MI355X (gfx950) implementation of the per-frame CRT
 * effect chain of jaylikesbunda/PythonCRT.
 *
 * The reference has no FFI: its de-facto boundary is two Python functions plus the mask
 * builders and the in-order persistence/quantise slice of process_video (all in
 * crt_filter.py, cited below as ref:LINE).  Each entry point names what it replaces.
 *
 * Conventions
 *   - every function returns 0 (CRTFX_OK) or a negative crtfx_status; the message is in
 *     crtfx_last_error(ctx) (thread-safe per ctx, not across ctxs sharing a thread-unsafe caller).
 *   - the caller owns every frame / state / mask / noise buffer (device pointers, e.g.
 *     tensor.data_ptr()); the library never frees or retains them beyond the stream work it
 *     enqueues.  Host tables passed to crtfx_set_params are copied before it returns.
 *   - all device work is enqueued on the caller's hipStream_t (passed as void*, NULL = the
 *     default stream); no entry point except create/destroy/set_params synchronises.  The calling
 *     thread's current HIP device must be the ctx's device (CRTFX_E_INVALID otherwise).
 *   - one ctx per (device, frame size, caller thread).  Distinct ctxs may be used concurrently
 *     (the reference calls apply_static_effects from 2 worker threads, ref:1015-1017, and
 *     apply_crt_effect from the GUI thread, ref:1810).
 *   - frames are H x W x 3 interleaved RGB, C-contiguous: uint8 (ref:489,502,1036) or, for
 *     CRTFX_PIX_F16, IEEE half on the same 0..255 scale.  Float images / state are float32 H x W x 3.
 */
#ifndef CRTFX_H
#define CRTFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRTFX_ABI_VERSION 1

typedef enum crtfx_status {
    CRTFX_OK = 0,
    CRTFX_E_INVALID = -1,      /* bad argument / params not set */
    CRTFX_E_HIP = -2,          /* a HIP runtime call failed */
    CRTFX_E_UNSUPPORTED = -3,  /* valid in the reference but outside this build's limits */
    CRTFX_E_NOMEM = -4
} crtfx_status;

typedef enum crtfx_pixfmt { CRTFX_PIX_U8 = 0, CRTFX_PIX_F16 = 1 } crtfx_pixfmt;

/* crtfx_params.flags — which stages of the chain run (gates as in ref:571-652 / :740-822). */
#define CRTFX_F_SATURATION   (1u << 0)   /* saturation != 1.0            ref:288 */
#define CRTFX_F_TEMPERATURE  (1u << 1)   /* temperature != 0.0           ref:292 */
#define CRTFX_F_BRIGHTCON    (1u << 2)   /* brightness != 0 or contrast != 1   ref:299 */
#define CRTFX_F_GAMMA        (1u << 3)   /* gamma != 1 and gamma > 0     ref:302 */
#define CRTFX_F_BLOOM        (1u << 4)   /* Gaussian bloom               ref:599,608-611 */
#define CRTFX_F_BLOOM_FAST   (1u << 5)   /* half-res bilinear bloom      ref:605-607 */
#define CRTFX_F_BLOOM_THR    (1u << 6)   /* bloom_threshold > 0          ref:602-604 */
#define CRTFX_F_TRIAD        (1u << 7)   /* triad mask present           ref:613 */
#define CRTFX_F_TRIAD_LUT    (1u << 8)   /* gamma-linearised path ref:246-262 (else plain multiply ref:240-245) */
#define CRTFX_F_TRIAD_LUMA   (1u << 9)   /* preserve luma                ref:253-259 */
#define CRTFX_F_SCANLINES    (1u << 10)  /* scanline_strength > 0        ref:617 */
#define CRTFX_F_VIGNETTE     (1u << 11)  /* vignette mask present        ref:626 */
#define CRTFX_F_FLICKER      (1u << 12)  /* flicker_strength>0 and hz>0  ref:630 */
#define CRTFX_F_NOISE        (1u << 13)  /* noise_strength > 0           ref:635 */
#define CRTFX_F_WARP         (1u << 14)  /* warp_strength != 0           ref:649 */
#define CRTFX_F_PIXELATE     (1u << 15)  /* pixel_size > 1               ref:578 */

/* Effect parameters.  Scalars are the values the reference's kwargs resolve to; tables are
 * HOST pointers copied by crtfx_set_params (the Python host builds them with the same numpy
 * expressions the reference uses, so LUT entries and row gains are the machine's own numpy
 * values); *_dev members are optional DEVICE pointers the caller keeps alive. */
typedef struct crtfx_params {
    uint32_t size;               /* = sizeof(crtfx_params); versions the struct */
    uint32_t flags;              /* CRTFX_F_* */
    int32_t  aberration_px;      /* ref:571-577: R from x-d, B from x+d, wrap-around */
    int32_t  grain_size;         /* ref:637 (1 = per-pixel grain) */
    int32_t  bloom_radius;       /* (ksize-1)/2 with ksize = max(1, round(3 sigma)*2+1)  ref:609; any radius in [0, 65536] (the reference takes any sigma) */
    int32_t  reserved0;
    float    saturation;         /* ref:290 */
    float    r_gain, b_gain;     /* clip(1 +/- 0.5 t, 0.5, 1.5)  ref:294-295 */
    float    contrast, brightness; /* ref:300 */
    float    inv_gamma;          /* 1/gamma  ref:303 */
    float    bloom_thr;          /* min(0.99, max(0, thr))  ref:603 */
    float    bloom_thr_den;      /* max(1e-6, 1 - thr)      ref:604 */
    float    bloom_strength;     /* ref:611 */
    float    noise_scale;        /* noise_strength / 255.0  ref:646 */
    float    warp_k;             /* 0.5 * warp_strength     ref:342 */
    float    warp_cx, warp_cy;   /* (W-1)/2, (H-1)/2        ref:336-337 */
    double   vignette_strength;  /* analytic vignette: v = 1 - s*clip(nx2[x]+ny2[y],0,1)  ref:275 */
    /* host tables */
    const float*  bloom_taps;    /* 2*bloom_radius+1 Gaussian taps (cv::getGaussianKernel, float) */
    const float*  triad_row;     /* W*3: one row of the triad mask (rows are identical, ref:230,234) or NULL */
    const float*  lut_g;         /* 1025: linspace(0,1)^gamma      ref:248-249 */
    const float*  lut_inv;       /* 1025: linspace(0,1)^(1/gamma)  ref:260 */
    const double* vig_nx2;       /* W: ((x-cx)/rx)^2   ref:272,274 */
    const double* vig_ny2;       /* H: ((y-cy)/ry)^2   ref:273,274 */
    const float*  warp_xhat;     /* W: (x-cx)/max(1,cx) float32   ref:338 */
    const float*  warp_yhat;     /* H: (y-cy)/max(1,cy) float32   ref:339 */
    const int32_t* pix_xmap;     /* W: source column of the nearest down/up pair  ref:582-583, or NULL */
    const int32_t* pix_ymap;     /* H */
    /* optional device-resident full masks (arbitrary arrays handed to apply_*) */
    const float*  triad_full_dev;    /* H*W*3 float32 */
    const double* vignette_full_dev; /* H*W   float64 */
    /* cv2.resize(INTER_LINEAR) axes, host tables: per destination index the source index of the first
     * tap (the second is +1, clamped) and the weight of the second tap, as OpenCV computes them
     * (fx = (float)((dx+0.5)*scale - 0.5); sx = floor(fx); fx -= sx; clamped at both ends). */
    const int32_t* grain_xofs; const float* grain_xw;   /* W: (H//g x W//g) grain plane -> frame   ref:642 */
    const int32_t* grain_yofs; const float* grain_yw;   /* H */
    int32_t grain_w, grain_h;                           /* max(1, W//g), max(1, H//g)              ref:638-639 */
    const int32_t* fbu_xofs; const float* fbu_xw;       /* W: fast bloom, half-res -> frame        ref:607 */
    const int32_t* fbu_yofs; const float* fbu_yw;       /* H */
    const int32_t* fbd_xofs; const float* fbd_xw;       /* max(1,W//2): frame -> half-res          ref:606; NULL = exact 2x */
    const int32_t* fbd_yofs; const float* fbd_yw;       /* max(1,H//2)    decimation (OpenCV's INTER_AREA 2x2 mean) */
    /* Optional (uint8 frames, saturation == 1): a1 + a4 tabulated per channel and code, 3 x 256 float32 = the value
     * ref:569 + ref:292-304 give channel c of a pixel whose stored sample is u (temperature gain, brightness /
     * contrast, gamma are all per-channel functions of the sample once the saturation mix is off).  Built on the
     * host with the reference's numpy expressions; replaces the per-pixel arithmetic (three powf with --gamma). */
    const float*  grade_lut;
} crtfx_params;

/* Per-frame inputs (everything that changes from frame to frame; ref:1043,1064). */
typedef struct crtfx_frame {
    const float*   scan_row_dev;     /* H float32 row gains (make_scanline_mask_dynamic, ref:213-217) or NULL */
    const float*   scan_plane_dev;   /* H*W float32 (make_scanline_mask_2d, ref:308-328) or NULL */
    const float*   noise_plane_dev;  /* N(0,1) float32, H*W (or (H/g)*(W/g) for grain_size g); NULL = counter-based RNG */
    const uint8_t* overlay_rgba_dev; /* H*W*4 uint8 text overlay (ref:588-598 / 653-663) or NULL */
    const int32_t* glitch_offs_dev;  /* glitch row offsets (ref:679-682 / 853-855) or NULL */
    double   flicker_factor;         /* 1 + 0.25 fs sin(2 pi hz t)  ref:632 */
    uint64_t noise_seed;             /* RNG stream key */
    uint64_t frame_index;            /* RNG counter high part */
    int32_t  overlay_after;          /* ref:588 vs :653 */
    int32_t  glitch_y0;              /* first row of the glitch band  ref:667 */
    int32_t  glitch_cols;            /* offsets per band row: 1 (per-row, preview ref:682), W (per-pixel, render ref:855), or the
                                      * number of segments when glitch_seg_len > 0 */
    int32_t  glitch_seg_len;         /* > 0: pixel x takes offsets[row * glitch_cols + x / glitch_seg_len] (the render variant's
                                      * per-segment offsets, ref:843-852, before their expansion to pixels); 0: see glitch_cols */
} crtfx_frame;

/* Persistence blend flavours. */
typedef enum crtfx_blend {
    CRTFX_BLEND_NONE = 0,     /* first frame / persistence 0: state = static  (ref:1094-1096) */
    CRTFX_BLEND_RENDER = 1,   /* clip(p*prev + (1-p)*static, 0, 1)            (ref:1092)      */
    CRTFX_BLEND_PREVIEW = 2   /* cv2.addWeighted(prev, p, static, 1-p, 0)     (ref:693)       */
} crtfx_blend;

typedef struct crtfx_ctx crtfx_ctx;

int crtfx_version(void);

/* ctx owns device-side tables and the H*W*3 float32 pre-warp scratch image.  crtfx_create, crtfx_destroy and
 * crtfx_set_params do their work on the ctx's device and restore the calling thread's current device before they
 * return; every other entry point launches on the CURRENT device and refuses (CRTFX_E_INVALID) when that is not the
 * ctx's. */
int crtfx_create(int device, int height, int width, int pix_fmt, crtfx_ctx** out_ctx);
int crtfx_destroy(crtfx_ctx* ctx);
const char* crtfx_last_error(const crtfx_ctx* ctx);

/* Replaces the keyword set of apply_crt_effect / apply_static_effects (ref:531-565, :702-734)
 * and the once-per-render mask construction (ref:919-920). Synchronises the device. */
int crtfx_set_params(crtfx_ctx* ctx, const crtfx_params* params);

/* apply_static_effects (ref:702-861): uint8/half frame -> float32 H*W*3 static image. */
int crtfx_apply_static(crtfx_ctx* ctx, const void* frame_dev, float* out_float_dev,
                       const crtfx_frame* frame, void* stream);

/* apply_crt_effect (ref:531-699) and, with blend RENDER, apply_static_effects followed by the
 * in-order commit of process_video (ref:1086-1098): chain + persistence + convertScaleAbs.
 * state_inout_dev may be NULL when blend is NONE and the caller does not want the float state;
 * out_float_dev (optional) receives the unblended static image. */
int crtfx_apply(crtfx_ctx* ctx, const void* frame_dev, void* out_pix_dev, float* state_inout_dev,
                float* out_float_dev, int blend, double persistence, const crtfx_frame* frame,
                void* stream);

/* The commit step alone (ref:1086-1098): state = blend(state, static); out = quantise(state). */
int crtfx_blend_quantise(crtfx_ctx* ctx, const float* static_dev, float* state_inout_dev,
                         void* out_pix_dev, int blend, double persistence, void* stream);

/* Frame-sharded persistence: local = state computed from a zero incoming state;
 * out = quantise(clip(local + coeff * carry_in)), coeff = p^(frames since the chunk start). */
int crtfx_halo_correct_quantise(crtfx_ctx* ctx, const float* local_dev, const float* carry_in_dev,
                                double coeff, float* state_out_dev, void* out_pix_dev, void* stream);

/* The same fix-up for the first n frames of a chunk in one launch: frame j (local state at local_base_dev +
 * j*H*W*3 floats) is re-quantised with coeff = persistence^(first_power + j) into out_base + j*out_stride_bytes.
 * The caller may stop at the frame where persistence^k drops below float32 resolution (shard.settle_frames). */
int crtfx_halo_correct_batch(crtfx_ctx* ctx, const float* local_base_dev, const float* carry_in_dev, double persistence,
                             int first_power, int n, void* out_base, size_t out_stride_bytes, void* stream);

/* A run of n frames back to back (the body of the loop at ref:1037-1131): frame i is read at
 * frames_base + i*frame_stride_bytes and written at out_base + i*out_stride_bytes; `frames` is
 * a HOST array of n per-frame records.  persistence > 0 threads state_inout_dev through the
 * run (blend RENDER; the first frame passes through when first_has_state == 0).
 * local_states_base (optional) receives every frame's float state at stride H*W*3 floats (frame i blends against
 * state i-1 there and writes state i: no copy per frame; it must not overlap state_inout_dev, which receives the
 * last state at the end).  Consecutive frames may be executed as one grouped launch (several frames per grid; a run of frames
 * that blend with their predecessor keeps its state in registers): state_inout_dev is defined when the call's work has
 * completed on `stream`, not frame by frame. */
int crtfx_process_batch(crtfx_ctx* ctx, const void* frames_base, size_t frame_stride_bytes,
                        void* out_base, size_t out_stride_bytes, int n, const crtfx_frame* frames,
                        float* state_inout_dev, double persistence, int first_has_state,
                        float* local_states_base, void* stream);

/* make_scanline_mask_2d (ref:308-328) computed on the device into out_dev (H x W float32, usable as
 * crtfx_frame.scan_plane_dev): gain = 1 - strength * (0.5 * (1 + sin(omega * (y + tan_theta * x + phase_px)))) ^ inv_sharp
 * in float64, cast to float32.  The caller passes omega = 2 pi / max(1e-6, period), tan_theta = tan(deg2rad(angle)),
 * inv_sharp = 1 / clip(thickness, 0.1, 4) as the host computed them (ref:319-324). */
int crtfx_scanline_plane(crtfx_ctx* ctx, double strength, double omega, double phase_px, double tan_theta, double inv_sharp,
                         float* out_dev, void* stream);

/* cv2.resize(state_prev, (W, H), INTER_LINEAR) of ref:689-690: the previous persistence state has another
 * size than this ctx's frames (the preview window was resized between ticks).  src_dev: src_h x src_w x 3
 * float32; dst_dev: H x W x 3 float32, then usable as state_inout_dev of crtfx_apply. */
int crtfx_resize_state(crtfx_ctx* ctx, const float* src_dev, int src_h, int src_w, float* dst_dev, void* stream);

/* The N(0,1) plane the in-kernel counter-based RNG draws for (seed, frame_index): lets a test
 * feed the identical grain to the CPU oracle (cv2.randn, ref:641,645, is unreproducible). */
int crtfx_noise_plane(crtfx_ctx* ctx, uint64_t seed, uint64_t frame_index, float* out_dev, void* stream);

/* The fixed-point sampling map of the barrel warp (ref:338-347 + cv2.remap's 1/32-px
 * quantisation): integer tap origin and packed (fy<<5|fx) fraction per output pixel. */
int crtfx_warp_map(crtfx_ctx* ctx, int32_t* ix_dev, int32_t* iy_dev, int32_t* fxy_dev, void* stream);

/* Testing / tuning switches of one ctx (the product path never needs them; nothing in the library reads the
 * environment).  Call crtfx_set_params again afterwards: launch shapes are planned there.
 *   FORCE_GENERIC        always take the general-purpose kernels (the LDS-ring k_phosphor, k_point, k_warp)
 *   FORCE_RUNTIME_FLAGS  never take a gate-folded instantiation
 *   NO_CC                full-chain launches that park a pre-warp image stay on k_phosphor_rr (A/B against k_phosphor_cc)
 *   FORCE_CC             ... take the column-owner family for every radius (k_phosphor_ct where it is built — uint8 frames, radii 1..15 —
 *                        unless NO_CT is set, k_phosphor_cc elsewhere); by default k_phosphor_cc serves radii 16..30 only
 *   NO_CT                ... stay on k_phosphor_cc instead of its composite-table build k_phosphor_ct (A/B)
 *   GROUP, SEG_ROWS      frames per grid (1..8 = CRTFX_MAX_GROUP) / rows per block of the register-window kernels; 0 = the planner's choice
 *   WARP_ROWS            output rows per k_warp_lean thread (1, 2, 4; 0 = the launcher's choice: 4, or 2 with a persistence chain)
 *   NO_PLAIN_WARP        unblended uint8 frames behind a warp stay on k_warp_lean's general build instead of its branch-free one (A/B)
 *   BAND_MB              N > 0: a frame whose float32 pre-warp image exceeds N MiB (an 8K frame is 398 MB; 224 keeps a band under the
 *                        256 MiB Infinity Cache) runs as bands of rows, each k_phosphor band followed at once by the k_warp_lean rows
 *                        whose taps it completes; 0 / -1 (default) = whole-frame launches — at 8K the bands measured +0.5 %
 *                        (profiles/r04_8k_bands.txt), so they stay an A/B switch; tests band small frames with 1
 *   POINT_TILES          rows per k_point block (1..16; 0 = default)
 *   OVERLAP              run k_warp(n) on a side stream beside k_phosphor(n+1) (measured slower; kept for A/B)
 *   SPLIT_FROM           Gaussian-bloom radii >= this run the split path (blur kernels of any radius + the pointwise chain)
 *                        instead of a fused build; default 31 (one fused build per radius up to 30), 0 = every radius
 *   SPLIT_SRC_PLANE      the split path writes the bloom source as a plane first instead of grading while it stages (A/B)
 *   DEBUG_PLAN           print the planned launch shape to stderr
 *   NO_FUSED_HALF        the fast-bloom render chain as two launches (k_half_group writes the half-resolution bloom source, k_point_lean_seq reads
 *                        it) instead of k_point_fused_seq, which forms the source in LDS (tests, A/B) */
typedef enum crtfx_option {
    CRTFX_OPT_FORCE_GENERIC = 1, CRTFX_OPT_FORCE_RUNTIME_FLAGS = 2, CRTFX_OPT_NO_CC = 3, CRTFX_OPT_GROUP = 4, CRTFX_OPT_SEG_ROWS = 5,
    CRTFX_OPT_WARP_ROWS = 6, CRTFX_OPT_POINT_TILES = 7, CRTFX_OPT_OVERLAP = 8, CRTFX_OPT_DEBUG_PLAN = 9, CRTFX_OPT_FORCE_CC = 10,
    CRTFX_OPT_SPLIT_FROM = 11, CRTFX_OPT_SPLIT_SRC_PLANE = 12, CRTFX_OPT_NO_CT = 13, CRTFX_OPT_NO_PLAIN_WARP = 14, CRTFX_OPT_BAND_MB = 15,
    CRTFX_OPT_NO_FUSED_HALF = 16
} crtfx_option;
int crtfx_set_option(crtfx_ctx* ctx, int option, int value);

/* Which build of each kernel class the most recent crtfx_apply_static / crtfx_apply / crtfx_process_batch call of this ctx launched, as a
 * NUL-terminated `key=value;...` string (truncated to n - 1 characters), e.g.
 *   phosphor=k_phosphor_ct<9,u8>;group=2;seg_rows=256;group_max=2;warp=k_warp_lean<f64,none,u8,rows=4,tile=128x8,plain>;warp_frames=2
 * keys: phosphor (the fused Gaussian-bloom chain), blur (split bloom passes), half (fast-bloom source), point (pointwise chain), group (frames
 * per grid of the phosphor / pointwise launch), seg_rows (rows per phosphor block), group_max (the planner's frames per grid), warp, warp_frames.
 * crtfx_process_batch reports its FULL-SIZE launch group — the one with the most frames, the later of equals: a batch that is not a multiple of
 * the group size ends in one shorter group with a launch shape of its own, which says nothing about the other launches of the call (a batch of
 * 8192 1080p frames = 1638 groups of 5 frames x 168-row blocks + one of 2 x 64: the plan reads group=5;seg_rows=168).  Every variant of a
 * kernel produces the same bits (tests/test_parity_gpu.py::test_kernel_variants_agree), so a planner regression is invisible to parity tests:
 * this record is what tests/test_plan_gpu.py pins for the BASELINE configs.  (The reference has no counterpart: its "plan" is the fixed
 * numpy / cv2 call sequence of ref:566-698.) */
int crtfx_last_plan(crtfx_ctx* ctx, char* buf, size_t n);

/* Host-side query (no GPU work, no ctx): the dynamic LDS bytes one block of a column-owner phosphor build asks for at launch —
 * build = "k_phosphor_ct" (radii 1..15; pix_fmt CRTFX_PIX_U8 or CRTFX_PIX_F16) or "k_phosphor_cc" (radii 1..30).  Their launch shape is
 * planned for FOUR resident blocks per CU (160 KB of LDS per CU: <= 40 960 bytes per block), and the size is not in the code-object
 * metadata (`extern __shared__`); tests/test_evidence_tools.py pins it next to the register counts it reads from the library's code
 * objects.  Returns the byte count, CRTFX_E_UNSUPPORTED for a radius the build does not serve, CRTFX_E_INVALID for a bad argument.
 * (No reference counterpart.) */
int crtfx_kernel_lds_bytes(const char* build, int radius, int pix_fmt);

/* -DCRTFX_STAMP diagnostic builds only (tools/phase_profile.py): device buffer the per-wave phase cycle sums are
 * written to.  CRTFX_E_UNSUPPORTED in the product build. */
int crtfx_debug_buffer(crtfx_ctx* ctx, void* dev_ptr);

/* HIP-event timing of the launches while profiling is on (events attached to the dispatch packets):
 * on = 0 off, 1 every frame, N > 1 every N-th frame (sampling keeps the overhead negligible).
 * kernel 0 = phosphor / pointwise chain (grade+bloom+masks+grain), 1 = warp/commit, 2 = the bloom passes launched in
 * front of a pointwise kernel (k_half / k_half_group, the split bloom's row and column passes).  Returns the mean duration of the timed
 * launches (what `rocprofv3 --kernel-trace --stats` reports as AverageNs), their count, and (frames may be NULL)
 * the number of frames they covered: crtfx_process_batch puts several frames into one grid. */
int crtfx_profile_enable(crtfx_ctx* ctx, int on);
int crtfx_profile_read(crtfx_ctx* ctx, int kernel, double* mean_launch_ms, int* launches, int* frames);

/* Host-side helper (no GPU work): horizontal Gaussian softening of ONE mask row with
 * BORDER_REPLICATE, taps accumulated in order with fmaf — the (k,1) cv2.GaussianBlur of
 * make_triad_mask (ref:231-234) applied to the only distinct row.  row_in/row_out: w*cn floats. */
int crtfx_host_blur_row(const float* row_in, float* row_out, int w, int cn, const float* taps, int ntaps);

#ifdef __cplusplus
}
#endif
#endif /* CRTFX_H */
