// crtfx.hip — libcrtfx.so: C-ABI (include/crtfx.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see csrc/build.py).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <vector>

#define CRTFX_MAIN_TU 1   // this TU owns the non-template kernels of crtfx_kernels.hip.h
#include "crtfx.h"
#include <cmath>
#include "crtfx_internal.h"

using namespace crtfx;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

// Gaussian bloom radii from here on run the split path (three blur kernels + the pointwise chain) instead of a fused
// register-window build: see k_sb_rows in crtfx_blur.hip.h and profiles/r02_sigma_sweep.txt for the crossover.
constexpr int SPLIT_FROM_RADIUS = 31;
constexpr int SPLIT_MAX_RADIUS = 1 << 16;     // sigma ~ 21845: far beyond any frame size; a sanity bound on the tap array only

struct crtfx_ctx {
    int device = 0;
    int H = 0, W = 0;
    int pix_fmt = CRTFX_PIX_U8;
    bool params_set = false;
    KParams kp{};
    DevBuf triad_row, lut_g, lut_inv, nx2, ny2, xhat, yhat, xmap, ymap, glut, consts;
    DevBuf gxo, gxw, gyo, gyw, uxo, uxw, uyo, uyw, dxo, dxw, dyo, dyw, ds;   // resize axes, half-res scratch (split bloom: two full-res planes)
    DevBuf tpad;                     // split bloom: zero-padded taps (sb_tpad_len)
    DevBuf tcomp;                    // composite triad tables of k_phosphor_ct: [2][LUT_N]
    bool split = false;              // this parameter set runs the Gaussian bloom as k_sb_src / k_sb_rows / k_sb_cols + the pointwise kernels
    int split_R = 0;                 // its radius (kp.R is 0 then)
    bool split_src_plane = false;    // CRTFX_OPT_SPLIT_SRC_PLANE
    int split_from = SPLIT_FROM_RADIUS;   // CRTFX_OPT_SPLIT_FROM: radii >= this take the split path (0 = every radius)
    float* pre = nullptr;            // pre_frames x H*W*3 float32 pre-warp scratch
    int pre_frames = 1;
    int group_max = 1;               // frames per grouped launch (fills the block slots at small frame sizes)
    // a frame whose float32 pre-warp image does not fit the Infinity Cache is produced and consumed in BANDS of row segments
    // (crtfx_set_params plans them; empty = whole frames): band b = source rows [band_src[b], band_src[b + 1]) of k_phosphor_*, then the
    // output rows [band_row[b], band_row[b + 1]) of k_warp_lean — every tap of those rows lies above the band's last source row
    std::vector<int> band_src, band_row;
    int group_seg = 128;             // rows per block when a full group is launched (plan_grid)
    int seg_for[4][MAX_GROUP + 1] = {};             // planned rows per block for a (partial) group of g frames, per kernel build (runtime gates / folded / cc / ct)
    // two-stream overlap of k_warp(n) with k_phosphor(n+1): side stream, per-slot events, 2 scratch slots
    bool overlap = false;
    hipStream_t side = nullptr;
    hipEvent_t ev_k1[2] = {nullptr, nullptr}, ev_k2[2] = {nullptr, nullptr};
    bool ev_k2_pending[2] = {false, false};
    int seg_rows = 0;                // rows per k_phosphor block
    unsigned long long* dbg = nullptr;   // -DCRTFX_STAMP builds only: crtfx_debug_buffer hands in a device buffer
    bool force_generic = false;      // CRTFX_OPT_FORCE_GENERIC: always take the LDS-ring kernel (tests)
    int warp_rows = 0;               // CRTFX_OPT_WARP_ROWS = 0 (the launcher's choice) |1|2|4: output rows per k_warp_lean thread
    int point_tiles = 0;             // CRTFX_OPT_POINT_TILES = n: rows (wavefronts) per k_point block, 1..16 (0 = default)
    bool force_runtime_flags = false; // CRTFX_OPT_FORCE_RUNTIME_FLAGS: never take a gate-folded instantiation (tests)
    bool force_cc = false;           // CRTFX_OPT_FORCE_CC: k_phosphor_cc for every radius and pixel format it is built for (tests)
    bool no_cc = false;              // CRTFX_OPT_NO_CC: pre-warp launches stay on k_phosphor_rr instead of k_phosphor_cc (tests, A/B)
    int band_mb = 0;                 // CRTFX_OPT_BAND_MB: > 0 = frames whose pre-warp image exceeds that many MiB run band by band (224 keeps a band under the Infinity Cache; tests band small frames with 1); 0 / -1 = whole frames (the default: no gain measured at 8K)
    bool no_plain_warp = false;      // CRTFX_OPT_NO_PLAIN_WARP: k_warp_lean's branch-free build off (tests, A/B)
    bool no_fused_half = false;      // CRTFX_OPT_NO_FUSED_HALF: the fast-bloom render chain on k_half_group + k_point_lean_seq instead of k_point_fused_seq (tests, A/B)
    bool no_ct = false;              // CRTFX_OPT_NO_CT: ... on k_phosphor_cc instead of k_phosphor_ct (tests, A/B)
    int opt_group = 0, opt_seg_rows = 0;   // CRTFX_OPT_GROUP / CRTFX_OPT_SEG_ROWS: override the launch-shape planner (0 = planner)
    // crtfx_last_plan: which build each kernel class of the most recent apply / process_batch call landed on (every variant of a kernel is
    // bit-identical by test, so only this record shows a planner regression)
    struct Plan { char phosphor[80], warp[96], point[80], half[64], blur[64]; int group, seg_rows, warp_frames; } plan = {};
    bool debug_plan = false;
    std::string err;
    // profiling
    bool prof = false;
    int prof_stride = 1;             // time the launches of every prof_stride-th frame
    unsigned prof_frame = 0;         // frames seen since profiling was switched on
    bool prof_this = false;          // the current frame is a sampled one
    std::vector<hipEvent_t> ev[3];   // pairs (start, stop) per launch, per kernel class (0 phosphor / point, 1 warp / commit, 2 bloom passes in front of class 0)
    std::vector<int> ev_frames[3];   // frames covered by each timed launch
    size_t ev_used[3] = {0, 0, 0};
};

namespace {

// crtfx_create / crtfx_destroy / crtfx_set_params work on the ctx's device and leave the calling thread's current
// device as they found it (a process that drives several GPUs keeps its own hipSetDevice state).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};

int fail(crtfx_ctx* c, int code, const char* fmt, ...) {
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

#define HIP_TRY(c, call)                                                                         \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) return fail((c), CRTFX_E_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

int upload(crtfx_ctx* c, DevBuf& b, const void* host, size_t bytes) {
    if (!host || bytes == 0) return CRTFX_OK;
    if (b.bytes < bytes) {
        if (b.p) HIP_TRY(c, hipFree(b.p));
        b.p = nullptr; b.bytes = 0;
        HIP_TRY(c, hipMalloc(&b.p, bytes));
        b.bytes = bytes;
    }
    HIP_TRY(c, hipMemcpy(b.p, host, bytes, hipMemcpyHostToDevice));
    return CRTFX_OK;
}

void free_buf(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.bytes = 0;
}

uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

void noise_keys(uint64_t seed, uint64_t frame, uint32_t& k0, uint32_t& k1) {
    const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    const uint32_t f0 = (uint32_t)frame, f1 = (uint32_t)(frame >> 32);
    k0 = mix32(s0 ^ mix32(f0 + 0x9E3779B9U) ^ mix32(f1 + 0x85EBCA6BU));
    k1 = mix32(s1 + 0xC2B2AE35U + mix32(k0 ^ f0));
}

KFrame make_kframe(const void* in, const crtfx_frame* f) {
    KFrame k{};
    k.in = static_cast<const uint8_t*>(in);
    if (f) {
        k.scan_row = f->scan_row_dev;
        k.scan_plane = f->scan_plane_dev;
        k.noise_plane = f->noise_plane_dev;
        k.overlay_before = (f->overlay_rgba_dev && !f->overlay_after) ? f->overlay_rgba_dev : nullptr;
        k.flicker = f->flicker_factor;
        noise_keys(f->noise_seed, f->frame_index, k.key0, k.key1);
    } else {
        k.flicker = 1.0;
        noise_keys(0, 0, k.key0, k.key1);
    }
    return k;
}

size_t phosphor_lds_bytes(int R) {
    const int pad = (R + 3) & ~3;
    const int SWP = TW + 2 * pad;
    const size_t floats = (size_t)NB * 3 * SWP + (size_t)(NB + 2 * R) * 3 * TW + (size_t)NB * 3 * TW + 2 * LUT_STRIDE;
    return floats * sizeof(float);
}

// A (start, stop) event pair for the next launch of kernel class k, or (nullptr, nullptr) when profiling
// is off.  The events are attached to the dispatch by CRTFX_LAUNCH, not recorded as separate packets.
struct ProfEv {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfEv(crtfx_ctx* c, int k, int frames = 1) {
        if (!c->prof || !c->prof_this) return;
        auto& v = c->ev[k];
        size_t& u = c->ev_used[k];
        if (u + 2 > v.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            v.push_back(a); v.push_back(b);
        }
        e0 = v[u]; e1 = v[u + 1];
        if (c->ev_frames[k].size() < u / 2 + 1) c->ev_frames[k].resize(u / 2 + 1);
        c->ev_frames[k][u / 2] = frames;
        u += 2;
    }
};

size_t phosphor_rr_lds_bytes(int R, int seg_rows, bool pixelate, int pix = 0, bool runtime = false, bool glut = false);

template <size_t N>
void plan_note(char (&dst)[N], const char* fmt, ...) __attribute__((format(printf, 2, 3)));
template <size_t N>
void plan_note(char (&dst)[N], const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, N, fmt, ap);
    va_end(ap);
}
const char* sf_name(uint32_t sf) { return sf == SF_FAST ? "fast" : sf == SF_FAST_PIX ? "fast+pixelate" : sf == SF_RUNTIME ? "runtime" : "full"; }
const char* pix_name(int pix) { return pix == CRTFX_PIX_F16 ? "half" : "u8"; }
const char* blend_name(int b) { return b == CRTFX_BLEND_RENDER ? "render" : b == CRTFX_BLEND_PREVIEW ? "preview" : "none"; }

// Rows per k_phosphor block.  Every block of the grid should be resident at once (a second,
// partial round of blocks costs a whole extra block lifetime), so the grid is sized to the
// number of block slots: blocks-per-CU (LDS-limited) x 256 CUs.  Measured (4K, R=9): 128 rows x 1020
// blocks 103 us; 184 rows x 720 blocks 118 us; 96 rows x 1380 blocks 119 us.  1080p, R=4: 32 rows x 1020
// blocks 29 us against 34 us at 64 rows — filling the slots beats the extra halo rows; floor 24 rows.
int pick_seg_rows(int H, int W, int R, int pix = 0, int group = 1) {
    const int strips = (W + TW - 1) / TW;
    const size_t lds = phosphor_rr_lds_bytes(rr_build_radius(R) ? R : 9, 128, false, pix);
    int bpc = (int)(163840 / lds);
    if (bpc > 4) bpc = 4;      // 4 waves per SIMD is what the register budget allows
    if (bpc < 1) bpc = 1;
    int segs = (bpc * 256) / (strips * group);
    if (segs < 1) segs = 1;
    int seg = (H + segs - 1) / segs;
    if (seg < 24) seg = 24;
    if (seg > 192) seg = 192;      // frames needing several rounds of blocks (8K: seg 184 -> 1554 fps, one-round seg 720 -> 1145 fps)
    seg = ((seg + NB - 1) / NB) * NB;
    const int hmax = ((H + NB - 1) / NB) * NB;
    return seg > hmax ? hmax : seg;
}

// Launch-shape planner for k_phosphor_rr.  A block of `rows` output rows costs ceil((rows + 2R) / NB) loop
// iterations (+ a fixed prologue); blocks are dealt in dispatch order (x fastest, then row segment, then frame)
// onto bpc x 256 resident slots.  For every candidate (frames per grid g, rows per block seg) the makespan of that
// list schedule is simulated and the pair with the fewest iterations PER FRAME wins.  This reproduces the
// measured landscape (4K, R = 9: g=1/seg=128 -> 19 it/frame = 89 us; g=2/seg=256 -> 17.5 = 82.5 us, the short
// last-segment blocks freeing slots for the overflow; g=2/seg=240 -> two full rounds = 103 us).
struct GridPlan { int g, seg; };
GridPlan plan_grid(int H, int W, int R, int pix, bool folded, bool glut, int gmin, int gmax_allowed, int cc = 0) {      // cc: 0 = k_phosphor_rr, 1 = k_phosphor_cc, 2 = k_phosphor_ct
    const int strips = (W + TW - 1) / TW;
    const int Rk = rr_build_radius(R) ? R : 9;
    const size_t lds = cc == 2 ? (size_t)ct_lds_words(Rk, pix) * 4 : cc ? (size_t)cc_lds_words(Rk, pix) * 4 : phosphor_rr_lds_bytes(Rk, 128, false, pix, !folded, glut);
    int bpc = (int)(163840 / lds);
    const int by_regs = cc == 2 ? ct_min_waves(Rk) : cc ? cc_min_waves(Rk) : rr_min_waves(Rk, folded);      // a block = one wave per SIMD
    bpc = bpc > by_regs ? by_regs : (bpc < 1 ? 1 : bpc);
    const int slots = bpc * 256;
    const int hcap = ((H + NB - 1) / NB) * NB;
    GridPlan best{1, hcap < 128 ? hcap : 128};
    double best_cost = 1e30;
    std::priority_queue<double, std::vector<double>, std::greater<double>> freeat;
    for (int g = gmin; g <= gmax_allowed; ++g) {
        for (int seg = 24; seg <= hcap && seg <= 256; seg += NB) {      // taller blocks lose more than the model sees (8K: 688 rows 474 us, 224 rows 426 us)
            const int segs = (H + seg - 1) / seg;
            const long nblk = (long)strips * segs * g;
            if (nblk > 8L * slots && seg < 184) continue;           // far too many tiny blocks: not worth simulating
            // list scheduling in dispatch order; all blocks of one (frame, segment) row share a duration
            while (!freeat.empty()) freeat.pop();
            for (int k = 0; k < slots; ++k) freeat.push(0.0);
            double makespan = 0.0;
            for (int z = 0; z < g; ++z)
                for (int ys = 0; ys < segs; ++ys) {
                    const int rows = (ys == segs - 1) ? H - ys * seg : seg;
                    const double d = (double)((rows + 2 * R + NB - 1) / NB) + 1.5;
                    for (int x = 0; x < strips; ++x) {
                        const double t = freeat.top() + d;      // the earliest-free slot takes the next block
                        freeat.pop();
                        freeat.push(t);
                        if (t > makespan) makespan = t;
                    }
                }
            double cost = makespan / g;
            if (g > 1) cost *= 1.0 + 0.01 * (g - 1);                // mild preference for small groups (scratch stays cache-sized)
            if (cost < best_cost - 1e-9) { best_cost = cost; best = GridPlan{g, seg}; }
        }
    }
    return best;
}

size_t phosphor_rr_lds_bytes(int R, int seg_rows, bool pixelate, int pix, bool runtime, bool glut) {
    return ((size_t)rr_lds_fixed_floats(R, pix, runtime) + 16 * 5 + (pixelate ? (size_t)seg_rows + 2 * R : 0) + (runtime && glut ? 768 : 0)) * sizeof(float);
}

void launch_generic(crtfx_ctx* c, const KFrame& kf, const KOut& ko, hipStream_t s) {
    const int seg = pick_seg_rows(c->H, c->W, 9, c->pix_fmt, 1);
    const int strips = (c->W + TW - 1) / TW;
    const int segs = (c->H + seg - 1) / seg;
    ProfEv pe(c, 0);
    plan_note(c->plan.phosphor, "k_phosphor<-1>");
    c->plan.group = 1; c->plan.seg_rows = seg;
    CRTFX_LAUNCH((k_phosphor<-1>), dim3(strips, segs), dim3(K1_THREADS), phosphor_lds_bytes(c->kp.R), s, pe.e0, pe.e1, c->kp, kf, ko, seg);
}

bool lean_ok(const crtfx_ctx* c, const KFrame& kf, const KOut& ko) {
    // the lean kernel has no per-pixel plane loads and no in-kernel blend compiled in; half frames have a lean
    // build only for the full-chain gate set
    const int R = c->kp.R;
    const bool folded = (c->kp.flags & ~(uint32_t)CRTFX_F_WARP) == SF_FULL && !c->force_runtime_flags;
    // a per-pixel scanline plane and a coarse grain plane (grain_size > 1) are handled by the runtime-gate build only (uint8 frames)
    // (text overlays before the effects, or after them when the same kernel also commits) are handled by the runtime-gate build only
    const bool needs_runtime = kf.scan_plane || ((c->kp.flags & CRTFX_F_NOISE) && c->kp.grain > 1) || kf.overlay_before || ko.overlay_after;
    (void)folded; (void)needs_runtime;      // both pixel formats have a gate-folded and a runtime-gate build
    return !c->force_generic && !c->split && rr_build_radius(R) == R && R >= 1 && !c->kp.triad_full && !c->kp.vig_full && !kf.noise_plane &&
           ko.blend == CRTFX_BLEND_NONE;
}

constexpr int CC_MIN_RADIUS = 8;
constexpr int CT_MAX_RADIUS = 15;      // k_phosphor_ct keeps four blocks per CU up to radius 15 (4 - 6 VGPRs spilled from 13); at 16 the window no longer fits 128 VGPRs (169 spills): k_phosphor_cc
// The column-owner kernels: k_phosphor_ct for radii 1 .. CT_MAX_RADIUS (round 3: ahead of the register-window kernel at every radius
// measured — 1080p R = 4: 63.6 vs 69.7 us per 5-frame launch, 4K R = 9: 113 vs 126), k_phosphor_cc from CC_MIN_RADIUS up where ct does not serve
bool use_cc(const crtfx_ctx* c, int R) {
    const bool by_radius = c->force_cc || R >= CC_MIN_RADIUS || (!c->no_ct && R >= 1 && R <= CT_MAX_RADIUS);
    return c->pix_fmt == CRTFX_PIX_U8 && by_radius && (size_t)c->H * c->W * 3 * sizeof(float) < ((size_t)1 << 31);
}
// Half frames: k_phosphor_ct<R, 1> for radii 1 .. CT_HALF_MAX_RADIUS (round 5: qword loads of the frame-row window, the centre samples in a register
// window instead of an LDS ring — 34.8 KB of LDS, four blocks per CU); there is no half build of k_phosphor_cc.  Its loads address the frame
// with 32-bit byte offsets.
bool use_ct_half(const crtfx_ctx* c, int R) {
    return c->pix_fmt == CRTFX_PIX_F16 && !c->no_ct && !c->no_cc && R >= 1 && R <= CT_HALF_MAX_RADIUS &&
           (size_t)c->H * c->W * 3 * sizeof(float) < ((size_t)1 << 31) && (size_t)c->H * c->W * 6 < ((size_t)1 << 32);
}

// g frames (1..MAX_GROUP) through the register-window kernel in one launch.
// [y0, y1) / seg_rows: a band of a frame (crtfx_process_batch) — those rows in segments of seg_rows rows, whatever build the launch lands on;
// y1 = 0: the whole frame at the planner's segment height for that build.
void launch_rr_group(crtfx_ctx* c, const KGroup& kg_in, int g, hipStream_t s, int y0 = 0, int y1 = 0, int seg_rows = 0, int prof_frames = -1) {
    KGroup kg = kg_in;
    kg.y0 = y1 > 0 ? y0 : 0;
    kg.y1 = y1 > 0 ? y1 : c->H;
    static rr_launch_fn table[MAX_RADIUS + 1] = {};
    static const bool table_ready = [] {
#define CRTFX_RR_ENTRY(r) table[r] = rr_launch_##r;
        CRTFX_RR_RADII(CRTFX_RR_ENTRY)
#undef CRTFX_RR_ENTRY
        return true;
    }();
    (void)table_ready;
    const int R = c->kp.R;          // the BUILD radius (crtfx_set_params pads the taps of a bucketed radius)
    const uint32_t gates = c->kp.flags & ~(uint32_t)CRTFX_F_WARP;
    // the full-chain gate set, and (uint8 frames) the same with pixelate — the reference CLI's default pixel size is 2 — have gate-folded builds
    const bool pix_fold = gates == (SF_FULL | CRTFX_F_PIXELATE) && c->pix_fmt == CRTFX_PIX_U8;
    bool folded = (gates == SF_FULL || pix_fold) && !c->force_runtime_flags;
    if ((c->kp.flags & CRTFX_F_NOISE) && c->kp.grain > 1) folded = false;
    for (int j = 0; j < g; ++j) if (kg.f[j].scan_plane || kg.f[j].overlay_before || kg.o[j].overlay_after) folded = false;
    // the column-owner kernels take the full-chain launches of uint8 frames that park a pre-warp image (warp and / or persistence behind
    // them): k_phosphor_ct for radii 1 .. CT_MAX_RADIUS = 15, k_phosphor_cc above (use_cc; against the register-window kernel k_phosphor_cc
    // alone paid from radius 8 only — profiles/r02_cc_ab.txt — ct pays at every radius, profiles/r03_ct_ablation.txt).  Their stores address a
    // frame's scratch image with 32-bit byte offsets.
    bool cc = folded && !pix_fold && !c->no_cc && use_cc(c, R);
    for (int j = 0; j < g && cc; ++j) cc = kg.o[j].pre != nullptr;
    bool ct = cc && !c->no_ct && c->pix_fmt == CRTFX_PIX_U8 && R <= CT_MAX_RADIUS;       // the dword-load / composite-table build of the same kernel (uint8 frames)
    // ... whose A phase reads a frame row as aligned dwords: a caller's frame that does not start on a 4-byte boundary (an odd base pointer
    // or frame stride through the C-ABI; torch allocations never are) takes the byte-wise k_phosphor_cc — same bits
    for (int j = 0; j < g && ct; ++j) ct = ((uintptr_t)kg.f[j].in & 3u) == 0;
    // half frames: the qword-load build of k_phosphor_ct (frames on an 8-byte boundary; else the register-window kernel — same bits)
    bool cth = folded && !pix_fold && use_ct_half(c, R);
    for (int j = 0; j < g && cth; ++j) cth = kg.o[j].pre != nullptr && ((uintptr_t)kg.f[j].in & 7u) == 0;
    if (cth) ct = true;
    int& seg_slot = c->seg_for[ct ? 3 : cc ? 2 : (folded ? 1 : 0)][g];
    if (!seg_slot) seg_slot = c->opt_seg_rows ? c->opt_seg_rows : plan_grid(c->H, c->W, R, c->pix_fmt, folded, c->kp.grade_lut != nullptr, g, g, ct ? 2 : cc ? 1 : 0).seg;   // planned once per (kernel build, group size)
    const int seg = seg_rows > 0 ? seg_rows : seg_slot;
    const int strips = (c->W + TW - 1) / TW;
    const int segs = (kg.y1 - kg.y0 + seg - 1) / seg;
    const int variant = cth ? 7 : ct ? 6 : cc ? 4 : (c->pix_fmt == CRTFX_PIX_F16 ? (folded ? 2 : 3) : (folded ? (pix_fold ? 5 : 1) : 0));
    const bool runtime = !folded;
    const size_t lds = ct ? (size_t)ct_lds_words(R, c->pix_fmt) * 4 : cc ? (size_t)cc_lds_words(R, c->pix_fmt) * 4
                          : phosphor_rr_lds_bytes(R, seg, (c->kp.flags & CRTFX_F_PIXELATE) != 0, c->pix_fmt, runtime, c->kp.grade_lut != nullptr);
    ProfEv pe(c, 0, prof_frames >= 0 ? prof_frames : g);
    if (cth) plan_note(c->plan.phosphor, "k_phosphor_ct<%d,half>", R);
    else if (ct) plan_note(c->plan.phosphor, "k_phosphor_ct<%d,u8>", R);
    else if (cc) plan_note(c->plan.phosphor, "k_phosphor_cc<%d,u8>", R);
    else plan_note(c->plan.phosphor, "k_phosphor_rr<%d,%s,%s>", R, folded ? (pix_fold ? "full+pixelate" : "full") : "runtime", pix_name(c->pix_fmt));
    c->plan.group = g; c->plan.seg_rows = seg;
    table[R](c->kp, kg, seg, dim3(strips, segs, g), lds, s, variant, pe.e0, pe.e1);
}

// Radii 1 .. 30 (sigma up to 10; the CLI default 1.2 -> 4, BASELINE config 3 sigma 3 -> 9) run a fused build (launch_rr_group: the
// column-owner kernels k_phosphor_ct / k_phosphor_cc or the register-window kernel k_phosphor_rr); radius 0 (a 1-tap copy), injected
// per-pixel planes and FORCE_GENERIC the generic LDS-ring kernel; larger radii never get here (the split bloom, crtfx_set_params).
void launch_phosphor(crtfx_ctx* c, const KFrame& kf, const KOut& ko, hipStream_t s) {
    if (lean_ok(c, kf, ko)) {
        KGroup kg{};
        kg.f[0] = kf; kg.o[0] = ko;
        launch_rr_group(c, kg, 1, s);
    } else {
        launch_generic(c, kf, ko, s);
    }
}

template <uint32_t SF, int BLEND>
void launch_point_lean2(crtfx_ctx* c, dim3 grid, dim3 block, hipStream_t s, hipEvent_t e0, hipEvent_t e1, const KFrame& kf, const KOut& ko) {
    plan_note(c->plan.point, "k_point_lean<%s,%s,%s>", sf_name(SF), pix_name(c->pix_fmt), blend_name(BLEND));
    if (c->pix_fmt == CRTFX_PIX_F16) { CRTFX_LAUNCH((k_point_lean<SF, CRTFX_PIX_F16, BLEND>), grid, block, 0, s, e0, e1, c->kp, kf, ko); }
    else { CRTFX_LAUNCH((k_point_lean<SF, CRTFX_PIX_U8, BLEND>), grid, block, 0, s, e0, e1, c->kp, kf, ko); }
}
void launch_point_lean(crtfx_ctx* c, bool pixelate, bool render, dim3 grid, dim3 block, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                       const KFrame& kf, const KOut& ko) {
    if (pixelate) { if (render) launch_point_lean2<SF_FAST_PIX, CRTFX_BLEND_RENDER>(c, grid, block, s, e0, e1, kf, ko); else launch_point_lean2<SF_FAST_PIX, CRTFX_BLEND_NONE>(c, grid, block, s, e0, e1, kf, ko); }
    else { if (render) launch_point_lean2<SF_FAST, CRTFX_BLEND_RENDER>(c, grid, block, s, e0, e1, kf, ko); else launch_point_lean2<SF_FAST, CRTFX_BLEND_NONE>(c, grid, block, s, e0, e1, kf, ko); }
}

#ifndef WL_SEQ
#define WL_SEQ 1
#endif
template <bool PROMOTE, int BLEND, int PIX>
void launch_warp_lean2(crtfx_ctx* c, const KWarpGroup& wg, dim3 grid, int ntot, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    // frames without a persistence chain: 4 rows per thread in a 128 x 8 tile; with one (the float32 state of every row in registers
    // as well): 2 rows, 64 x 8 — 1080p, 5 frames per launch: 40.0 against 46.7 us with 4 rows (profiles/r03_ct_ablation.txt, E)
    constexpr int WX = BLEND == CRTFX_BLEND_RENDER ? 1 : 2;
    const int rows = c->warp_rows ? c->warp_rows : (BLEND == CRTFX_BLEND_RENDER ? 2 : 4);      // output rows per thread
    // frames a thread takes one after the other: all of a persistence chain (the state stays in registers).  Without a blend one frame per
    // thread (WL_SEQ = 1): sharing the map coordinates and weights between the frames of a group (WL_SEQ > 1) keeps them live across
    // the frame loop — 164 VGPRs instead of 70, 3 waves per SIMD — and measures 56 - 65 against 52.8 - 53.9 us per 2-frame 4K launch
    constexpr bool SEQ = BLEND == CRTFX_BLEND_RENDER || WL_SEQ > 1;
    const int nseq = BLEND == CRTFX_BLEND_RENDER ? ntot : min(WL_SEQ, ntot);
    grid.x = (grid.x + WX - 1) / WX;
    grid.y = ((int)grid.y + (4 / WX) * rows - 1) / ((4 / WX) * rows);      // the caller's grid.y = the ROWS to cover (a band, or the frame)
    grid.z = (ntot + nseq - 1) / nseq;
    c->plan.warp_frames = ntot;
    plan_note(c->plan.warp, "k_warp_lean<%s,%s,%s,rows=%d,tile=%dx%d,general>", PROMOTE ? "f64" : "f32", blend_name(BLEND), pix_name(PIX), rows, 64 * WX, (4 / WX) * rows);
    // (Padding the 60 tile columns of a 4K frame to 64 — a tile and the tile below it then land on the same XCD, eight
    // dispatches apart, to share their source rows in its L2 — measured SLOWER: 60.0 vs 57.4 us per 2-frame launch.)
    // the headline shape — unblended uint8 frames, none of which keeps a float state, rows of whole dwords — on the branch-free build
    if constexpr (BLEND == CRTFX_BLEND_NONE && !SEQ) {
        // rows of whole dwords: uint8 frames with W % 4 == 0; half frames with W % 2 == 0 on a dword-aligned base
        bool plain = rows == 4 && !c->no_plain_warp && (PIX == CRTFX_PIX_F16 ? (c->W & 1) == 0 && (size_t)c->H * c->W * 6 < ((size_t)1 << 31) : (c->W & 3) == 0);
        for (int j = 0; j < ntot && plain; ++j)
            plain = wg.o[j].out_u8 != nullptr && wg.o[j].state == nullptr && (PIX != CRTFX_PIX_F16 || ((uintptr_t)wg.o[j].out_u8 & 3u) == 0);
        if (plain) {
            plan_note(c->plan.warp, "k_warp_lean<%s,%s,%s,rows=4,tile=%dx%d,plain>", PROMOTE ? "f64" : "f32", blend_name(BLEND), pix_name(PIX), 64 * WX, (4 / WX) * 4);
            CRTFX_LAUNCH((k_warp_lean<PROMOTE, BLEND, PIX, 4, false, WX, SEQ, true>), grid, dim3(256), 0, s, e0, e1, c->kp, wg, nseq, ntot); return;
        }
    }
    // a persistence chain of uint8 frames that all blend into ONE state buffer (the render loop's runs; not the sharded render's per-frame
    // local states), rows of whole dwords: the branch-free build too
    if constexpr (BLEND == CRTFX_BLEND_RENDER && PIX == CRTFX_PIX_U8 && SEQ) {
        bool plain = rows == 2 && (c->W & 3) == 0 && !c->no_plain_warp && (size_t)c->H * c->W * 12 < ((size_t)1 << 31);
        for (int j = 0; j < ntot && plain; ++j) plain = wg.o[j].out_u8 != nullptr && wg.o[j].state != nullptr && wg.o[j].state == wg.o[0].state;
        if (plain) {
            plan_note(c->plan.warp, "k_warp_lean<%s,%s,%s,rows=2,tile=%dx%d,plain>", PROMOTE ? "f64" : "f32", blend_name(BLEND), pix_name(PIX), 64 * WX, (4 / WX) * 2);
            CRTFX_LAUNCH((k_warp_lean<PROMOTE, BLEND, PIX, 2, false, WX, SEQ, true>), grid, dim3(256), 0, s, e0, e1, c->kp, wg, nseq, ntot); return;
        }
    }
    if (rows == 4) { CRTFX_LAUNCH((k_warp_lean<PROMOTE, BLEND, PIX, 4, false, WX, SEQ>), grid, dim3(256), 0, s, e0, e1, c->kp, wg, nseq, ntot); }
    else if (rows == 2) { CRTFX_LAUNCH((k_warp_lean<PROMOTE, BLEND, PIX, 2, false, WX, SEQ>), grid, dim3(256), 0, s, e0, e1, c->kp, wg, nseq, ntot); }
    else { CRTFX_LAUNCH((k_warp_lean<PROMOTE, BLEND, PIX, 1, false, WX, SEQ>), grid, dim3(256), 0, s, e0, e1, c->kp, wg, nseq, ntot); }
}
template <bool PROMOTE, int BLEND>
void launch_warp_lean(crtfx_ctx* c, const KWarpGroup& wg, dim3 grid, int ntot, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if (c->pix_fmt == CRTFX_PIX_F16) launch_warp_lean2<PROMOTE, BLEND, CRTFX_PIX_F16>(c, wg, grid, ntot, s, e0, e1);
    else launch_warp_lean2<PROMOTE, BLEND, CRTFX_PIX_U8>(c, wg, grid, ntot, s, e0, e1);
}

// plain render frames (no glitch band, overlay or float output; every frame of the group with the same blend) take k_warp_lean
bool warp_lean_ok(const crtfx_ctx* c, const KWarpGroup& wg, int g, bool identity) {
    bool lean = !c->force_generic && (size_t)c->H * c->W * 12 < ((size_t)1 << 31);      // k_warp_lean reads the image through a 32-bit buffer resource
    if (identity && !(wg.o[0].blend == CRTFX_BLEND_RENDER && c->pix_fmt == CRTFX_PIX_U8)) lean = false;      // commit-only lean build: render blend, uint8 frames
    for (int j = 0; j < g && lean; ++j) {
        const KOut& o = wg.o[j];
        lean = !o.overlay_after && !o.glitch_offs && !o.out_f32 && o.blend == wg.o[0].blend &&
               (o.blend == CRTFX_BLEND_NONE || o.blend == CRTFX_BLEND_RENDER);
    }
    return lean;
}

// chain: the g frames are consecutive frames of ONE persistence recurrence (frame j + 1 blends with frame j's state); else independent frames.
// [y0, y1): the output rows of a banded launch (lean, unblended, warp on: the caller has checked warp_lean_ok); y1 < 0 = the whole frame.
void launch_warp_group(crtfx_ctx* c, const KWarpGroup& wg_in, int g, bool identity, hipStream_t s, bool chain = false, int y0 = 0, int y1 = -1, int prof_frames = -1) {
    KWarpGroup wg = wg_in;
    wg.y0 = y0;
    const int rows_out = (y1 < 0 ? c->H : y1) - y0;
    dim3 grid((c->W + TW - 1) / TW, rows_out, g);      // y: rows here, tiles below
    const bool lean = warp_lean_ok(c, wg, g, identity);
    if (lean) {
        ProfEv pe(c, 1, prof_frames >= 0 ? prof_frames : g);
        const bool prom = (c->kp.flags & (CRTFX_F_VIGNETTE | CRTFX_F_FLICKER)) != 0;
        const bool rend = wg.o[0].blend == CRTFX_BLEND_RENDER;
        if (rend) grid.z = 1;       // the g frames of a persistence chain: one after the other inside each thread, the state in registers
        if (identity) {             // no warp behind the Gaussian chain: the blend and the commit only
            grid.y = (rows_out + 7) / 8;
            c->plan.warp_frames = g;
            plan_note(c->plan.warp, "k_warp_lean<%s,render,u8,rows=2,commit-only>", prom ? "f64" : "f32");
            if (prom) { CRTFX_LAUNCH((k_warp_lean<true, CRTFX_BLEND_RENDER, CRTFX_PIX_U8, 2, true>), grid, dim3(256), 0, s, pe.e0, pe.e1, c->kp, wg, g, g); }
            else { CRTFX_LAUNCH((k_warp_lean<false, CRTFX_BLEND_RENDER, CRTFX_PIX_U8, 2, true>), grid, dim3(256), 0, s, pe.e0, pe.e1, c->kp, wg, g, g); }
            return;
        }
        if (prom) { if (rend) launch_warp_lean<true, CRTFX_BLEND_RENDER>(c, wg, grid, g, s, pe.e0, pe.e1); else launch_warp_lean<true, CRTFX_BLEND_NONE>(c, wg, grid, g, s, pe.e0, pe.e1); }
        else { if (rend) launch_warp_lean<false, CRTFX_BLEND_RENDER>(c, wg, grid, g, s, pe.e0, pe.e1); else launch_warp_lean<false, CRTFX_BLEND_NONE>(c, wg, grid, g, s, pe.e0, pe.e1); }
        return;
    }
    grid.y = (rows_out + 3) / 4;    // the general kernel: 64 x 4 tiles (whole frames only)
    c->plan.warp_frames = g;
    plan_note(c->plan.warp, "k_warp<%s>", identity ? "commit-only" : "gather");
    if (chain && g > 1) {           // a persistence chain on the general kernel: its frames commit strictly in order (ref:1081-1105)
        for (int j = 0; j < g; ++j) {
            KWarpGroup one{};
            one.pre[0] = wg.pre[j]; one.o[0] = wg.o[j];
            ProfEv pj(c, 1, 1);
            CRTFX_LAUNCH(k_warp, dim3(grid.x, grid.y, 1), dim3(256), 0, s, pj.e0, pj.e1, c->kp, one, identity ? 1 : 0);
        }
        return;
    }
    ProfEv pe(c, 1, g);
    CRTFX_LAUNCH(k_warp, grid, dim3(256), 0, s, pe.e0, pe.e1, c->kp, wg, identity ? 1 : 0);
}

// Split Gaussian bloom: P.ds (plane A) <- blur of the frame's bloom source; plane B is the row-pass intermediate.
void launch_split_blur(crtfx_ctx* c, const KFrame& kf, hipStream_t s) {
    const int H = c->H, W = c->W, R = c->split_R;
    float* A = (float*)c->ds.p;
    float* B = A + (size_t)H * W * 3;
    const unsigned long long* tp = (const unsigned long long*)c->tpad.p;
    const int npairs = sb_tpad_len(R) / 2;
    const dim3 gr((W + SB_SPAN - 1) / SB_SPAN, (H + SB_RW - 1) / SB_RW), br(64 * SB_RW);
    // timed as kernel class 2 (the bloom passes in front of the pointwise kernel); the first launch carries the frame count
    ProfEv p0(c, 2, 1), p1(c, 2, 0);
    plan_note(c->plan.blur, "%sk_sb_rows<%s>+%s", c->split_src_plane ? "k_sb_src+" : "", c->split_src_plane ? "plane" : pix_name(c->pix_fmt),
              ((W & 3) == 0 && !c->split_src_plane) ? "k_sb_cols_lds" : ((W & 3) == 0 ? "k_sb_cols<4>" : "k_sb_cols<1>"));
    if (c->split_src_plane) {      // CRTFX_OPT_SPLIT_SRC_PLANE (A/B): the bloom source as its own plane first
        dim3 gs((W + 63) / 64, (H + 3) / 4);
        ProfEv ps(c, 2, 0);
        if (c->pix_fmt == CRTFX_PIX_F16) { CRTFX_LAUNCH((k_sb_src<CRTFX_PIX_F16>), gs, dim3(256), 0, s, ps.e0, ps.e1, c->kp, kf); }
        else { CRTFX_LAUNCH((k_sb_src<CRTFX_PIX_U8>), gs, dim3(256), 0, s, ps.e0, ps.e1, c->kp, kf); }
        CRTFX_LAUNCH((k_sb_rows<-1>), gr, br, 0, s, p0.e0, p0.e1, c->kp, kf, (const float*)A, B, R, tp, npairs);
    } else if (c->pix_fmt == CRTFX_PIX_F16) { CRTFX_LAUNCH((k_sb_rows<CRTFX_PIX_F16>), gr, br, 0, s, p0.e0, p0.e1, c->kp, kf, (const float*)A, B, R, tp, npairs); }
    else { CRTFX_LAUNCH((k_sb_rows<CRTFX_PIX_U8>), gr, br, 0, s, p0.e0, p0.e1, c->kp, kf, (const float*)A, B, R, tp, npairs); }
    const int rowlen = 3 * W;
    if ((W & 3) == 0 && !c->split_src_plane) {
        const int nbx = (rowlen + 255) / 256, nby = (H + SBC_W * SB_N - 1) / (SBC_W * SB_N);
        CRTFX_LAUNCH(k_sb_cols_lds, dim3(8 * ((nbx * nby + 7) / 8)), dim3(64 * SBC_W), 0, s, p1.e0, p1.e1, (const float*)B, A, H, rowlen, R, tp, npairs, nbx, nby);
    }
    else if ((W & 3) == 0) { CRTFX_LAUNCH((k_sb_cols<4>), dim3((rowlen + 255) / 256, (H + 4 * SB_N - 1) / (4 * SB_N)), dim3(256), 0, s, p1.e0, p1.e1, (const float*)B, A, H, rowlen, R, tp, npairs); }
    else { CRTFX_LAUNCH((k_sb_cols<1>), dim3((rowlen + 63) / 64, (H + 4 * SB_N - 1) / (4 * SB_N)), dim3(256), 0, s, p1.e0, p1.e1, (const float*)B, A, H, rowlen, R, tp, npairs); }
}

// The whole chain for one frame.  ko describes the FINAL outputs.
int run_chain(crtfx_ctx* c, const void* in, const crtfx_frame* f, KOut ko, hipStream_t s) {
    ko.pix = c->pix_fmt;
    c->prof_this = c->prof && (c->prof_frame++ % (unsigned)c->prof_stride == 0);
    if (!c->params_set) return fail(c, CRTFX_E_INVALID, "crtfx_set_params has not been called");
    if (!in) return fail(c, CRTFX_E_INVALID, "frame pointer is NULL");
    const uint32_t fl = c->kp.flags;
    if ((fl & CRTFX_F_SCANLINES) && !(f && (f->scan_row_dev || f->scan_plane_dev)))
        return fail(c, CRTFX_E_INVALID, "scanlines are on but the frame record carries no scan_row_dev / scan_plane_dev");
    const KFrame kf = make_kframe(in, f);
    const bool warp = (fl & CRTFX_F_WARP) != 0;
    const bool gauss = (fl & CRTFX_F_BLOOM) && !(fl & CRTFX_F_BLOOM_FAST) && !c->split;
    const bool ov_after = f && f->overlay_rgba_dev && f->overlay_after;
    const bool glitch = f && f->glitch_offs_dev;
    if (glitch && f->glitch_seg_len < 0) return fail(c, CRTFX_E_INVALID, "glitch_seg_len is negative");
    if (glitch && f->glitch_seg_len > 0 && f->glitch_cols != (c->W + f->glitch_seg_len - 1) / f->glitch_seg_len)
        return fail(c, CRTFX_E_INVALID, "glitch_cols must be ceil(W / glitch_seg_len)");
    if (glitch && f->glitch_seg_len == 0 && !(f->glitch_cols == 1 || f->glitch_cols == c->W))
        return fail(c, CRTFX_E_INVALID, "glitch_cols must be 1 or W");
    if (glitch && (f->glitch_y0 < 0 || f->glitch_y0 >= c->H)) return fail(c, CRTFX_E_INVALID, "glitch_y0 outside the frame");
    // Two-kernel path (pre-warp float32 image + k_warp): warp on; a glitch gather; or a persistence blend
    // behind the Gaussian bloom kernel (kept lean; the state is float32 anyway).  An overlay-after with
    // no warp stays in ONE kernel (the general-purpose build), so that it blends the unrounded image
    // exactly as ref:653-662 does.
    const bool two = warp || glitch || ((gauss || c->split) && ko.blend != CRTFX_BLEND_NONE);
    if (ov_after) ko.overlay_after = f->overlay_rgba_dev;
    if (glitch) { ko.glitch_offs = f->glitch_offs_dev; ko.glitch_y0 = f->glitch_y0; ko.glitch_cols = f->glitch_cols; ko.glitch_seg_len = f->glitch_seg_len; }
    KOut k1 = ko;
    if (two) { k1 = KOut{}; k1.pre = c->pre; }
    k1.dbg = c->dbg;
    if (gauss) {
        launch_phosphor(c, kf, k1, s);
    } else {
        if (c->split) launch_split_blur(c, kf, s);      // Gaussian bloom of any radius: the blurred plane first
        else if (fl & CRTFX_F_BLOOM) {   // fast bloom: half-res source first
            dim3 gh((c->kp.hw + 63) / 64, (c->kp.hh + 3) / 4);
            const uint32_t g0 = fl & ~(uint32_t)CRTFX_F_WARP;
            const bool fold = !c->force_generic && !c->force_runtime_flags && (g0 == SF_FAST || g0 == SF_FAST_PIX) && !kf.overlay_before;
            const bool f16 = c->pix_fmt == CRTFX_PIX_F16;
            ProfEv ph(c, 2);
            plan_note(c->plan.half, "k_half<%s,%s>", fold ? sf_name(g0) : "runtime", fold ? pix_name(c->pix_fmt) : "any");
            if (!fold) { CRTFX_LAUNCH((k_half<SF_RUNTIME, 0>), gh, dim3(256), 0, s, ph.e0, ph.e1, c->kp, kf); }
            else if (g0 == SF_FAST_PIX) {
                if (f16) { CRTFX_LAUNCH((k_half<SF_FAST_PIX, CRTFX_PIX_F16>), gh, dim3(256), 0, s, ph.e0, ph.e1, c->kp, kf); }
                else { CRTFX_LAUNCH((k_half<SF_FAST_PIX, CRTFX_PIX_U8>), gh, dim3(256), 0, s, ph.e0, ph.e1, c->kp, kf); }
            } else {
                if (f16) { CRTFX_LAUNCH((k_half<SF_FAST, CRTFX_PIX_F16>), gh, dim3(256), 0, s, ph.e0, ph.e1, c->kp, kf); }
                else { CRTFX_LAUNCH((k_half<SF_FAST, CRTFX_PIX_U8>), gh, dim3(256), 0, s, ph.e0, ph.e1, c->kp, kf); }
            }
        }
        ProfEv pe(c, 0);
        c->plan.group = 1;                                               // one frame per launch on this path (crtfx_last_plan)
        const int waves = c->point_tiles > 0 ? c->point_tiles : 8;       // rows per block (CRTFX_POINT_TILES): 1080p 4 rows 34.8 us, 8 rows 32.9, 16 rows 37.5
        dim3 grid((c->W + TW - 1) / TW, (c->H + waves - 1) / waves);
        const uint32_t gates = fl & ~(uint32_t)CRTFX_F_WARP;
        const bool lean = !c->force_generic && !c->force_runtime_flags && (gates == SF_FAST || gates == SF_FAST_PIX) && !c->kp.triad_full &&
                          !c->kp.vig_full && !kf.scan_plane && !kf.noise_plane && !kf.overlay_before && c->kp.grain <= 1 &&
                          !k1.overlay_after && !k1.out_f32 && (k1.blend == CRTFX_BLEND_NONE || k1.blend == CRTFX_BLEND_RENDER);
        if (lean) {
            dim3 glean(grid.x, (c->H + waves * CRTFX_POINT_ROWS - 1) / (waves * CRTFX_POINT_ROWS));
            launch_point_lean(c, gates == SF_FAST_PIX, k1.blend == CRTFX_BLEND_RENDER, glean, dim3(64 * waves), s, pe.e0, pe.e1, kf, k1);
        }
        else if (!c->force_generic) {      // any gate set, loads branch-free
            const bool one = !(fl & CRTFX_F_PIXELATE) && !((fl & CRTFX_F_BLOOM) && (fl & CRTFX_F_BLOOM_FAST));      // no load address depends on another load
            plan_note(c->plan.point, "k_point_sel<%s,%s>", pix_name(c->pix_fmt), one ? "one-round" : "two-round");
            if (c->pix_fmt == CRTFX_PIX_F16) {
                if (one) { CRTFX_LAUNCH((k_point_sel<CRTFX_PIX_F16, true>), grid, dim3(64 * waves), 0, s, pe.e0, pe.e1, c->kp, kf, k1); }
                else { CRTFX_LAUNCH((k_point_sel<CRTFX_PIX_F16, false>), grid, dim3(64 * waves), 0, s, pe.e0, pe.e1, c->kp, kf, k1); }
            } else {
                if (one) { CRTFX_LAUNCH((k_point_sel<CRTFX_PIX_U8, true>), grid, dim3(64 * waves), 0, s, pe.e0, pe.e1, c->kp, kf, k1); }
                else { CRTFX_LAUNCH((k_point_sel<CRTFX_PIX_U8, false>), grid, dim3(64 * waves), 0, s, pe.e0, pe.e1, c->kp, kf, k1); }
            }
        }
        else { plan_note(c->plan.point, "k_point<runtime>"); CRTFX_LAUNCH((k_point<SF_RUNTIME>), grid, dim3(64 * waves), 0, s, pe.e0, pe.e1, c->kp, kf, k1); }
    }
    if (two) {
        KWarpGroup wg{};
        wg.pre[0] = c->pre; wg.o[0] = ko;
        launch_warp_group(c, wg, 1, !warp, s);
    }
    HIP_TRY(c, hipGetLastError());
    return CRTFX_OK;
}

int check_blend(crtfx_ctx* c, int blend, double p, const float* state) {
    if (blend != CRTFX_BLEND_NONE && blend != CRTFX_BLEND_RENDER && blend != CRTFX_BLEND_PREVIEW)
        return fail(c, CRTFX_E_INVALID, "unknown blend mode %d", blend);
    if (blend != CRTFX_BLEND_NONE && !state) return fail(c, CRTFX_E_INVALID, "blend needs state_inout_dev");
    if (blend != CRTFX_BLEND_NONE && !(p > 0.0 && p < 1.0)) return fail(c, CRTFX_E_INVALID, "persistence %g outside (0,1)", p);
    return CRTFX_OK;
}

// Launches go to the calling thread's CURRENT device (as with any stream-taking HIP library); refuse loudly when
// that is not the device the ctx (its scratch, tables and the caller's stream) lives on.
int check_device(crtfx_ctx* c) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return fail(c, CRTFX_E_HIP, "hipGetDevice failed");
    if (dev != c->device) return fail(c, CRTFX_E_INVALID, "current device %d is not the ctx's device %d (call hipSetDevice first)", dev, c->device);
    return CRTFX_OK;
}

}  // namespace

extern "C" {

int crtfx_version(void) { return CRTFX_ABI_VERSION; }

const char* crtfx_last_error(const crtfx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int crtfx_create(int device, int height, int width, int pix_fmt, crtfx_ctx** out_ctx) {
    if (!out_ctx) return CRTFX_E_INVALID;
    *out_ctx = nullptr;
    if (height <= 0 || width <= 0 || height > 32767 || width > 32767) return CRTFX_E_INVALID;
    if (pix_fmt != CRTFX_PIX_U8 && pix_fmt != CRTFX_PIX_F16) return CRTFX_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CRTFX_E_HIP;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return CRTFX_E_HIP;
    crtfx_ctx* c = new (std::nothrow) crtfx_ctx();
    if (!c) return CRTFX_E_NOMEM;
    c->device = device; c->H = height; c->W = width; c->pix_fmt = pix_fmt;
    if (hipMalloc((void**)&c->pre, (size_t)height * width * 3 * sizeof(float)) != hipSuccess) { delete c; return CRTFX_E_NOMEM; }
    c->seg_rows = pick_seg_rows(height, width, 9, pix_fmt);
    *out_ctx = c;
    return CRTFX_OK;
}

int crtfx_destroy(crtfx_ctx* c) {
    if (!c) return CRTFX_OK;
    DeviceGuard guard(c->device);
    (void)hipDeviceSynchronize();
    for (DevBuf* b : {&c->consts, &c->glut, &c->triad_row, &c->lut_g, &c->lut_inv, &c->nx2, &c->ny2, &c->xhat, &c->yhat, &c->xmap, &c->ymap, &c->gxo, &c->gxw,
                      &c->gyo, &c->gyw, &c->uxo, &c->uxw, &c->uyo, &c->uyw, &c->dxo, &c->dxw, &c->dyo, &c->dyw, &c->ds, &c->tpad, &c->tcomp}) free_buf(*b);
    if (c->pre) (void)hipFree(c->pre);
    for (auto& v : c->ev) for (hipEvent_t e : v) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) { if (c->ev_k1[i]) (void)hipEventDestroy(c->ev_k1[i]); if (c->ev_k2[i]) (void)hipEventDestroy(c->ev_k2[i]); }
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return CRTFX_OK;
}

int crtfx_set_params(crtfx_ctx* c, const crtfx_params* p) {
    if (!c || !p) return CRTFX_E_INVALID;
    if (p->size != sizeof(crtfx_params)) return fail(c, CRTFX_E_INVALID, "crtfx_params.size %u != %zu", p->size, sizeof(crtfx_params));
    DeviceGuard guard(c->device);
    HIP_TRY(c, guard.err);
    HIP_TRY(c, hipDeviceSynchronize());   // tables may be in use by enqueued work
    const uint32_t fl = p->flags;
    const int H = c->H, W = c->W;
    if ((fl & CRTFX_F_BLOOM) && (fl & CRTFX_F_BLOOM_FAST)) {
        if (!(p->fbu_xofs && p->fbu_xw && p->fbu_yofs && p->fbu_yw)) return fail(c, CRTFX_E_INVALID, "fast bloom needs the fbu_* upsample axes");
        if ((p->fbd_xofs != nullptr) != (p->fbd_yofs != nullptr) || (p->fbd_xofs && !(p->fbd_xw && p->fbd_yw))) return fail(c, CRTFX_E_INVALID, "fbd_* axes must be given together");
    } else if (fl & CRTFX_F_BLOOM) {
        if (p->bloom_radius < 0 || p->bloom_radius > SPLIT_MAX_RADIUS)
            return fail(c, CRTFX_E_INVALID, "bloom radius %d outside [0,%d]", p->bloom_radius, SPLIT_MAX_RADIUS);
        if (!p->bloom_taps) return fail(c, CRTFX_E_INVALID, "bloom on but bloom_taps NULL");
    }
    if ((fl & CRTFX_F_TRIAD) && !p->triad_row && !p->triad_full_dev) return fail(c, CRTFX_E_INVALID, "triad on but no mask");
    if ((fl & CRTFX_F_TRIAD) && (fl & CRTFX_F_TRIAD_LUT) && !(p->lut_g && p->lut_inv)) return fail(c, CRTFX_E_INVALID, "triad LUT path needs lut_g/lut_inv");
    if ((fl & CRTFX_F_VIGNETTE) && !p->vignette_full_dev && !(p->vig_nx2 && p->vig_ny2)) return fail(c, CRTFX_E_INVALID, "vignette on but no tables");
    if ((fl & CRTFX_F_WARP) && !(p->warp_xhat && p->warp_yhat)) return fail(c, CRTFX_E_INVALID, "warp on but no axis tables");
    if ((fl & CRTFX_F_PIXELATE) && !(p->pix_xmap && p->pix_ymap)) return fail(c, CRTFX_E_INVALID, "pixelate on but no index maps");
    if (p->aberration_px < -8 || p->aberration_px > 8) return fail(c, CRTFX_E_INVALID, "aberration_px %d outside [-8,8] (ref:1230)", p->aberration_px);

    int rc;
    const bool fastb = (fl & CRTFX_F_BLOOM) && (fl & CRTFX_F_BLOOM_FAST);
    const int R_asked = ((fl & CRTFX_F_BLOOM) && !fastb) ? p->bloom_radius : 0;
    // injected per-pixel planes / FORCE_GENERIC take the LDS-ring kernel (radii <= GENERIC_MAX_RADIUS, and only below split_from)
    const bool ring_only = c->force_generic || p->triad_full_dev || p->vignette_full_dev;
    // ... and every radius from split_from on (or past what those kernels are built for) runs the split path: blur
    // kernels with no radius limit + the pointwise chain; the fused kernels then see no bloom radius at all
    const bool split = (fl & CRTFX_F_BLOOM) && !fastb &&
                       (R_asked >= c->split_from || R_asked > GENERIC_MAX_RADIUS || (R_asked > RR_MAX_RADIUS && !ring_only));
    const int R = split ? 0 : R_asked;       // <= RR_MAX_RADIUS: a register-window build or the LDS-ring kernel; up to GENERIC_MAX_RADIUS: the latter only
    c->split = split;
    c->split_R = R_asked;
    const bool grain_up = (fl & CRTFX_F_NOISE) && p->grain_size > 1;
    if (grain_up && !(p->grain_xofs && p->grain_xw && p->grain_yofs && p->grain_yw && p->grain_w >= 1 && p->grain_h >= 1))
        return fail(c, CRTFX_E_INVALID, "grain_size > 1 needs the grain_* resize axes");
    if ((rc = upload(c, c->triad_row, p->triad_row, (size_t)W * 3 * sizeof(float)))) return rc;
    if ((rc = upload(c, c->lut_g, p->lut_g, LUT_N * sizeof(float)))) return rc;
    const bool use_glut = p->grade_lut && c->pix_fmt == CRTFX_PIX_U8 && !(fl & CRTFX_F_SATURATION) &&
                          (fl & (CRTFX_F_TEMPERATURE | CRTFX_F_BRIGHTCON | CRTFX_F_GAMMA));
    if ((rc = upload(c, c->glut, use_glut ? p->grade_lut : nullptr, 3 * 256 * sizeof(float)))) return rc;
    if ((rc = upload(c, c->lut_inv, p->lut_inv, LUT_N * sizeof(float)))) return rc;
    if ((rc = upload(c, c->nx2, p->vig_nx2, (size_t)W * sizeof(double)))) return rc;
    if ((rc = upload(c, c->ny2, p->vig_ny2, (size_t)H * sizeof(double)))) return rc;
    if ((rc = upload(c, c->xhat, p->warp_xhat, (size_t)W * sizeof(float)))) return rc;
    if ((rc = upload(c, c->yhat, p->warp_yhat, (size_t)H * sizeof(float)))) return rc;
    if ((rc = upload(c, c->xmap, p->pix_xmap, (size_t)W * sizeof(int32_t)))) return rc;
    if ((rc = upload(c, c->ymap, p->pix_ymap, (size_t)H * sizeof(int32_t)))) return rc;
    const int hw = W / 2 > 1 ? W / 2 : 1, hh = H / 2 > 1 ? H / 2 : 1;
    if (grain_up) {
        if ((rc = upload(c, c->gxo, p->grain_xofs, (size_t)W * 4)) || (rc = upload(c, c->gxw, p->grain_xw, (size_t)W * 4)) ||
            (rc = upload(c, c->gyo, p->grain_yofs, (size_t)H * 4)) || (rc = upload(c, c->gyw, p->grain_yw, (size_t)H * 4))) return rc;
    }
    if (fastb) {
        if ((rc = upload(c, c->uxo, p->fbu_xofs, (size_t)W * 4)) || (rc = upload(c, c->uxw, p->fbu_xw, (size_t)W * 4)) ||
            (rc = upload(c, c->uyo, p->fbu_yofs, (size_t)H * 4)) || (rc = upload(c, c->uyw, p->fbu_yw, (size_t)H * 4))) return rc;
        if (p->fbd_xofs) {
            if ((rc = upload(c, c->dxo, p->fbd_xofs, (size_t)hw * 4)) || (rc = upload(c, c->dxw, p->fbd_xw, (size_t)hw * 4)) ||
                (rc = upload(c, c->dyo, p->fbd_yofs, (size_t)hh * 4)) || (rc = upload(c, c->dyw, p->fbd_yw, (size_t)hh * 4))) return rc;
        }
        const size_t need = (size_t)MAX_GROUP * hw * hh * 3 * sizeof(float);      // one slot per frame of a grouped launch (k_half_group)
        if (c->ds.bytes < need) {
            free_buf(c->ds);
            HIP_TRY(c, hipMalloc(&c->ds.p, need));
            c->ds.bytes = need;
        }
    }
    if (split) {
        const size_t need = (size_t)2 * H * W * 3 * sizeof(float);      // plane A (source, then the blur) and plane B (row pass)
        if (c->ds.bytes < need) {
            free_buf(c->ds);
            HIP_TRY(c, hipMalloc(&c->ds.p, need));
            c->ds.bytes = need;
        }
        const size_t L = (size_t)sb_tpad_len(R_asked);      // [tpad | tpadB]: tpadB = tpad moved down one float, so that both the
        std::vector<float> tp(2 * L, 0.0f);                 // even and the odd tap pairs are aligned 64-bit scalar loads
        std::memcpy(tp.data() + (SB_N - 1), p->bloom_taps, (2 * (size_t)R_asked + 1) * sizeof(float));
        std::memcpy(tp.data() + L + (SB_N - 2), p->bloom_taps, (2 * (size_t)R_asked + 1) * sizeof(float));
        if ((rc = upload(c, c->tpad, tp.data(), tp.size() * sizeof(float)))) return rc;
    }

    KParams k{};
    k.H = H; k.W = W; k.pix = c->pix_fmt; k.flags = fl & 0xFFFFu; k.ab = p->aberration_px; k.R = R; k.grain = p->grain_size;
    k.sat = p->saturation; k.r_gain = p->r_gain; k.b_gain = p->b_gain;
    k.contrast = p->contrast; k.brightness = p->brightness; k.inv_gamma = p->inv_gamma;
    k.thr = p->bloom_thr; k.thr_den = p->bloom_thr_den; k.bloom_strength = p->bloom_strength;
    k.noise_scale = p->noise_scale; k.warp_k = p->warp_k; k.cx = p->warp_cx; k.cy = p->warp_cy;
    k.vig_strength = p->vignette_strength;
    if ((fl & CRTFX_F_BLOOM) && !fastb && !split) {
        std::memset(k.taps, 0, sizeof k.taps);
        std::memcpy(k.taps, p->bloom_taps, (2 * R + 1) * sizeof(float));
    }
    k.triad_row = p->triad_row ? (const float*)c->triad_row.p : nullptr;
    k.triad_full = p->triad_full_dev;
    k.lut_g = (const float*)c->lut_g.p; k.lut_inv = (const float*)c->lut_inv.p;
    k.grade_lut = use_glut ? (const float*)c->glut.p : nullptr;
    // k_phosphor_ct: the two LUT steps of the triad mask composed per mask value, T_m[i] = lut_inv[idx(lut_g[i] * m)] (ref:246-263 with
    // preserve-luma off), for the two most frequent values of the mask row — the float32 product and the truncation are the
    // kernels' own (tail_masks: lut_g[i] * m, lut_index), so a gather from T_m returns the very float the two gathers would
    k.triad_comp = nullptr; k.comp_m0 = k.comp_m1 = 0u;
    if ((fl & CRTFX_F_TRIAD) && (fl & CRTFX_F_TRIAD_LUT) && !(fl & CRTFX_F_TRIAD_LUMA) && p->triad_row && !p->triad_full_dev) {
        const float* row = p->triad_row;
        uint32_t vals[2] = {0u, 0u};
        size_t cnt[2] = {0, 0};
        {   // the two most frequent bit patterns: a softened period-3 mask has two or three interior values plus a few border ones
            std::vector<std::pair<uint32_t, size_t>> hist;
            bool many = false;                          // a mask with 64 or more distinct values is not a period-3 row: no composite form, no tables built
            for (size_t i = 0; i < (size_t)W * 3 && !many; ++i) {
                uint32_t b; std::memcpy(&b, row + i, 4);
                size_t j = 0;
                for (; j < hist.size(); ++j) if (hist[j].first == b) { ++hist[j].second; break; }
                if (j == hist.size()) { if (hist.size() >= 64) many = true; else hist.emplace_back(b, 1); }
            }
            if (many) hist.clear();
            for (auto& h : hist) {
                if (h.second > cnt[0]) { vals[1] = vals[0]; cnt[1] = cnt[0]; vals[0] = h.first; cnt[0] = h.second; }
                else if (h.second > cnt[1]) { vals[1] = h.first; cnt[1] = h.second; }
            }
            if (cnt[1] == 0) vals[1] = vals[0];
        }
        if (cnt[0] > 0) {
            const float* lg = static_cast<const float*>(p->lut_g);
            const float* li = static_cast<const float*>(p->lut_inv);
            std::vector<float> tab(2 * LUT_N);
            for (int t = 0; t < 2; ++t) {
                float m; std::memcpy(&m, &vals[t], 4);
                for (int i = 0; i < LUT_N; ++i) {
                    volatile float q = lg[i] * m;                       // one float32 rounding, as the kernels' v_mul_f32
                    float v = q;
                    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);       // lut_index: (int)(clip01(v) * 1024.0f); a NaN product clips to 0 as fminf(fmaxf()) does
                    if (!(q == q)) v = 0.0f;
                    tab[(size_t)t * LUT_N + i] = li[(int)(v * 1024.0f)];
                }
            }
            if ((rc = upload(c, c->tcomp, tab.data(), tab.size() * sizeof(float)))) return rc;
            k.triad_comp = (const float*)c->tcomp.p; k.comp_m0 = vals[0]; k.comp_m1 = vals[1];
        }
    }
    {
        float cst[32 + 256] = {1.0f, 1.0f, 1.0f, 1.0f};      // then zeros: the address a disabled stage loads from in k_point_sel
        for (int i = 0; i < 256; ++i) cst[32 + i] = (float)i / 255.0f;      // u / 255 (the IEEE quotient, = norm_u8)
        if ((rc = upload(c, c->consts, cst, sizeof(cst)))) return rc;
        k.consts = (const float*)c->consts.p;
    }

    k.vig_nx2 = (const double*)c->nx2.p; k.vig_ny2 = (const double*)c->ny2.p;
    k.vig_full = p->vignette_full_dev;
    k.xhat = (const float*)c->xhat.p; k.yhat = (const float*)c->yhat.p;
    k.xmap = (const int*)c->xmap.p; k.ymap = (const int*)c->ymap.p;
    k.gx_ofs = (const int*)c->gxo.p; k.gx_a = (const float*)c->gxw.p; k.gy_ofs = (const int*)c->gyo.p; k.gy_a = (const float*)c->gyw.p;
    k.gw = p->grain_w; k.gh = p->grain_h;
    k.ux_ofs = (const int*)c->uxo.p; k.ux_a = (const float*)c->uxw.p; k.uy_ofs = (const int*)c->uyo.p; k.uy_a = (const float*)c->uyw.p;
    const bool down_tab = fastb && p->fbd_xofs;
    k.dx_ofs = down_tab ? (const int*)c->dxo.p : nullptr; k.dx_a = down_tab ? (const float*)c->dxw.p : nullptr;
    k.dy_ofs = down_tab ? (const int*)c->dyo.p : nullptr; k.dy_a = down_tab ? (const float*)c->dyw.p : nullptr;
    k.hw = hw; k.hh = hh; k.ds = (float*)c->ds.p;
    if (fastb && !down_tab && !((W % 2 == 0) && (H % 2 == 0)))
        return fail(c, CRTFX_E_INVALID, "fast bloom on an odd frame size needs the fbd_* downsample axes");
    if ((fl & CRTFX_F_VIGNETTE) && !p->vignette_full_dev && p->vignette_strength >= 0.0 && p->vignette_strength <= 1.0) k.flags |= KF_VIG_UNIT;
    c->kp = k;
    c->params_set = true;
    c->seg_rows = pick_seg_rows(H, W, R, c->pix_fmt);
    {   // launch shape of the grouped path: frames per grid and rows per block from the planner
        // keep the float32 scratch of a whole group inside the 256 MiB Infinity Cache (k_warp reads it right back:
        // 4K with 4 frames per grid = 398 MB and k_warp goes from 38 to 44 us per frame)
        int gcap = (int)(((size_t)224 << 20) / ((size_t)H * W * 3 * sizeof(float)));
        gcap = gcap < 1 ? 1 : (gcap > MAX_GROUP ? MAX_GROUP : gcap);
        const uint32_t gates_plan = k.flags & ~(uint32_t)CRTFX_F_WARP;
        const bool pix_fold_plan = gates_plan == (SF_FULL | CRTFX_F_PIXELATE) && c->pix_fmt == CRTFX_PIX_U8;
        const bool folded_plan = (gates_plan == SF_FULL || pix_fold_plan) && !c->force_runtime_flags && !((k.flags & CRTFX_F_NOISE) && k.grain > 1);
        // the render loop's full-chain launches with warp on park a pre-warp image -> k_phosphor_cc (launch_rr_group)
        const bool cc_plan = folded_plan && !pix_fold_plan && !c->no_cc && (k.flags & CRTFX_F_WARP) && use_cc(c, R);
        const bool cth_plan = folded_plan && !pix_fold_plan && (k.flags & CRTFX_F_WARP) && use_ct_half(c, R);
        const int cc_build = cth_plan ? 2 : cc_plan ? ((!c->no_ct && c->pix_fmt == CRTFX_PIX_U8 && R <= CT_MAX_RADIUS) ? 2 : 1) : 0;
        GridPlan gp = plan_grid(H, W, R, c->pix_fmt, folded_plan, k.grade_lut != nullptr, 1, gcap, cc_build);
        if (c->opt_group >= 1 && c->opt_group <= MAX_GROUP) gp = plan_grid(H, W, R, c->pix_fmt, folded_plan, k.grade_lut != nullptr, c->opt_group, c->opt_group, cc_build);      // the planner's rows per block for the group size asked for
        if (c->opt_seg_rows >= NB) gp.seg = ((c->opt_seg_rows + NB - 1) / NB) * NB;
        const int need = c->overlap ? 2 * gp.g : gp.g;
        if (need > c->pre_frames) {
            (void)hipFree(c->pre);
            c->pre = nullptr;
            HIP_TRY(c, hipMalloc((void**)&c->pre, (size_t)need * H * W * 3 * sizeof(float)));
            c->pre_frames = need;
        }
        // Bands.  One frame's float32 pre-warp image larger than the Infinity Cache (8K: 398 MB) used to be written whole and read back
        // from HBM; split into bands of row segments of <= 224 MB each, k_phosphor(band b) is followed at once by the k_warp_lean rows
        // that only need source rows above the band's end.  The barrel map is monotone in y and, for a fixed row, extreme at the frame's
        // centre column or its edges, so the last source row an output row touches is found from three columns of the host's own axis
        // tables with the kernels' float32 arithmetic (+ 1 for the lower tap, + 1 row of margin).
        c->band_src.clear(); c->band_row.clear();
        const size_t frame_scratch = (size_t)H * W * 3 * sizeof(float);
        const size_t band_bytes = (size_t)(c->band_mb > 0 ? c->band_mb : 224) << 20;
        // OFF unless asked for (BAND_MB > 0): measured on 8K float16 frames (profiles/r04_8k_bands.txt), two bands of 199 MB leave k_phosphor
        // where it was (2 x 139.5 against 278 us per frame) and take k_warp_lean from 141 to 137 us — the 8K warp is bound by its 6-byte
        // output pixels and its taps' instruction count, not by Infinity-Cache misses; four bands are 5 % slower
        if ((k.flags & CRTFX_F_WARP) && gp.g == 1 && frame_scratch > band_bytes && c->band_mb > 0 && p->warp_xhat && p->warp_yhat &&
            (fl & CRTFX_F_BLOOM) && !fastb && !split && frame_scratch < ((size_t)1 << 31)) {
            const int nb = (int)((frame_scratch + band_bytes - 1) / band_bytes);
            const int hband = (((H + nb - 1) / nb) + NB - 1) / NB * NB;
            // the launch shape of ONE band: the planner's for a frame of the band's height (its short last segment frees block slots for
            // the overflow of a grid slightly larger than one resident round, as in the 4K two-frame group)
            if (!(c->opt_seg_rows >= NB)) gp.seg = plan_grid(hband, W, R, c->pix_fmt, folded_plan, k.grade_lut != nullptr, 1, 1, cc_build).seg;
            std::vector<int> need(H);
            const float* xh = static_cast<const float*>(p->warp_xhat);
            const float* yh = static_cast<const float*>(p->warp_yhat);
            const int xs[4] = {0, W - 1, (W - 1) / 2, W / 2};
            for (int y = 0; y < H; ++y) {
                int m = -(1 << 30);
                for (int xi = 0; xi < 4; ++xi) {
                    const float xv = xh[xs[xi]], yv = yh[y];
                    const float r2 = xv * xv + yv * yv;
                    const float factor = 1.0f + k.warp_k * r2;
                    const float my = (yv * factor) * k.cy + k.cy;
                    const int sy = (int)rintf(my * 32.0f);
                    const int iy = sy >> 5;
                    if (iy > m) m = iy;
                }
                need[y] = m + 2;
            }
            c->band_src.push_back(0); c->band_row.push_back(0);
            for (int b = 1; b <= nb; ++b) {
                const int src_end = b * hband < H ? b * hband : H;      // source rows [0, src_end) exist once band b - 1 has been written
                int row = H;
                if (src_end < H) {
                    row = c->band_row.back();
                    while (row < H && need[row] < src_end) ++row;
                    row &= ~15;                                  // whole tiles of the tallest k_warp_lean shape (4 / WX waves x ROWS rows: 8 without a blend, 16 at most)
                    if (row < c->band_row.back()) row = c->band_row.back();
                }
                c->band_src.push_back(src_end); c->band_row.push_back(row);
                if (src_end >= H) break;
            }
            if (c->band_src.size() < 3) { c->band_src.clear(); c->band_row.clear(); }      // one band = the whole frame
        }
        c->group_max = gp.g;
        c->group_seg = gp.seg;
        if (c->debug_plan) fprintf(stderr, "[crtfx] %dx%d R=%d: %d frame(s) per grid, %d rows per block%s\n", W, H, R, gp.g, gp.seg, cc_build == 2 ? " (k_phosphor_ct)" : cc_build ? " (k_phosphor_cc)" : "");
        for (int b = 0; b < 4; ++b) for (int g = 1; g <= MAX_GROUP; ++g) c->seg_for[b][g] = 0;
        c->seg_for[cc_build == 2 ? 3 : cc_build ? 2 : (folded_plan ? 1 : 0)][gp.g] = gp.seg;
    }

    if ((fl & CRTFX_F_BLOOM) && R <= GENERIC_MAX_RADIUS) {
        const size_t lds = phosphor_lds_bytes(R);
        if (lds > 160 * 1024) return fail(c, CRTFX_E_UNSUPPORTED, "bloom radius %d needs %zu B of LDS", R, lds);
        // opt in to > 64 KiB of dynamic LDS
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_phosphor<-1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    return CRTFX_OK;
}

int crtfx_apply_static(crtfx_ctx* c, const void* frame_dev, float* out_float_dev, const crtfx_frame* frame, void* stream) {
    if (!c) return CRTFX_E_INVALID;
    if (int rcdev = check_device(c)) return rcdev;
    if (!out_float_dev) return fail(c, CRTFX_E_INVALID, "out_float_dev is NULL");
    KOut ko{};
    ko.out_f32 = out_float_dev;
    ko.blend = CRTFX_BLEND_NONE;
    c->plan = {};
    return run_chain(c, frame_dev, frame, ko, (hipStream_t)stream);
}

int crtfx_apply(crtfx_ctx* c, const void* frame_dev, void* out_pix_dev, float* state_inout_dev, float* out_float_dev,
                int blend, double persistence, const crtfx_frame* frame, void* stream) {
    if (!c) return CRTFX_E_INVALID;
    if (int rcdev = check_device(c)) return rcdev;
    int rc = check_blend(c, blend, persistence, state_inout_dev);
    if (rc) return rc;
    if (!out_pix_dev && !state_inout_dev && !out_float_dev) return fail(c, CRTFX_E_INVALID, "no output requested");
    KOut ko{};
    ko.out_u8 = static_cast<uint8_t*>(out_pix_dev);
    ko.state = state_inout_dev;
    ko.out_f32 = out_float_dev;
    ko.blend = blend;
    ko.p = persistence; ko.q = 1.0 - persistence;
    c->plan = {};
    return run_chain(c, frame_dev, frame, ko, (hipStream_t)stream);
}

int crtfx_blend_quantise(crtfx_ctx* c, const float* static_dev, float* state_inout_dev, void* out_pix_dev, int blend,
                         double persistence, void* stream) {
    if (!c) return CRTFX_E_INVALID;
    if (int rcdev = check_device(c)) return rcdev;
    if (!static_dev) return fail(c, CRTFX_E_INVALID, "static_dev is NULL");
    int rc = check_blend(c, blend, persistence, state_inout_dev);
    if (rc) return rc;
    KOut ko{};
    ko.out_u8 = static_cast<uint8_t*>(out_pix_dev);
    ko.state = state_inout_dev;
    ko.blend = blend;
    ko.p = persistence; ko.q = 1.0 - persistence;
    ko.pix = c->pix_fmt;
    ProfEv pe(c, 1);
    dim3 grid((c->W + TW - 1) / TW, (c->H + 3) / 4);
    CRTFX_LAUNCH(k_commit, grid, dim3(256), 0, (hipStream_t)stream, pe.e0, pe.e1, c->H, c->W, static_dev, (const float*)nullptr, 0.0, ko, 0);
    HIP_TRY(c, hipGetLastError());
    return CRTFX_OK;
}

int crtfx_halo_correct_quantise(crtfx_ctx* c, const float* local_dev, const float* carry_in_dev, double coeff,
                                float* state_out_dev, void* out_pix_dev, void* stream) {
    if (!c) return CRTFX_E_INVALID;
    if (int rcdev = check_device(c)) return rcdev;
    if (!local_dev || !carry_in_dev) return fail(c, CRTFX_E_INVALID, "local_dev / carry_in_dev is NULL");
    KOut ko{};
    ko.out_u8 = static_cast<uint8_t*>(out_pix_dev);
    ko.state = state_out_dev;
    ko.blend = CRTFX_BLEND_NONE;
    ko.pix = c->pix_fmt;
    ProfEv pe(c, 1);
    dim3 grid((c->W + TW - 1) / TW, (c->H + 3) / 4);
    CRTFX_LAUNCH(k_commit, grid, dim3(256), 0, (hipStream_t)stream, pe.e0, pe.e1, c->H, c->W, local_dev, carry_in_dev, coeff, ko, 1);
    HIP_TRY(c, hipGetLastError());
    return CRTFX_OK;
}

int crtfx_halo_correct_batch(crtfx_ctx* c, const float* local_base_dev, const float* carry_in_dev, double persistence, int first_power,
                             int n, void* out_base, size_t out_stride_bytes, void* stream) {
    if (!c) return CRTFX_E_INVALID;
    if (int rcdev = check_device(c)) return rcdev;
    if (!local_base_dev || !carry_in_dev || !out_base || n < 0 || first_power < 1) return fail(c, CRTFX_E_INVALID, "bad halo batch arguments");
    if (!(persistence > 0.0 && persistence < 1.0)) return fail(c, CRTFX_E_INVALID, "persistence %g outside (0,1)", persistence);
    const size_t frame_elems = (size_t)c->H * c->W * 3;
    dim3 grid((c->W + TW - 1) / TW, (c->H + 3) / 4);
    for (int done = 0; done < n; done += HALO_MAX_FRAMES) {
        const int m = n - done < HALO_MAX_FRAMES ? n - done : HALO_MAX_FRAMES;
        HaloCoeffs K{};
        for (int j = 0; j < m; ++j) K.c[j] = (float)pow(persistence, (double)(first_power + done + j));      // float32(p ** k), as the single-frame entry
        ProfEv pe(c, 1, m);
        CRTFX_LAUNCH(k_halo_batch, grid, dim3(256), 0, (hipStream_t)stream, pe.e0, pe.e1, c->H, c->W,
                     local_base_dev + (size_t)done * frame_elems, frame_elems, carry_in_dev, K, m,
                     static_cast<uint8_t*>(out_base) + (size_t)done * out_stride_bytes, out_stride_bytes, c->pix_fmt);
    }
    HIP_TRY(c, hipGetLastError());
    return CRTFX_OK;
}

int crtfx_process_batch(crtfx_ctx* c, const void* frames_base, size_t frame_stride_bytes, void* out_base,
                        size_t out_stride_bytes, int n, const crtfx_frame* frames, float* state_inout_dev,
                        double persistence, int first_has_state, float* local_states_base, void* stream) {
    if (!c) return CRTFX_E_INVALID;
    if (int rcdev = check_device(c)) return rcdev;
    if (n < 0 || !frames_base || (!out_base && !local_states_base)) return fail(c, CRTFX_E_INVALID, "bad batch arguments");
    if (persistence > 0.0 && !state_inout_dev) return fail(c, CRTFX_E_INVALID, "persistence > 0 needs state_inout_dev");
    if (local_states_base && !(persistence > 0.0)) return fail(c, CRTFX_E_INVALID, "local_states_base needs persistence > 0");
    if (!c->params_set) return fail(c, CRTFX_E_INVALID, "crtfx_set_params has not been called");
    hipStream_t s = (hipStream_t)stream;
    const size_t frame_elems = (size_t)c->H * c->W * 3;
    const uint32_t fl = c->kp.flags;
    const bool warp = (fl & CRTFX_F_WARP) != 0;
    const bool gauss = (fl & CRTFX_F_BLOOM) && !(fl & CRTFX_F_BLOOM_FAST);
    const bool blend_on = persistence > 0.0;
    c->plan = {};

    auto final_out = [&](int i) {
        KOut ko{};
        ko.pix = c->pix_fmt;
        ko.out_u8 = out_base ? static_cast<uint8_t*>(out_base) + (size_t)i * out_stride_bytes : nullptr;
        ko.p = persistence; ko.q = 1.0 - persistence;
        if (frames && frames[i].overlay_rgba_dev && frames[i].overlay_after) ko.overlay_after = frames[i].overlay_rgba_dev;
        if (blend_on) {
            ko.state = state_inout_dev; ko.blend = (i > 0 || first_has_state) ? CRTFX_BLEND_RENDER : CRTFX_BLEND_NONE;
            if (local_states_base) {     // per-frame states wanted: frame i reads state i-1 and writes state i in place of a copy per frame
                ko.state = local_states_base + (size_t)i * frame_elems;
                ko.state_in = i ? local_states_base + (size_t)(i - 1) * frame_elems : state_inout_dev;
            }
        }
        return ko;
    };
    auto frame_in = [&](int i) { return static_cast<const uint8_t*>(frames_base) + (size_t)i * frame_stride_bytes; };
    int i = 0, group_no = 0;
    // crtfx_last_plan describes the call's FULL-SIZE launch group (the one with the most frames; the later of equals): a batch that is not a
    // multiple of the group size ends in a shorter group with its own launch shape, which is one launch in hundreds (round 5's bench lines
    // read `group: 2, seg_rows: 64` for a batch of 1638 groups of 5 x 168 rows and one of 2)
    crtfx_ctx::Plan best = {};
    auto group_done = [&]() { if (c->plan.group >= best.group) best = c->plan; };
    while (i < n) {
        c->plan = {};                              // the record of THIS group's launches only
        // ---- grouped path: Gaussian-bloom chain on the register-window kernel, up to group_max frames per launch ----
        int g = 0;
        KGroup kg{};
        if (gauss) {
            const int gmax = n - i < c->group_max ? n - i : c->group_max;
            const bool two = warp || blend_on;
            const bool ovl = c->overlap && two;
            const int slot = ovl ? (group_no & 1) : 0;
            for (; g < gmax; ++g) {
                const crtfx_frame* f = frames ? &frames[i + g] : nullptr;
                if ((fl & CRTFX_F_SCANLINES) && !(f && (f->scan_row_dev || f->scan_plane_dev)))
                    return fail(c, CRTFX_E_INVALID, "scanlines are on but the frame record carries no scan_row_dev / scan_plane_dev");
                if (f && f->glitch_offs_dev) break;
                KFrame kf = make_kframe(frame_in(i + g), f);
                KOut k1{};
                if (two) {
                    k1.pre = c->pre + ((size_t)slot * c->group_max + g) * frame_elems; k1.pix = c->pix_fmt;
                } else k1 = final_out(i + g);
                k1.dbg = c->dbg;
                if (!lean_ok(c, kf, k1)) break;
                kg.f[g] = kf; kg.o[g] = k1;
            }
            if (g > 0) {
                c->prof_this = c->prof && (c->prof_frame++ % (unsigned)c->prof_stride == 0);
                // two-stream overlap: this group's warp/commit kernels run on the side stream, concurrently with the
                // NEXT group's k_phosphor on the caller's stream (k_warp is gather-latency bound, k_phosphor VALU bound,
                // and a 42-VGPR warp wave fits beside four 115-VGPR phosphor waves per SIMD).  Two scratch slots.
                hipStream_t sw = ovl ? c->side : s;
                if (ovl && c->ev_k2_pending[slot]) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_k2[slot], 0));   // slot free again
                bool banded = false;
                if (two && !blend_on && warp && g == 1 && !ovl && c->band_src.size() >= 3) {
                    // a frame larger than the Infinity Cache, band by band: k_phosphor over the band's row segments, then the output rows
                    // whose taps it completes (planned in crtfx_set_params); the launches of a frame are timed as ONE frame
                    KWarpGroup wg{};
                    wg.pre[0] = c->pre + (size_t)slot * c->group_max * frame_elems; wg.o[0] = final_out(i);
                    if (warp_lean_ok(c, wg, 1, false)) {
                        banded = true;
                        for (size_t b = 0; b + 1 < c->band_src.size(); ++b) {
                            launch_rr_group(c, kg, 1, s, c->band_src[b], c->band_src[b + 1], c->group_seg, b == 0 ? 1 : 0);
                            if (c->band_row[b + 1] > c->band_row[b])
                                launch_warp_group(c, wg, 1, false, s, false, c->band_row[b], c->band_row[b + 1], b == 0 ? 1 : 0);
                        }
                    }
                }
                if (!banded) launch_rr_group(c, kg, g, s);
                if (ovl) { HIP_TRY(c, hipEventRecord(c->ev_k1[slot], s)); HIP_TRY(c, hipStreamWaitEvent(sw, c->ev_k1[slot], 0)); }
                if (two && !banded) {
                    const float* pre0 = c->pre + (size_t)slot * c->group_max * frame_elems;
                    if (!blend_on) {
                        KWarpGroup wg{};
                        for (int j = 0; j < g; ++j) { wg.pre[j] = pre0 + (size_t)j * frame_elems; wg.o[j] = final_out(i + j); }
                        launch_warp_group(c, wg, g, !warp, sw);
                    } else {                 // the persistence IIR commits frames strictly in order (ref:1081-1105): runs of frames that
                        int j = 0;           // blend with their predecessor go to one launch, each thread carrying its pixels' state
                        while (j < g) {
                            int nrun = 1;
                            if (final_out(i + j).blend == CRTFX_BLEND_RENDER)
                                while (j + nrun < g && final_out(i + j + nrun).blend == CRTFX_BLEND_RENDER) ++nrun;
                            KWarpGroup wg{};
                            for (int k = 0; k < nrun; ++k) { wg.pre[k] = pre0 + (size_t)(j + k) * frame_elems; wg.o[k] = final_out(i + j + k); }
                            launch_warp_group(c, wg, nrun, !warp, sw, nrun > 1);
                            j += nrun;
                        }
                    }
                }
                if (ovl) { HIP_TRY(c, hipEventRecord(c->ev_k2[slot], sw)); c->ev_k2_pending[slot] = true; }
                HIP_TRY(c, hipGetLastError());
                group_done();
                i += g;
                ++group_no;
                continue;
            }
        }
        // ---- grouped path 2: runs of frames of the pointwise render chain (fast / no bloom, no warp: the reference CLI's defaults
        // and everything one knob away from them) — up to MAX_GROUP consecutive frames that all blend with their predecessor
        // (persistence) or that do not blend at all, in two launches: their half-res bloom sources side by side (k_half_group),
        // then k_point_lean_seq / k_point_sel_seq, each thread taking the frames one after the other with its pixels' state in registers
        {
            const uint32_t gates = fl & ~(uint32_t)CRTFX_F_WARP;
            const bool fastb = (fl & CRTFX_F_BLOOM) && (fl & CRTFX_F_BLOOM_FAST);
            const bool seq_ok = !gauss && !c->split && !c->force_generic &&
                                (!warp || (size_t)c->H * c->W * 12 < ((size_t)1 << 31));      // with a warp behind it: pre-warp images parked, then k_warp_lean
            // "lean" = no per-pixel plane anywhere in the chain (full-size triad / vignette masks, coarse grain, and — checked per frame below — a 2-D
            // scanline plane, an injected grain plane, text overlays).  The gate sets of the reference CLI's defaults have builds with the gate word
            // folded at compile time; every other plane-free gate set (a colour grade, a bloom threshold, stages switched off, flicker, preserve-luma)
            // runs the same kernels with the gate word at run time (SF_LEAN_RT)
            const int waves = c->point_tiles > 0 ? c->point_tiles : 8;
            // the fast-bloom source formed inside the pointwise kernel (k_point_fused_seq): exact 2x decimation (W, H even: no dx / dy tap tables), a block
            // of 4 .. 8 wavefronts (its first 34 x (waves + 2) threads form the half-resolution tiles of the run's frames)
            const bool can_fuse = fastb && !c->no_fused_half && !c->kp.dx_ofs && waves >= 4 && waves <= 8;
            // coarse grain (--grain-size > 1) has one lean build: the defaults' gate set on the fused kernel, uint8 frames (KF_COARSE); anything else
            // with coarse grain stays on the general kernels
            const bool coarse = (fl & CRTFX_F_NOISE) && c->kp.grain > 1;
            const bool coarse_knob = coarse && can_fuse && !c->force_runtime_flags && c->pix_fmt == CRTFX_PIX_U8 &&
                                     ((gates & ~(uint32_t)CRTFX_F_BLOOM_THR) == SF_FAST || (gates & ~(uint32_t)CRTFX_F_BLOOM_THR) == SF_FAST_PIX);
            const bool lean_gates = !c->force_runtime_flags && !c->kp.triad_full && !c->kp.vig_full && (!coarse || coarse_knob);
            // a 2-D scanline plane per frame (--scanline-angle / --scanline-thickness) has one lean build too: the defaults' gate set, uint8 frames, every
            // frame of the group with a plane (KF_SCANPLANE)
            const bool scan_ok = can_fuse && !coarse && c->pix_fmt == CRTFX_PIX_U8 &&
                                 ((gates & ~(uint32_t)CRTFX_F_BLOOM_THR) == SF_FAST || (gates & ~(uint32_t)CRTFX_F_BLOOM_THR) == SF_FAST_PIX);
            int nplane = 0;
            // (a bloom threshold does not choose the build: the folded _seq kernels keep that one bit at run time — it acts on the bloom source only;
            // k_half_group's folded builds do not, so a thresholded source that is a plane comes from the general k_half_group)
            const bool thr = (gates & CRTFX_F_BLOOM_THR) != 0;
            const uint32_t gates_nt = gates & ~(uint32_t)CRTFX_F_BLOOM_THR;
            const bool folded_gates = (gates_nt == SF_FAST || gates_nt == SF_FAST_PIX) && !coarse;
            // ... and the defaults with ONE knob turned that the grade table cannot express — a saturation change, preserve-luma, flicker, or one of
            // grain / vignette / triad / scanlines switched off: folded builds of k_point_fused_seq for uint8 frames (CRTFX_KNOB_SETS below)
            const char* knob = coarse_knob ? "+coarse" : nullptr;
            if (!folded_gates && !coarse && c->pix_fmt == CRTFX_PIX_U8) {
#define CRTFX_KNOB_SETS(X)                                                                                                                                  \
    X(| CRTFX_F_SATURATION, "+sat") X(| CRTFX_F_TRIAD_LUMA, "+luma") X(| CRTFX_F_FLICKER, "+flicker") X(& ~(uint32_t)CRTFX_F_NOISE, "-grain")               \
    X(& ~(uint32_t)(CRTFX_F_VIGNETTE | KF_VIG_UNIT), "-vignette") X(& ~(uint32_t)(CRTFX_F_TRIAD | CRTFX_F_TRIAD_LUT), "-triad") X(& ~(uint32_t)CRTFX_F_SCANLINES, "-scanlines")
#define CRTFX_KNOB_NAME(OP, NAME) if (gates_nt == (uint32_t)(SF_FAST OP) || gates_nt == (uint32_t)(SF_FAST_PIX OP)) knob = NAME;
                CRTFX_KNOB_SETS(CRTFX_KNOB_NAME)
#undef CRTFX_KNOB_NAME
            }
            // ... and the defaults with the bloom switched off (--bloom-strength 0): a folded k_point_lean_seq (no half-resolution source at all)
            constexpr uint32_t SF_NOBLOOM = SF_FAST & ~(uint32_t)(CRTFX_F_BLOOM | CRTFX_F_BLOOM_FAST), SF_NOBLOOM_PIX = SF_NOBLOOM | CRTFX_F_PIXELATE;
            const bool nobloom = !folded_gates && c->pix_fmt == CRTFX_PIX_U8 && (gates_nt == SF_NOBLOOM || gates_nt == SF_NOBLOOM_PIX);
            const uint32_t core = gates & ~GRADE_RT_MASK;                    // ... without the purely arithmetic gates
            const bool grade_any = !folded_gates && (core == SF_FAST || core == SF_FAST_PIX);     // the defaults' loads + a grade / threshold / luma / flicker
            // ... of which: uint8 frames whose only extra gates are per-channel grade stages (no saturation) read a1 + a4 from the host's table
            const bool grade_lut = grade_any && c->pix_fmt == CRTFX_PIX_U8 && c->kp.grade_lut != nullptr &&
                                   (gates_nt & GRADE_RT_MASK & ~(uint32_t)(CRTFX_F_TEMPERATURE | CRTFX_F_BRIGHTCON | CRTFX_F_GAMMA)) == 0;
            const bool grade_rt = grade_any && !grade_lut;
            bool lean = lean_gates;
            KGroup kg{};
            KWarpGroup wg{};                     // warp on: the frames' FINAL outputs (the point kernels then only park pre-warp images)
            int gmax = n - i < MAX_GROUP ? n - i : MAX_GROUP;
            if (warp && gmax > c->pre_frames) gmax = c->pre_frames;
            for (; seq_ok && g < gmax; ++g) {
                const crtfx_frame* f = frames ? &frames[i + g] : nullptr;
                if ((fl & CRTFX_F_SCANLINES) && !(f && (f->scan_row_dev || f->scan_plane_dev))) break;
                if (f && f->glitch_offs_dev) break;
                const KOut ko = final_out(i + g);
                if (ko.blend != CRTFX_BLEND_RENDER && ko.blend != CRTFX_BLEND_NONE) break;
                if (g > 0 && ko.blend != (warp ? wg.o[0].blend : kg.o[0].blend)) break;       // a run = frames that all blend with their predecessor, or frames that do not blend at all
                kg.f[g] = make_kframe(frame_in(i + g), f);
                if (warp) {
                    wg.pre[g] = c->pre + (size_t)g * frame_elems; wg.o[g] = ko;
                    KOut k1{};
                    k1.pre = c->pre + (size_t)g * frame_elems; k1.pix = c->pix_fmt; k1.dbg = c->dbg;
                    kg.o[g] = k1;
                } else kg.o[g] = ko;
                if (kg.f[g].noise_plane || kg.f[g].overlay_before || kg.o[g].overlay_after) lean = false;
                if (kg.f[g].scan_plane) ++nplane;
            }
            const bool scan_knob = nplane > 0 && nplane == g && scan_ok && lean;
            if (nplane > 0 && !scan_knob) lean = false;
            if (g >= 2) {
                c->prof_this = c->prof && (c->prof_frame++ % (unsigned)c->prof_stride == 0);
                const bool pixelate = (gates & CRTFX_F_PIXELATE) != 0, f16 = c->pix_fmt == CRTFX_PIX_F16;
                // (g * 34 * (waves + 2) * 16 bytes of dynamic LDS for the tiles: 43.5 KB for 8 frames at 8 wavefronts)
                const bool fused = lean && can_fuse;
                const bool knob_build = fused && (knob != nullptr || scan_knob);      // (the one-knob folded builds exist for the fused kernel only; elsewhere: the run-time forms)
                char knob_name[40];
                if (knob_build) snprintf(knob_name, sizeof knob_name, "fast%s%s", pixelate ? "+pixelate" : "", scan_knob ? "+scan2d" : knob);
                const char* gname = (folded_gates && !scan_knob) ? sf_name(gates_nt) : knob_build ? knob_name : nobloom ? (pixelate ? "fast+pixelate-bloom" : "fast-bloom") : grade_lut ? (pixelate ? "fast+pixelate+gradelut" : "fast+gradelut")
                                    : grade_rt ? (pixelate ? "fast+pixelate+grade" : "fast+grade") : "runtime";
                const size_t fused_lds = (size_t)g * (TW / 2 + 2) * (waves + 2) * 16;
                if (fastb && !fused) {
                    dim3 gh((c->kp.hw + 63) / 64, (c->kp.hh + 3) / 4, g);
                    ProfEv ph(c, 2, g);
                    plan_note(c->plan.half, "k_half_group<%s,%s>", (lean && folded_gates && !thr) ? sf_name(gates) : "runtime", (lean && folded_gates && !thr) ? pix_name(c->pix_fmt) : "any");
                    if (!lean || !folded_gates || thr) { CRTFX_LAUNCH((k_half_group<SF_RUNTIME, 0>), gh, dim3(256), 0, s, ph.e0, ph.e1, c->kp, kg); }
                    else if (pixelate) { if (f16) { CRTFX_LAUNCH((k_half_group<SF_FAST_PIX, CRTFX_PIX_F16>), gh, dim3(256), 0, s, ph.e0, ph.e1, c->kp, kg); } else { CRTFX_LAUNCH((k_half_group<SF_FAST_PIX, CRTFX_PIX_U8>), gh, dim3(256), 0, s, ph.e0, ph.e1, c->kp, kg); } }
                    else { if (f16) { CRTFX_LAUNCH((k_half_group<SF_FAST, CRTFX_PIX_F16>), gh, dim3(256), 0, s, ph.e0, ph.e1, c->kp, kg); } else { CRTFX_LAUNCH((k_half_group<SF_FAST, CRTFX_PIX_U8>), gh, dim3(256), 0, s, ph.e0, ph.e1, c->kp, kg); } }
                }
                ProfEv pe(c, 0, g);
                c->plan.group = g;
                if (lean) plan_note(c->plan.point, "%s<%s,%s,%s>", fused ? "k_point_fused_seq" : "k_point_lean_seq", gname, pix_name(c->pix_fmt), blend_name(kg.o[0].blend == CRTFX_BLEND_RENDER ? CRTFX_BLEND_RENDER : CRTFX_BLEND_NONE));
                else plan_note(c->plan.point, "k_point_sel_seq<%s,%s>", pix_name(c->pix_fmt), (!(fl & CRTFX_F_PIXELATE) && !fastb) ? "one-round" : "two-round");
                if (lean) {
                    dim3 gp((c->W + TW - 1) / TW, (c->H + waves * CRTFX_POINT_ROWS - 1) / (waves * CRTFX_POINT_ROWS));
#define CRTFX_SEQ(SFV, PIXV)                                                                                                                             \
                    do {                                                                                                                                     \
                        if (kg.o[0].blend == CRTFX_BLEND_RENDER) { CRTFX_LAUNCH((k_point_lean_seq<SFV, PIXV, CRTFX_BLEND_RENDER>), gp, dim3(64 * waves), 0, s, pe.e0, pe.e1, c->kp, kg, g); } \
                        else { CRTFX_LAUNCH((k_point_lean_seq<SFV, PIXV, CRTFX_BLEND_NONE>), gp, dim3(64 * waves), 0, s, pe.e0, pe.e1, c->kp, kg, g); }                      \
                    } while (0)
#define CRTFX_FSEQ(SFV, PIXV)                                                                                                                            \
                    do {                                                                                                                                     \
                        if (kg.o[0].blend == CRTFX_BLEND_RENDER) { CRTFX_LAUNCH((k_point_fused_seq<SFV, PIXV, CRTFX_BLEND_RENDER>), gp, dim3(64 * waves), fused_lds, s, pe.e0, pe.e1, c->kp, kg, g); } \
                        else { CRTFX_LAUNCH((k_point_fused_seq<SFV, PIXV, CRTFX_BLEND_NONE>), gp, dim3(64 * waves), fused_lds, s, pe.e0, pe.e1, c->kp, kg, g); }                     \
                    } while (0)
                    if (knob_build && scan_knob) { if (pixelate) CRTFX_FSEQ(SF_FAST_PIX | KF_SCANPLANE, CRTFX_PIX_U8); else CRTFX_FSEQ(SF_FAST | KF_SCANPLANE, CRTFX_PIX_U8); }
                    else if (knob_build && coarse_knob) { if (pixelate) CRTFX_FSEQ(SF_FAST_PIX | KF_COARSE, CRTFX_PIX_U8); else CRTFX_FSEQ(SF_FAST | KF_COARSE, CRTFX_PIX_U8); }
                    else if (knob_build) {
#define CRTFX_KNOB_LAUNCH(OP, NAME)                                                                                          \
                        if (gates_nt == (uint32_t)(SF_FAST OP)) CRTFX_FSEQ((uint32_t)(SF_FAST OP), CRTFX_PIX_U8);               \
                        else if (gates_nt == (uint32_t)(SF_FAST_PIX OP)) CRTFX_FSEQ((uint32_t)(SF_FAST_PIX OP), CRTFX_PIX_U8);  \
                        else
                        CRTFX_KNOB_SETS(CRTFX_KNOB_LAUNCH) {}
#undef CRTFX_KNOB_LAUNCH
                    }
                    else if (fused) {
                        if (grade_lut) { if (pixelate) CRTFX_FSEQ(SF_FAST_PIX | KF_GRADE_LUT, CRTFX_PIX_U8); else CRTFX_FSEQ(SF_FAST | KF_GRADE_LUT, CRTFX_PIX_U8); }
                        else if (grade_rt) {
                            if (pixelate) { if (f16) CRTFX_FSEQ(SF_FAST_PIX | KF_GRADE_RT, CRTFX_PIX_F16); else CRTFX_FSEQ(SF_FAST_PIX | KF_GRADE_RT, CRTFX_PIX_U8); }
                            else { if (f16) CRTFX_FSEQ(SF_FAST | KF_GRADE_RT, CRTFX_PIX_F16); else CRTFX_FSEQ(SF_FAST | KF_GRADE_RT, CRTFX_PIX_U8); }
                        }
                        else if (!folded_gates) { if (f16) CRTFX_FSEQ(SF_LEAN_RT, CRTFX_PIX_F16); else CRTFX_FSEQ(SF_LEAN_RT, CRTFX_PIX_U8); }
                        else if (pixelate) { if (f16) CRTFX_FSEQ(SF_FAST_PIX, CRTFX_PIX_F16); else CRTFX_FSEQ(SF_FAST_PIX, CRTFX_PIX_U8); }
                        else { if (f16) CRTFX_FSEQ(SF_FAST, CRTFX_PIX_F16); else CRTFX_FSEQ(SF_FAST, CRTFX_PIX_U8); }
                    }
                    else if (nobloom) { if (pixelate) CRTFX_SEQ(SF_NOBLOOM_PIX, CRTFX_PIX_U8); else CRTFX_SEQ(SF_NOBLOOM, CRTFX_PIX_U8); }
                    else if (grade_lut) { if (pixelate) CRTFX_SEQ(SF_FAST_PIX | KF_GRADE_LUT, CRTFX_PIX_U8); else CRTFX_SEQ(SF_FAST | KF_GRADE_LUT, CRTFX_PIX_U8); }
                    else if (grade_rt) {
                        if (pixelate) { if (f16) CRTFX_SEQ(SF_FAST_PIX | KF_GRADE_RT, CRTFX_PIX_F16); else CRTFX_SEQ(SF_FAST_PIX | KF_GRADE_RT, CRTFX_PIX_U8); }
                        else { if (f16) CRTFX_SEQ(SF_FAST | KF_GRADE_RT, CRTFX_PIX_F16); else CRTFX_SEQ(SF_FAST | KF_GRADE_RT, CRTFX_PIX_U8); }
                    }
                    else if (!folded_gates) { if (f16) CRTFX_SEQ(SF_LEAN_RT, CRTFX_PIX_F16); else CRTFX_SEQ(SF_LEAN_RT, CRTFX_PIX_U8); }
                    else if (pixelate) { if (f16) CRTFX_SEQ(SF_FAST_PIX, CRTFX_PIX_F16); else CRTFX_SEQ(SF_FAST_PIX, CRTFX_PIX_U8); }
                    else { if (f16) CRTFX_SEQ(SF_FAST, CRTFX_PIX_F16); else CRTFX_SEQ(SF_FAST, CRTFX_PIX_U8); }
#undef CRTFX_FSEQ
#undef CRTFX_SEQ
#undef CRTFX_KNOB_SETS
                } else {
                    dim3 gp((c->W + TW - 1) / TW, (c->H + waves - 1) / waves);
                    const bool one = !(fl & CRTFX_F_PIXELATE) && !fastb;
                    if (f16) {
                        if (one) { CRTFX_LAUNCH((k_point_sel_seq<CRTFX_PIX_F16, true>), gp, dim3(64 * waves), 0, s, pe.e0, pe.e1, c->kp, kg, g); }
                        else { CRTFX_LAUNCH((k_point_sel_seq<CRTFX_PIX_F16, false>), gp, dim3(64 * waves), 0, s, pe.e0, pe.e1, c->kp, kg, g); }
                    } else {
                        if (one) { CRTFX_LAUNCH((k_point_sel_seq<CRTFX_PIX_U8, true>), gp, dim3(64 * waves), 0, s, pe.e0, pe.e1, c->kp, kg, g); }
                        else { CRTFX_LAUNCH((k_point_sel_seq<CRTFX_PIX_U8, false>), gp, dim3(64 * waves), 0, s, pe.e0, pe.e1, c->kp, kg, g); }
                    }
                }
                if (warp) launch_warp_group(c, wg, g, false, s, wg.o[0].blend == CRTFX_BLEND_RENDER && g > 1);
                HIP_TRY(c, hipGetLastError());
                group_done();
                i += g;
                continue;
            }
            g = 0;
        }
        // ---- general path, one frame ------------------------------------------------------------------------
        int rc = run_chain(c, frame_in(i), frames ? &frames[i] : nullptr, final_out(i), s);
        if (rc) return rc;
        group_done();
        ++i;
    }
    if (n > 0) c->plan = best;

    if (c->overlap) {      // everything issued on the side stream is ordered before whatever the caller enqueues next
        for (int k = 0; k < 2; ++k)
            if (c->ev_k2_pending[k]) { HIP_TRY(c, hipStreamWaitEvent(s, c->ev_k2[k], 0)); c->ev_k2_pending[k] = false; }
    }
    if (local_states_base && n > 0)      // the carried state = the last frame's
        HIP_TRY(c, hipMemcpyAsync(state_inout_dev, local_states_base + (size_t)(n - 1) * frame_elems, frame_elems * sizeof(float),
                                  hipMemcpyDeviceToDevice, s));
    return CRTFX_OK;
}

int crtfx_noise_plane(crtfx_ctx* c, uint64_t seed, uint64_t frame_index, float* out_dev, void* stream) {
    if (!c || !out_dev) return CRTFX_E_INVALID;
    if (int rcdev = check_device(c)) return rcdev;
    uint32_t k0, k1;
    noise_keys(seed, frame_index, k0, k1);
    const int n = c->H * c->W;
    hipLaunchKernelGGL(k_noise_plane, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, k0, k1, out_dev);
    HIP_TRY(c, hipGetLastError());
    return CRTFX_OK;
}

int crtfx_warp_map(crtfx_ctx* c, int32_t* ix_dev, int32_t* iy_dev, int32_t* fxy_dev, void* stream) {
    if (!c || !ix_dev || !iy_dev || !fxy_dev) return CRTFX_E_INVALID;
    if (int rcdev = check_device(c)) return rcdev;
    if (!c->params_set || !(c->kp.flags & CRTFX_F_WARP)) return fail(c, CRTFX_E_INVALID, "warp is not enabled in the current params");
    hipLaunchKernelGGL(k_warp_map, dim3((c->W + 255) / 256, c->H), dim3(256), 0, (hipStream_t)stream, c->kp, ix_dev, iy_dev, fxy_dev);
    HIP_TRY(c, hipGetLastError());
    return CRTFX_OK;
}

int crtfx_scanline_plane(crtfx_ctx* c, double strength, double omega, double phase_px, double tan_theta, double inv_sharp,
                         float* out_dev, void* stream) {
    if (!c) return CRTFX_E_INVALID;
    if (int rcdev = check_device(c)) return rcdev;
    if (!out_dev) return fail(c, CRTFX_E_INVALID, "out_dev is NULL");
    hipLaunchKernelGGL(k_scan_plane, dim3((c->W + 255) / 256, c->H), dim3(256), 0, (hipStream_t)stream, c->H, c->W, strength, omega,
                       phase_px, tan_theta, inv_sharp, out_dev);
    HIP_TRY(c, hipGetLastError());
    return CRTFX_OK;
}

int crtfx_resize_state(crtfx_ctx* c, const float* src_dev, int src_h, int src_w, float* dst_dev, void* stream) {
    if (!c) return CRTFX_E_INVALID;
    if (int rcdev = check_device(c)) return rcdev;
    if (!src_dev || !dst_dev || src_h <= 0 || src_w <= 0) return fail(c, CRTFX_E_INVALID, "bad resize arguments");
    if (!c->params_set) return fail(c, CRTFX_E_INVALID, "crtfx_set_params has not been called");
    if (src_h > 32767 || src_w > 32767) return fail(c, CRTFX_E_UNSUPPORTED, "state of %dx%d exceeds 32767", src_w, src_h);
    const double sx = (double)src_w / c->W, sy = (double)src_h / c->H;
    dim3 grid((c->W + 255) / 256, c->H);
    // the reference's state is float64 exactly when its chain promotes (vignette / flicker: DESIGN.md 3)
    if (c->kp.flags & (CRTFX_F_VIGNETTE | CRTFX_F_FLICKER))
        hipLaunchKernelGGL(k_resize_state<double>, grid, dim3(256), 0, (hipStream_t)stream, src_dev, src_h, src_w, dst_dev, c->H, c->W, sx, sy);
    else
        hipLaunchKernelGGL(k_resize_state<float>, grid, dim3(256), 0, (hipStream_t)stream, src_dev, src_h, src_w, dst_dev, c->H, c->W, sx, sy);
    HIP_TRY(c, hipGetLastError());
    return CRTFX_OK;
}

int crtfx_host_blur_row(const float* row_in, float* row_out, int w, int cn, const float* taps, int ntaps) {
    if (!row_in || !row_out || !taps || w <= 0 || cn <= 0 || ntaps <= 0 || !(ntaps & 1)) return CRTFX_E_INVALID;
    const int r = ntaps / 2;
    for (int x = 0; x < w; ++x)
        for (int ch = 0; ch < cn; ++ch) {
            float s = 0.0f;
            for (int k = 0; k < ntaps; ++k) {
                int xx = x + k - r;
                xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
                s = __builtin_fmaf(row_in[(size_t)xx * cn + ch], taps[k], s);
            }
            row_out[(size_t)x * cn + ch] = s;
        }
    return CRTFX_OK;
}

int crtfx_kernel_lds_bytes(const char* build, int radius, int pix_fmt) {
    if (!build || (pix_fmt != CRTFX_PIX_U8 && pix_fmt != CRTFX_PIX_F16)) return CRTFX_E_INVALID;
    if (!strcmp(build, "k_phosphor_ct")) {
        if (radius < 1 || radius > (pix_fmt == CRTFX_PIX_F16 ? CT_HALF_MAX_RADIUS : CT_MAX_RADIUS)) return CRTFX_E_UNSUPPORTED;
        return ct_lds_words(radius, pix_fmt) * 4;
    }
    if (!strcmp(build, "k_phosphor_cc")) {
        if (radius < 1 || radius > RR_MAX_RADIUS) return CRTFX_E_UNSUPPORTED;
        return cc_lds_words(radius, pix_fmt) * 4;
    }
    return CRTFX_E_UNSUPPORTED;
}

int crtfx_last_plan(crtfx_ctx* c, char* buf, size_t n) {
    if (!c || !buf || n == 0) return CRTFX_E_INVALID;
    const crtfx_ctx::Plan& p = c->plan;
    size_t o = 0;
    buf[0] = 0;
    auto add = [&](const char* key, const char* val) {
        if (!val[0] || o + 1 >= n) return;
        const int w = snprintf(buf + o, n - o, "%s%s=%s", o ? ";" : "", key, val);
        if (w > 0) o = (o + (size_t)w < n) ? o + (size_t)w : n - 1;
    };
    char num[32];
    add("phosphor", p.phosphor);
    add("blur", p.blur);
    add("half", p.half);
    add("point", p.point);
    if (p.phosphor[0] || p.point[0]) {
        snprintf(num, sizeof num, "%d", p.group); add("group", num);
        if (p.phosphor[0]) {
            snprintf(num, sizeof num, "%d", p.seg_rows); add("seg_rows", num);
            snprintf(num, sizeof num, "%d", c->group_max); add("group_max", num);      // the planner's frames per grid (a batch's last group may be shorter)
        }
    }
    add("warp", p.warp);
    if (p.warp[0]) { snprintf(num, sizeof num, "%d", p.warp_frames); add("warp_frames", num); }
    return CRTFX_OK;
}

int crtfx_set_option(crtfx_ctx* c, int option, int value) {
    if (!c) return CRTFX_E_INVALID;
    switch (option) {
    case CRTFX_OPT_FORCE_GENERIC: c->force_generic = value != 0; break;
    case CRTFX_OPT_FORCE_RUNTIME_FLAGS: c->force_runtime_flags = value != 0; break;
    case CRTFX_OPT_NO_CC: c->no_cc = value != 0; break;
    case CRTFX_OPT_NO_CT: c->no_ct = value != 0; break;
    case CRTFX_OPT_NO_FUSED_HALF: c->no_fused_half = value != 0; break;
    case CRTFX_OPT_NO_PLAIN_WARP: c->no_plain_warp = value != 0; break;
    case CRTFX_OPT_BAND_MB: if (value < -1 || value > 4096) return fail(c, CRTFX_E_INVALID, "band_mb %d outside -1..4096", value); c->band_mb = value; break;
    case CRTFX_OPT_FORCE_CC: c->force_cc = value != 0; break;
    case CRTFX_OPT_SPLIT_SRC_PLANE: c->split_src_plane = value != 0; break;
    case CRTFX_OPT_SPLIT_FROM: if (value < 0) return fail(c, CRTFX_E_INVALID, "split_from %d < 0", value); c->split_from = value; break;
    case CRTFX_OPT_GROUP: if (value < 0 || value > MAX_GROUP) return fail(c, CRTFX_E_INVALID, "group %d outside 0..%d", value, MAX_GROUP); c->opt_group = value; break;
    case CRTFX_OPT_SEG_ROWS: if (value < 0) return fail(c, CRTFX_E_INVALID, "seg_rows %d < 0", value); c->opt_seg_rows = value ? ((value + NB - 1) / NB) * NB : 0; break;
    case CRTFX_OPT_WARP_ROWS: if (value != 0 && value != 1 && value != 2 && value != 4) return fail(c, CRTFX_E_INVALID, "warp rows must be 0 (automatic), 1, 2 or 4"); c->warp_rows = value; break;
    case CRTFX_OPT_POINT_TILES: if (value < 0 || value > 16) return fail(c, CRTFX_E_INVALID, "point tiles outside 0..16"); c->point_tiles = value; break;
    case CRTFX_OPT_DEBUG_PLAN: c->debug_plan = value != 0; break;
    case CRTFX_OPT_OVERLAP:
        if (value && !c->side) {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) {
                HIP_TRY(c, hipEventCreateWithFlags(&c->ev_k1[i], hipEventDisableTiming));
                HIP_TRY(c, hipEventCreateWithFlags(&c->ev_k2[i], hipEventDisableTiming));
            }
        }
        c->overlap = value != 0;
        break;
    default: return fail(c, CRTFX_E_INVALID, "unknown option %d", option);
    }
    c->params_set = false;      // launch shapes are planned in crtfx_set_params: the caller sets the parameters again
    return CRTFX_OK;
}

int crtfx_debug_buffer(crtfx_ctx* c, void* dev_ptr) {
#ifdef CRTFX_STAMP
    if (!c) return CRTFX_E_INVALID;
    c->dbg = static_cast<unsigned long long*>(dev_ptr);
    return CRTFX_OK;
#else
    (void)dev_ptr;
    return fail(c, CRTFX_E_UNSUPPORTED, "phase stamps need a -DCRTFX_STAMP build of libcrtfx");
#endif
}

int crtfx_profile_enable(crtfx_ctx* c, int on) {
    if (!c) return CRTFX_E_INVALID;
    c->prof = on != 0;
    c->prof_stride = on > 1 ? on : 1;      // on = N > 1: sample every N-th frame (a timed dispatch costs ~8 % when every launch is timed)
    c->prof_frame = 0;
    c->prof_this = c->prof;
    c->ev_used[0] = c->ev_used[1] = c->ev_used[2] = 0;
    return CRTFX_OK;
}

int crtfx_profile_read(crtfx_ctx* c, int kernel, double* mean_launch_ms, int* launches, int* frames) {
    if (!c || kernel < 0 || kernel > 2 || !mean_launch_ms || !launches) return CRTFX_E_INVALID;
    const size_t u = c->ev_used[kernel];
    double total = 0.0;
    int cnt = 0, fr = 0;
    for (size_t i = 0; i + 1 < u; i += 2) {
        HIP_TRY(c, hipEventSynchronize(c->ev[kernel][i + 1]));
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[kernel][i], c->ev[kernel][i + 1]));
        total += ms; ++cnt;
        fr += (i / 2 < c->ev_frames[kernel].size()) ? c->ev_frames[kernel][i / 2] : 1;      // a grouped launch covers several frames
    }
    *mean_launch_ms = cnt ? total / cnt : 0.0;
    *launches = cnt;
    if (frames) *frames = fr;
    c->ev_used[kernel] = 0;
    return CRTFX_OK;
}

}  // extern "C"
