// crtfx_internal.h — host-side declarations shared by the translation units of libcrtfx.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "crtfx_kernels.hip.h"

namespace crtfx {

// One per radius, each in its own translation unit (crtfx_rr.hip compiled with -DRR_R=n) so the
// thirty sets of instantiations build in parallel.  variant: 0 = runtime gates (uint8), 1 = SF_FULL gates folded (uint8), 2 = SF_FULL + half frames, 3 = runtime gates + half frames,
// 4 = k_phosphor_cc (SF_FULL gates, uint8 frames, pre-warp image out), 5 = SF_FULL + pixelate gates folded (uint8),
// 6 = k_phosphor_ct (as 4: composite triad tables, centre samples from the frame, scalar row tables), 7 = k_phosphor_ct<R, half> (as 6 for half frames: qword loads,
// a one-trip raw tile, the centre samples in a register window).
using rr_launch_fn = void (*)(const KParams&, const KGroup&, int seg_rows, dim3 grid, size_t lds, hipStream_t, int variant,
                              hipEvent_t ev_start, hipEvent_t ev_stop);

#define CRTFX_RR_RADII(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) \
                          X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30)
#define CRTFX_RR_DECL(r) void rr_launch_##r(const KParams&, const KGroup&, int, dim3, size_t, hipStream_t, int, hipEvent_t, hipEvent_t);
CRTFX_RR_RADII(CRTFX_RR_DECL)
#undef CRTFX_RR_DECL

constexpr int RR_MAX_RADIUS = 30;      // one build per radius up to here: bloom sigma <= 10, the reference GUI's range (ref:1475)
// Larger radii (the reference accepts any sigma, ref:609-610, :1231) run the split path (k_sb_rows / k_sb_cols + the
// pointwise chain, crtfx.hip launch_split_blur).  (Round 2 first served 31..128 with "bucket" builds of this kernel at
// R = 36 .. 128 on zero-padded taps: bit-exact too, but 380 us -> 29 ms per 4K frame against the split path's 344 -> 640.)
inline int rr_build_radius(int R) { return (R >= 1 && R <= RR_MAX_RADIUS) ? R : 0; }     // the compiled radius that serves bloom radius R (0 = none)

// Launch with optional timing events attached to the dispatch packet itself (hipExtLaunchKernelGGL):
// no extra packets on the stream.  Separate hipEventRecord calls around every kernel cost ~5 us of
// idle GPU per boundary (12 % of the 4K frame rate).
#define CRTFX_LAUNCH(kern, grid, block, lds, stream, ev0, ev1, ...)                                            \
    do {                                                                                                       \
        if (ev0) hipExtLaunchKernelGGL(kern, grid, block, lds, stream, ev0, ev1, 0, __VA_ARGS__);              \
        else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                  \
    } while (0)

}  // namespace crtfx
