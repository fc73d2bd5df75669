// crtfx_internal.h — host-side declarations shared by the translation units of libcrtfx.so.
#pragma once
#include <hip/hip_runtime.h>
#include "crtfx_kernels.hip.h"

namespace crtfx {

// One per radius, each in its own translation unit (crtfx_rr.hip compiled with -DRR_R=n) so the
// twelve sets of instantiations build in parallel.  variant: 0 = runtime gates (uint8), 1 = SF_FULL gates folded (uint8), 2 = SF_FULL + half frames.
using rr_launch_fn = void (*)(const KParams&, const KFrame&, const KOut&, int seg_rows, dim3 grid, size_t lds, hipStream_t, int variant);

#define CRTFX_RR_DECL(r) void rr_launch_##r(const KParams&, const KFrame&, const KOut&, int, dim3, size_t, hipStream_t, int);
CRTFX_RR_DECL(1) CRTFX_RR_DECL(2) CRTFX_RR_DECL(3) CRTFX_RR_DECL(4) CRTFX_RR_DECL(5) CRTFX_RR_DECL(6)
CRTFX_RR_DECL(7) CRTFX_RR_DECL(8) CRTFX_RR_DECL(9) CRTFX_RR_DECL(10) CRTFX_RR_DECL(11) CRTFX_RR_DECL(12)
#undef CRTFX_RR_DECL

constexpr int RR_MAX_RADIUS = 12;

}  // namespace crtfx
