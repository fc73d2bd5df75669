// crtfx_rr.hip — instantiations of k_phosphor_rr for ONE radius (-DRR_R=n): the gate-folded
// full-chain variant and the runtime-flag variant.  Compiled once per radius, in parallel.
#include <atomic>
#include "crtfx_internal.h"

#ifndef RR_R
#error "compile with -DRR_R=<radius 1..30>"
#endif

namespace crtfx {

#define CRTFX_CAT2(a, b) a##b
#define CRTFX_CAT(a, b) CRTFX_CAT2(a, b)

void CRTFX_CAT(rr_launch_, RR_R)(const KParams& kp, const KGroup& kg, int seg_rows, dim3 grid, size_t lds,
                                 hipStream_t s, int variant, hipEvent_t e0, hipEvent_t e1) {
    // a build that parks more than 64 KiB of LDS opts in once per kernel and device (none of the radii 1..30 does today)
#define CRTFX_BIG_LDS(kern)                                                                                              \
    do {                                                                                                                  \
        if (lds > 65536) {                                                                                                \
            static std::atomic<bool> done[64];                                                                            \
            int dev = 0;                                                                                                  \
            (void)hipGetDevice(&dev);                                                                                     \
            if (dev < 0 || dev >= 64 || !done[dev].load(std::memory_order_acquire)) {                                     \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                if (dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);                               \
            }                                                                                                             \
        }                                                                                                                 \
    } while (0)
#if RR_R <= 15
    if (variant == 6) {     // as variant 4 on the dword-load / composite-table build (k_phosphor_ct, radii 1 .. 15 = CT_MAX_RADIUS)
        CRTFX_LAUNCH((k_phosphor_ct<RR_R>), grid, dim3(RR_THREADS), lds, s, e0, e1, kp, kg, seg_rows);
        return;
    }
#endif
#if RR_R <= CT_HALF_MAX_RADIUS
    if (variant == 7) {     // as variant 6 for half frames: qword loads, a one-trip raw tile, the centre samples in a register window
        CRTFX_LAUNCH((k_phosphor_ct<RR_R, 1>), grid, dim3(RR_THREADS), lds, s, e0, e1, kp, kg, seg_rows);
        return;
    }
#endif
#if RR_R <= 30
    if (variant == 4) {     // full-chain gates, pre-warp image out: the column-owner kernel (uint8 frames)
        CRTFX_LAUNCH((k_phosphor_cc<RR_R, 0>), grid, dim3(RR_THREADS), lds, s, e0, e1, kp, kg, seg_rows);
        return;
    }
#endif
    if (variant == 5) {     // uint8 frames, full-chain gates + pixelate (the reference CLI's default pixel size 2 with --no-fast-bloom)
        CRTFX_LAUNCH((k_phosphor_rr<RR_R, SF_FULL | CRTFX_F_PIXELATE, 0>), grid, dim3(RR_THREADS), lds, s, e0, e1, kp, kg, seg_rows);
        return;
    }
    if (variant == 2) {     // half frames, full-chain gates
        CRTFX_BIG_LDS((k_phosphor_rr<RR_R, SF_FULL, 1>));
        CRTFX_LAUNCH((k_phosphor_rr<RR_R, SF_FULL, 1>), grid, dim3(RR_THREADS), lds, s, e0, e1, kp, kg, seg_rows);
    }
    else if (variant == 1) {
        CRTFX_BIG_LDS((k_phosphor_rr<RR_R, SF_FULL, 0>));
        CRTFX_LAUNCH((k_phosphor_rr<RR_R, SF_FULL, 0>), grid, dim3(RR_THREADS), lds, s, e0, e1, kp, kg, seg_rows);
    }
    else if (variant == 3) {     // half frames, runtime gates
        CRTFX_BIG_LDS((k_phosphor_rr<RR_R, SF_RUNTIME, 1>));
        CRTFX_LAUNCH((k_phosphor_rr<RR_R, SF_RUNTIME, 1>), grid, dim3(RR_THREADS), lds, s, e0, e1, kp, kg, seg_rows);
    }
    else {                       // runtime gates (large radii park up to 35 KB of graded centre pixels)
        CRTFX_BIG_LDS((k_phosphor_rr<RR_R, SF_RUNTIME, 0>));
        CRTFX_LAUNCH((k_phosphor_rr<RR_R, SF_RUNTIME, 0>), grid, dim3(RR_THREADS), lds, s, e0, e1, kp, kg, seg_rows);
    }
#undef CRTFX_BIG_LDS
}

}  // namespace crtfx
