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

//
// Parts: crtfx_common.hip.h (parameter blocks, per-pixel stages), crtfx_blur.hip.h (split bloom), crtfx_point.hip.h (pointwise chain),
// crtfx_phosphor.hip.h + crtfx_phosphor_ct.hip.h (fused bloom chain), crtfx_warp.hip.h (warp + utility kernels).  Kernels only the main translation unit
// needs sit behind CRTFX_MAIN_TU; crtfx_rr.hip (one TU per radius) sees k_phosphor_rr / k_phosphor_cc and their helpers.
#pragma once
#include "crtfx_common.hip.h"
#include "crtfx_blur.hip.h"
#include "crtfx_point.hip.h"
#include "crtfx_phosphor.hip.h"
#include "crtfx_phosphor_ct.hip.h"
#include "crtfx_warp.hip.h"
