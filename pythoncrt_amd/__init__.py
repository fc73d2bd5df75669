"""pythoncrt_amd — MI355X-native per-frame CRT effect chain behind PythonCRT's own API.

    from pythoncrt_amd import apply_crt_effect, apply_static_effects, make_triad_mask, make_vignette
    from pythoncrt_amd import process_frames      # the loop of process_video (ref:1037-1131) over the caller's frame iterator and writer

See DESIGN.md (path, kernels, roofline) and INTEGRATION.md (how the reference binds to it).
"""
from .effects import (DeviceState, TriadMask, VignetteMask, apply_crt_effect, apply_static_effects, make_triad_mask,
                      make_vignette)
from .render import iter_rgb24, process_frames

__all__ = ["DeviceState", "TriadMask", "VignetteMask", "apply_crt_effect", "apply_static_effects", "make_triad_mask", "make_vignette",
           "process_frames", "iter_rgb24"]
