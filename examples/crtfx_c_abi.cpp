// A consumer of the C-ABI alone (no Python, no torch): include/crtfx.h + the HIP runtime for device memory.
//
//   hipcc --offload-arch=gfx950 -Iinclude examples/crtfx_c_abi.cpp -Lpythoncrt_amd -lcrtfx -Wl,-rpath,$PWD/pythoncrt_amd -o build/crtfx_c_abi
//   build/crtfx_c_abi in.rgb W H N out.rgb
//
// Renders N raw rgb24 frames with chromatic aberration 2 px, vignette 0.4, barrel warp 0.2 and persistence 0.5 through
// crtfx_process_batch.  The host tables these stages need are closed-form IEEE expressions (ref:266-276, :336-339), so
// this program builds them itself and tests/test_cli_gpu.py::test_c_abi_consumer expects the same bytes as from the
// Python host (whose tables come from numpy).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "crtfx.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CRTCHK(x) do { int r_ = (x); if (r_ != CRTFX_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #x, r_, crtfx_last_error(ctx)); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc != 6) { std::fprintf(stderr, "usage: %s in.rgb W H N out.rgb\n", argv[0]); return 1; }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[4]);
    const size_t frame_bytes = (size_t)W * H * 3;
    std::vector<unsigned char> in(frame_bytes * N), out(frame_bytes * N);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(in.data(), 1, in.size(), f) != in.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    std::fclose(f);

    HIPCHK(hipSetDevice(0));
    crtfx_ctx* ctx = nullptr;
    if (crtfx_create(0, H, W, CRTFX_PIX_U8, &ctx) != CRTFX_OK) { std::fprintf(stderr, "crtfx_create failed\n"); return 3; }

    // vignette axes (ref:268-274): nx = (x - (W-1)/2) / max(1, W/2), squared, in float64
    std::vector<double> nx2(W), ny2(H);
    const double cxd = (W - 1) / 2.0, cyd = (H - 1) / 2.0, rx = std::max(1.0, W / 2.0), ry = std::max(1.0, H / 2.0);
    for (int x = 0; x < W; ++x) { const double n = (x - cxd) / rx; nx2[x] = n * n; }
    for (int y = 0; y < H; ++y) { const double n = (y - cyd) / ry; ny2[y] = n * n; }
    // barrel-warp axes (ref:336-339): float32 (x - cx) / max(1, cx)
    std::vector<float> xhat(W), yhat(H);
    const float cx = (float)((W - 1) / 2.0), cy = (float)((H - 1) / 2.0);
    for (int x = 0; x < W; ++x) xhat[x] = ((float)x - cx) / std::max(1.0f, cx);
    for (int y = 0; y < H; ++y) yhat[y] = ((float)y - cy) / std::max(1.0f, cy);

    crtfx_params p;
    std::memset(&p, 0, sizeof(p));
    p.size = sizeof(p);
    p.flags = CRTFX_F_VIGNETTE | CRTFX_F_WARP;
    p.aberration_px = 2;
    p.grain_size = 1;
    p.saturation = 1.0f; p.contrast = 1.0f;
    p.vignette_strength = 0.4;
    p.vig_nx2 = nx2.data(); p.vig_ny2 = ny2.data();
    p.warp_k = 0.5f * 0.2f; p.warp_cx = cx; p.warp_cy = cy;
    p.warp_xhat = xhat.data(); p.warp_yhat = yhat.data();
    CRTCHK(crtfx_set_params(ctx, &p));

    unsigned char *d_in = nullptr, *d_out = nullptr;
    float* d_state = nullptr;
    HIPCHK(hipMalloc(&d_in, in.size()));
    HIPCHK(hipMalloc(&d_out, out.size()));
    HIPCHK(hipMalloc(&d_state, frame_bytes * sizeof(float)));
    hipStream_t s;
    HIPCHK(hipStreamCreate(&s));
    HIPCHK(hipMemcpyAsync(d_in, in.data(), in.size(), hipMemcpyHostToDevice, s));
    std::vector<crtfx_frame> rec(N);
    for (int i = 0; i < N; ++i) { std::memset(&rec[i], 0, sizeof(crtfx_frame)); rec[i].flicker_factor = 1.0; rec[i].frame_index = (uint64_t)i; }
    CRTCHK(crtfx_process_batch(ctx, d_in, frame_bytes, d_out, frame_bytes, N, rec.data(), d_state, 0.5, /*first_has_state=*/0,
                               /*local_states_base=*/nullptr, s));
    HIPCHK(hipMemcpyAsync(out.data(), d_out, out.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    f = std::fopen(argv[5], "wb");
    if (!f || std::fwrite(out.data(), 1, out.size(), f) != out.size()) { std::fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
    std::fclose(f);
    unsigned long long sum = 0;
    for (unsigned char v : out) sum += v;
    std::printf("crtfx ABI v%d: %d frames of %dx%d, output checksum %llu\n", crtfx_version(), N, W, H, sum);
    char plan[256];
    CRTCHK(crtfx_last_plan(ctx, plan, sizeof plan));          // which kernel builds the call above landed on
    std::printf("plan: %s\n", plan);
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_state); (void)hipStreamDestroy(s);
    crtfx_destroy(ctx);
    return 0;
}
