// wino_probe.hip — dev probe (GPU box): times k_wino_conv of csrc/wino_conv2d.hip stand-alone on the BEV shapes
// (no torch, no dispatcher: the kernel's own time under hipEvents).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DWINO_FORCE_NB=2] tools/hip/wino_probe.hip -o /tmp/wino_probe && /tmp/wino_probe
// (the -DWINO_NO_* / -DWINO_STAMP builds that located the round-2 bottlenecks lived in the kernel source while it was being
// written — profiles/r02_conv_experiments.md has what they measured — and were removed with it finished)
#include "../../tsm-det-pointcloud-_amd/csrc/wino_conv2d.hip"

#include <stdio.h>
#include <vector>

int main() {
  const int shapes[2][4] = {{4, 128, 200, 176}, {4, 256, 100, 88}};
  // -DWINO_SMALL: a map that stays in L2 (is the input fetch what the transform waits for?)
#ifdef WINO_SMALL
  const_cast<int&>(shapes[0][0]) = 1; const_cast<int&>(shapes[0][2]) = 100; const_cast<int&>(shapes[0][3]) = 88;
#endif
  for (int si = 0; si < 2; ++si) {
    const int n = shapes[si][0], c = shapes[si][1], h = shapes[si][2], w = shapes[si][3];
    const size_t px = (size_t)n * h * w;
    float *x, *y, *u;
    hipMalloc(&x, px * c * 4);
    hipMalloc(&y, px * c * 4);
    hipMalloc(&u, (size_t)16 * c * c * 4);
    std::vector<float> hx(px * c), hu((size_t)16 * c * c);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    for (size_t i = 0; i < hu.size(); ++i) hu[i] = (float)((i * 40503u >> 4) & 0xfff) / 4096.f - 0.5f;
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(u, hu.data(), hu.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) spx_conv2d_wino(x, c, u, n, h, w, c, c, nullptr, nullptr, 0, y, c, nullptr, nullptr);
    hipDeviceSynchronize();
    const int iters = 20;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) spx_conv2d_wino(x, c, u, n, h, w, c, c, nullptr, nullptr, 0, y, c, nullptr, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, gf = 2.0 * 9 * c * c * (double)px / 1e9;
    printf("%dx%dx%dx%d: %.1f us  (%.1f TF/s direct-equivalent, %.1f TF/s executed)\n", n, c, h, w, us, gf / us * 1e3,
           gf / 2.25 / us * 1e3);
    hipFree(x); hipFree(y); hipFree(u);
  }
  return 0;
}
