// l2_stream_probe.hip — dev probe (GPU box): how fast can every CU stream the SAME small (L2-resident) image with
// wave-contiguous 1 KiB loads?  This is the B-fragment traffic of csrc/wino_conv2d.hip: 8 waves per CU, each reading its own
// 8 KiB slice of a 64 KiB chunk, chunk after chunk through a 1 MiB image that all workgroups share.
//   hipcc --offload-arch=gfx950 -O3 tools/hip/l2_stream_probe.hip -o /tmp/l2_stream_probe && /tmp/l2_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, bool ROT>
__global__ void __launch_bounds__(512) k_stream(const float* __restrict__ img, int nchunk, int reps, float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* base = img + wave * 2048 + lane * 4;   // 8 KiB per wave per chunk
  f32x4 acc = {0, 0, 0, 0};
  const int rot = ROT ? (int)((blockIdx.x >> 3) % nchunk) : 0;
  for (int r = 0; r < reps; ++r)
    for (int c = 0; c < nchunk; ++c) {
      int cc = c + rot;
      if (cc >= nchunk) cc -= nchunk;
      const float* p = base + (size_t)cc * 16384;
      f32x4 v[DEPTH];
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (i % 8) * 256);
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) acc += v[i];
    }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) out[threadIdx.x] = acc[0];
}

template <int DEPTH, bool ROT>
static void run(const float* img, float* out, int nchunk, const char* name) {
  const int reps = 20, grid = 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_stream<DEPTH, ROT><<<grid, 512>>>(img, nchunk, 2, out);
  hipDeviceSynchronize();
  hipEventRecord(e0, nullptr);
  k_stream<DEPTH, ROT><<<grid, 512>>>(img, nchunk, reps, out);
  hipEventRecord(e1, nullptr);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * reps * nchunk * 8 * DEPTH * 1024;
  printf("%-28s image %4d KiB: %.2f TB/s  (%.1f GB/s per CU, %.1f B/clk/CU at 2.1 GHz)\n", name, nchunk * 64, bytes / ms * 1e-9,
         bytes / ms * 1e-6 / 256, bytes / ms * 1e-6 / 256 / 2.1);
}

int main() {
  float *img, *out;
  hipMalloc(&img, 8 << 20);
  hipMalloc(&out, 4096);
  hipMemset(img, 0, 8 << 20);
  for (int nchunk : {16, 64}) {
    run<8, false>(img, out, nchunk, "8 loads in flight/wave");
    run<8, true>(img, out, nchunk, "8 loads, rotated per WG");
    run<16, false>(img, out, nchunk, "16 loads in flight/wave");
    run<16, true>(img, out, nchunk, "16 loads, rotated per WG");
  }
  return 0;
}
