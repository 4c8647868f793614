// densify.hip — SparseConvTensor.dense() and its backward (pure HBM scatter / gather).
//
// layout 0: dense[b][c][z][y][x]   (contiguous NCDHW — what spconv's .dense() returns).  A 64-row x C tile
//           is read coalesced, transposed through LDS, and written with the ROW on the lane: rows are in
//           ascending (b,z,y,x) order after a strided conv, so lanes hit runs of consecutive x.
// layout 1: dense[b][y][x][c][z]   (same logical [B,C,D,H,W] tensor; view(B,C*D,H,W) is then channels_last
//           with BEV channel c*D+z fastest).  Channel on the lane; both z-slices of a BEV pixel interleave
//           into one contiguous C*D*4-byte line.
// Replaces reference call site pcdet/models/backbones_2d/map_to_bev/height_compression.py:21.
#include "spx_common.h"

namespace {

constexpr int kRows = 64;

template <bool BWD>
__global__ __launch_bounds__(256) void k_densify_ncdhw(float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                       int64_t n, const int64_t* d_n, int C, int batch, Int3 shape,
                                                       float* __restrict__ dense) {
  extern __shared__ float tile[];  // [kRows][C+1]
  __shared__ int64_t cell[kRows];
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t row0 = (int64_t)blockIdx.x * kRows;
  if (row0 >= nlive) return;
  const int nr = (int)((nlive - row0) < kRows ? (nlive - row0) : kRows);
  const int64_t vol = (int64_t)shape.v[0] * shape.v[1] * shape.v[2];
  const int ldt = C + 1;
  if (threadIdx.x < kRows) {
    int64_t off = -1;
    if ((int)threadIdx.x < nr) {
      int4 c = reinterpret_cast<const int4*>(idx)[row0 + threadIdx.x];
      if ((unsigned)c.x < (unsigned)batch && (unsigned)c.y < (unsigned)shape.v[0] &&
          (unsigned)c.z < (unsigned)shape.v[1] && (unsigned)c.w < (unsigned)shape.v[2])
        off = (int64_t)c.x * C * vol + ((int64_t)c.y * shape.v[1] + c.z) * shape.v[2] + c.w;
    }
    cell[threadIdx.x] = off;
  }
  if (!BWD) {
    for (int t = threadIdx.x; t < nr * C; t += 256) tile[(t / C) * ldt + (t % C)] = feat[row0 * C + t];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t off = cell[lane];
  if (!BWD) {
    if (lane < nr && off >= 0)
      for (int c = wave; c < C; c += 4) dense[off + (int64_t)c * vol] = tile[lane * ldt + c];
  } else {
    for (int c = wave; c < C; c += 4) tile[lane * ldt + c] = (lane < nr && off >= 0) ? dense[off + (int64_t)c * vol] : 0.f;
    __syncthreads();
    for (int t = threadIdx.x; t < nr * C; t += 256) feat[row0 * C + t] = tile[(t / C) * ldt + (t % C)];
  }
}

template <bool BWD>
__global__ void k_densify_cl(float* __restrict__ feat, const int32_t* __restrict__ idx, int64_t n, const int64_t* d_n,
                             int C, int batch, Int3 shape, float* __restrict__ dense) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = t / C;
  int c = (int)(t % C);
  if (row >= spx_live_n(d_n, n)) return;
  int4 q = reinterpret_cast<const int4*>(idx)[row];
  bool ok = (unsigned)q.x < (unsigned)batch && (unsigned)q.y < (unsigned)shape.v[0] &&
            (unsigned)q.z < (unsigned)shape.v[1] && (unsigned)q.w < (unsigned)shape.v[2];
  int64_t off = ((((int64_t)q.x * shape.v[1] + q.z) * shape.v[2] + q.w) * C + c) * shape.v[0] + q.y;
  if (!BWD) {
    if (ok) dense[off] = feat[t];
  } else {
    feat[t] = ok ? dense[off] : 0.f;
  }
}

template <bool BWD>
static int run(float* feat, const int32_t* idx, int64_t n, const int64_t* d_n, int c, int batch, const int32_t* shape,
               int layout, float* dense, hipStream_t s) {
  if (!feat || !idx || !dense || !shape || n < 0 || c <= 0 || batch <= 0) return SPX_ERR_INVALID_ARG;
  if (layout != 0 && layout != 1) return SPX_ERR_INVALID_ARG;
  if (n >= (int64_t(1) << 31)) return SPX_ERR_TOO_LARGE;
  if (n == 0) return SPX_OK;
  if (layout == 0) {
    if (c > 1024) return SPX_ERR_UNSUPPORTED;
    size_t lds = sizeof(float) * (size_t)kRows * (c + 1);
    unsigned nb = (unsigned)((n + kRows - 1) / kRows);
    hipLaunchKernelGGL((k_densify_ncdhw<BWD>), dim3(nb), dim3(256), lds, s, feat, idx, n, d_n, c, batch, spx_i3(shape),
                       dense);
  } else {
    int64_t total = n * c;
    unsigned nb = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL((k_densify_cl<BWD>), dim3(nb), dim3(256), 0, s, feat, idx, n, d_n, c, batch, spx_i3(shape),
                       dense);
  }
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

}  // namespace

extern "C" int spx_densify(const float* feat, const int32_t* idx, int64_t n, const int64_t* d_n, int c, int batch,
                           const int32_t* shape, int layout, float* dense, spx_stream_t stream) {
  return run<false>(const_cast<float*>(feat), idx, n, d_n, c, batch, shape, layout, dense, spx_s(stream));
}

extern "C" int spx_densify_bwd(const float* ddense, const int32_t* idx, int64_t n, const int64_t* d_n, int c,
                               int batch, const int32_t* shape, int layout, float* dfeat, spx_stream_t stream) {
  return run<true>(dfeat, idx, n, d_n, c, batch, shape, layout, const_cast<float*>(ddense), spx_s(stream));
}
