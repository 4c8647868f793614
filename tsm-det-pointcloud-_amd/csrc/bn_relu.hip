// bn_relu.hip — BatchNorm1d (+ReLU) over the rows of a sparse feature matrix, training mode.
//
// Restates nn.BatchNorm1d(C, eps, momentum) followed by nn.ReLU on SparseConvTensor.features [N, C], as the reference
// builds them at pcdet/models/backbones_3d/spconv_backbone.py:81,26-27: batch statistics over ALL active rows of the
// batch, biased variance for normalisation, unbiased for the running estimate.
// The stock kernels cost three launches forward and three backward per layer and stream [N,C] at ~0.4 TB/s; here
//   forward : k_bn_stats (one coalesced float4 pass, per-block partial sums) -> k_bn_finalize (fp64 combine, running
//             statistics update) -> k_bn_apply (normalise + affine + ReLU, one read one write)
//   backward: k_bn_reduce<true> (sum dz, sum dz*xhat; the ReLU mask is RECOMPUTED from x, y is not read back) ->
//             k_bn_bwd_finalize -> k_bn_bwd_apply (dx)
// Partials are combined in a fixed order (no float atomics): bitwise reproducible.
// With a residual (`res`, the identity branch of SparseBasicBlock, reference spconv_backbone.py:56-72: bn2, add, ReLU)
// the same kernels compute y = relu(bn(x) + res) and, backward, hand the masked gradient to the identity branch (dres):
// the add and the second ReLU cost no pass of their own.
#include "spx_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxBlocks = 1024;

// z = (x - mean) * invstd * gamma + beta, written so that forward and backward execute the SAME instruction sequence:
// the backward pass recomputes the ReLU mask (z > 0) from x instead of reading y back (one [N, C] pass less in the
// reduce and one less in the apply kernel), and the mask must agree with the forward bit for bit.
__device__ __forceinline__ f32x4 bn_affine(f32x4 v, f32x4 mu, f32x4 is, f32x4 ga, f32x4 be) {
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float t = __fmul_rn(__fsub_rn(v[e], mu[e]), is[e]);
    o[e] = __builtin_fmaf(t, ga[e], be[e]);
  }
  return o;
}

// bn_affine (+ residual element i when there is one): the pre-activation value of forward and backward alike
__device__ __forceinline__ f32x4 bn_out(f32x4 v, f32x4 mu, f32x4 is, f32x4 ga, f32x4 be, const float* __restrict__ res,
                                        int64_t i) {
  f32x4 o = bn_affine(v, mu, is, ga, be);
  if (res) {
    const f32x4 r = reinterpret_cast<const f32x4*>(res)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(o[e], r[e]);
  }
  return o;
}

// element offset of float4 piece i of a [N][C] matrix stored with row stride ld (C is a power of two: C | 1024)
__device__ __forceinline__ int64_t strided4(int64_t i, int C, int cshift, int64_t ld) {
  const int64_t e = i * 4;
  return (e >> cshift) * ld + (e & (C - 1));
}

// grid-stride over float4 elements of x[N][C]; because 256*4 % C == 0 is NOT assumed, each thread recomputes its channel
template <bool BWD>
__global__ __launch_bounds__(256) void k_bn_reduce(const float* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ dy,
                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                   int64_t n, const int64_t* d_n, int C, int relu,
                                                   const float* __restrict__ res, int64_t dy_ld, int cshift,
                                                   float* __restrict__ partial /*[grid][2][C]*/) {
  __shared__ float sm[256][8];   // per-thread partials (4 channels x {a, b})
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t total4 = nlive * C / 4;
  // every thread keeps the same 4 channels for its whole walk when the stride (grid*256*4) is a multiple of C;
  // the launcher guarantees that (C divides 1024)
  const int64_t start = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int c0 = (int)((start * 4) % C);
  f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 mu = f32x4{0.f, 0.f, 0.f, 0.f}, is = f32x4{1.f, 1.f, 1.f, 1.f};
  f32x4 ga = f32x4{1.f, 1.f, 1.f, 1.f}, be = f32x4{0.f, 0.f, 0.f, 0.f};
  if (BWD) {
    mu = *reinterpret_cast<const f32x4*>(mean + c0);
    is = *reinterpret_cast<const f32x4*>(invstd + c0);
    ga = *reinterpret_cast<const f32x4*>(gamma + c0);
    be = *reinterpret_cast<const f32x4*>(beta + c0);
  }
  for (int64_t i = start; i < total4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    if (!BWD) {
      a += v;
      b += v * v;
    } else {
      f32x4 g = *reinterpret_cast<const f32x4*>(dy + strided4(i, C, cshift, dy_ld));
      if (relu) {
        const f32x4 o = bn_out(v, mu, is, ga, be, res, i);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
      }
      a += g;
      b += g * ((v - mu) * is);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sm[threadIdx.x][e] = a[e];
    sm[threadIdx.x][4 + e] = b[e];
  }
  __syncthreads();
  // threads t, t + C/4, t + 2C/4, ... hold the same 4 channels: one thread per channel adds them in a fixed order
  const int G = C / 4;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / 4, e = c % 4;
    float sa = 0.f, sb = 0.f;
    for (int t = g; t < 256; t += G) {
      sa += sm[t][e];
      sb += sm[t][4 + e];
    }
    partial[(size_t)blockIdx.x * 2 * C + c] = sa;
    partial[(size_t)blockIdx.x * 2 * C + C + c] = sb;
  }
}

// one wave per channel: lane l adds partials l, l+64, ... in fp64, then a fixed-order shuffle tree
__device__ __forceinline__ void wave_sum2(double& s, double& q) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    s += __shfl_down(s, d, 64);
    q += __shfl_down(q, d, 64);
  }
}

// The finalize kernels sit on the critical path between two passes over the map, 52 times per training step; with one wave per
// channel every lane walked up to 16 partial rows one dependent load after the other (6.7 us forward, 13 us backward beside the
// side-stream kernels).  Four waves per channel: every thread has at most four rows (issued together), then a fixed-order
// combine — wave shuffle tree, the four wave sums added in wave order.
constexpr int kFinThreads = 256;
__device__ __forceinline__ void block_sum2(const float* __restrict__ partial, int nblk, int C, int c, double& s, double& q) {
  __shared__ double sh[2][kFinThreads / 64];
  double ps[4], pq[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int b = threadIdx.x + u * kFinThreads;
    const bool ok = b < nblk;
    const size_t o = (size_t)(ok ? b : 0) * 2 * C + c;
    const float vs = partial[o], vq = partial[o + C];
    ps[u] = ok ? (double)vs : 0.0;
    pq[u] = ok ? (double)vq : 0.0;
  }
  s = (ps[0] + ps[1]) + (ps[2] + ps[3]);
  q = (pq[0] + pq[1]) + (pq[2] + pq[3]);
  for (int b = threadIdx.x + 4 * kFinThreads; b < nblk; b += kFinThreads) {     // (more than 1024 rows: the Winograd partials)
    s += (double)partial[(size_t)b * 2 * C + c];
    q += (double)partial[(size_t)b * 2 * C + C + c];
  }
  wave_sum2(s, q);
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = s;
    sh[1][threadIdx.x >> 6] = q;
  }
  __syncthreads();
  s = ((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3];
  q = ((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3];
}

__global__ __launch_bounds__(kFinThreads) void k_bn_finalize(const float* __restrict__ partial, int nblk, int C, int64_t n,
                                                    const int64_t* d_n, float eps, float momentum, float* __restrict__ mean,
                                                    float* __restrict__ invstd, float* __restrict__ running_mean,
                                                    float* __restrict__ running_var, int64_t* __restrict__ nbt) {
  const int c = blockIdx.x;
  double s, q;
  block_sum2(partial, nblk, C, c, s, q);
  if (threadIdx.x != 0) return;
  if (nbt && c == 0) *nbt += 1;              // nn.BatchNorm's num_batches_tracked, without a launch of its own
  const double N = (double)spx_live_n(d_n, n);
  double m = N > 0 ? s / N : 0.0;
  double var = N > 0 ? q / N - m * m : 0.0;
  if (var < 0) var = 0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    double unbiased = N > 1 ? var * N / (N - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, const float* __restrict__ mean,
                                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int64_t n, const int64_t* d_n, int C,
                                                  int relu, const float* __restrict__ res, int64_t y_ld, int cshift,
                                                  float* __restrict__ y) {
  const int64_t total4 = spx_live_n(d_n, n) * C / 4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    const int c0 = (int)((i * 4) % C);
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0), is = *reinterpret_cast<const f32x4*>(invstd + c0);
    f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
    f32x4 o = bn_out(v, mu, is, ga, be, res, i);
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(y + strided4(i, C, cshift, y_ld)) = o;
  }
}

__global__ __launch_bounds__(kFinThreads) void k_bn_bwd_finalize(const float* __restrict__ partial, int nblk, int C,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x;
  double s, q;
  block_sum2(partial, nblk, C, c, s, q);
  if (threadIdx.x == 0) {
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ beta,
                                                      const float* __restrict__ dy, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                      int64_t n, const int64_t* d_n, int C, int relu,
                                                      const float* __restrict__ res, int64_t dy_ld, int cshift,
                                                      float* __restrict__ dx, float* __restrict__ dres) {
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t total4 = nlive * C / 4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const float invN = 1.0f / (float)(nlive > 0 ? nlive : 1);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    const int c0 = (int)((i * 4) % C);
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 g = *reinterpret_cast<const f32x4*>(dy + strided4(i, C, cshift, dy_ld));
    f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0), is = *reinterpret_cast<const f32x4*>(invstd + c0);
    f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0);
    if (relu) {
      const f32x4 o = bn_out(v, mu, is, ga, *reinterpret_cast<const f32x4*>(beta + c0), res, i);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    }
    if (dres) reinterpret_cast<f32x4*>(dres)[i] = g;
    f32x4 dg = *reinterpret_cast<const f32x4*>(dgamma + c0), db = *reinterpret_cast<const f32x4*>(dbeta + c0);
    f32x4 xh = (v - mu) * is;
    reinterpret_cast<f32x4*>(dx)[i] = ga * is * (g - db * invN - xh * dg * invN);
  }
}

static inline int bn_blocks(int64_t n, int C) {
  int64_t total4 = n * C / 4;
  int64_t nb = (total4 + 256 * 8 - 1) / (256 * 8);
  if (nb > kMaxBlocks) nb = kMaxBlocks;
  if (nb < 1) nb = 1;
  // stride = nb*256 float4 = nb*1024 floats must be a multiple of C so that a thread keeps its channels: C | 1024
  return (int)nb;
}

}  // namespace

extern "C" size_t spx_bn_relu_ws_bytes(int c) { return spx_align((size_t)kMaxBlocks * 2 * c * 4); }

static inline int log2_of(int c) {
  int sft = 0;
  while ((1 << sft) < c) ++sft;
  return sft;
}

extern "C" int spx_bn_add_relu_fwd(const float* x, const float* res, int64_t n, const int64_t* d_n, int c, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                   float momentum, float eps, int relu, float* y, int64_t y_ld, float* save_mean,
                                   float* save_invstd, void* ws, size_t ws_bytes, spx_stream_t stream) {
  if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || n < 0 || c <= 0) return SPX_ERR_INVALID_ARG;
  if (c % 4 != 0 || 1024 % c != 0) return SPX_ERR_UNSUPPORTED;   // 4,8,16,...,1024
  if (y_ld == 0) y_ld = c;
  if (y_ld < c || y_ld % 4 != 0 || ((uintptr_t)y & 15) != 0) return SPX_ERR_INVALID_ARG;
  if (!ws || ws_bytes < spx_bn_relu_ws_bytes(c)) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  float* partial = reinterpret_cast<float*>(ws);
  int nb = bn_blocks(n, c);
  const int cshift = log2_of(c);
  hipLaunchKernelGGL((k_bn_reduce<false>), dim3(nb), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr,
                     nullptr, n, d_n, c, relu, nullptr, (int64_t)c, cshift, partial);
  hipLaunchKernelGGL(k_bn_finalize, dim3(c), dim3(kFinThreads), 0, s, partial, nb, c, n, d_n, eps, momentum, save_mean,
                     save_invstd, running_mean, running_var, num_batches_tracked);
  if (n > 0)
    hipLaunchKernelGGL(k_bn_apply, dim3(nb), dim3(256), 0, s, x, save_mean, save_invstd, gamma, beta, n, d_n, c, relu, res,
                       y_ld, cshift, y);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

// Training-mode BatchNorm (+ReLU) whose per-channel sums were already taken by the PRODUCER of x — csrc/wino_conv2d.hip
// writes, per block of 32 output tiles, the sums of y and y*y it stores (partial[nblk][2][C]) — so the statistics pass over x
// is skipped: finalize (fp64 combine of the rows, running statistics) + apply.
extern "C" int spx_bn_relu_fwd_from_sums(const float* x, int64_t n, const int64_t* d_n, int c, const float* partial, int64_t nblk,
                                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                                         int64_t* num_batches_tracked, float momentum, float eps, int relu, float* y,
                                         int64_t y_ld, float* save_mean, float* save_invstd, spx_stream_t stream) {
  if (!x || !partial || !gamma || !beta || !y || !save_mean || !save_invstd || n <= 0 || c <= 0 || nblk <= 0)
    return SPX_ERR_INVALID_ARG;
  if (c % 4 != 0 || 1024 % c != 0) return SPX_ERR_UNSUPPORTED;
  if (nblk > 0x7fffffff) return SPX_ERR_TOO_LARGE;
  if (y_ld == 0) y_ld = c;
  if (y_ld < c || y_ld % 4 != 0 || ((uintptr_t)y & 15) != 0) return SPX_ERR_INVALID_ARG;
  hipStream_t s = spx_s(stream);
  const int cshift = log2_of(c);
  hipLaunchKernelGGL(k_bn_finalize, dim3(c), dim3(kFinThreads), 0, s, partial, (int)nblk, c, n, d_n, eps, momentum, save_mean,
                     save_invstd, running_mean, running_var, num_batches_tracked);
  hipLaunchKernelGGL(k_bn_apply, dim3(bn_blocks(n, c)), dim3(256), 0, s, x, save_mean, save_invstd, gamma, beta, n, d_n, c,
                     relu, nullptr, y_ld, cshift, y);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

extern "C" int spx_bn_add_relu_bwd(const float* x, const float* res, const float* dy, int64_t dy_ld, int64_t n,
                                   const int64_t* d_n, int c, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                                   int relu, float* dx, float* dres, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                   spx_stream_t stream) {
  if (!x || !dy || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || n <= 0 || c <= 0)
    return SPX_ERR_INVALID_ARG;
  if (c % 4 != 0 || 1024 % c != 0) return SPX_ERR_UNSUPPORTED;
  if (dy_ld == 0) dy_ld = c;
  if (dy_ld < c || dy_ld % 4 != 0 || ((uintptr_t)dy & 15) != 0) return SPX_ERR_INVALID_ARG;
  if (!ws || ws_bytes < spx_bn_relu_ws_bytes(c)) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  float* partial = reinterpret_cast<float*>(ws);
  int nb = bn_blocks(n, c);
  const int cshift = log2_of(c);
  hipLaunchKernelGGL((k_bn_reduce<true>), dim3(nb), dim3(256), 0, s, x, gamma, beta, dy, save_mean, save_invstd, n,
                     d_n, c, relu, res, dy_ld, cshift, partial);
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(c), dim3(kFinThreads), 0, s, partial, nb, c, dgamma, dbeta);
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(nb), dim3(256), 0, s, x, beta, dy, save_mean, save_invstd, gamma, dgamma, dbeta,
                     n, d_n, c, relu, res, dy_ld, cshift, dx, dres);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

extern "C" int spx_bn_relu_fwd(const float* x, int64_t n, const int64_t* d_n, int c, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, int relu, float* y,
                               float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, spx_stream_t stream) {
  return spx_bn_add_relu_fwd(x, nullptr, n, d_n, c, gamma, beta, running_mean, running_var, nullptr, momentum, eps, relu, y,
                             c, save_mean, save_invstd, ws, ws_bytes, stream);
}

extern "C" int spx_bn_relu_bwd(const float* x, const float* dy, int64_t n, const int64_t* d_n, int c, const float* gamma,
                               const float* beta,
                               const float* save_mean, const float* save_invstd, int relu, float* dx, float* dgamma,
                               float* dbeta, void* ws, size_t ws_bytes, spx_stream_t stream) {
  return spx_bn_add_relu_bwd(x, nullptr, dy, c, n, d_n, c, gamma, beta, save_mean, save_invstd, relu, dx, nullptr, dgamma, dbeta, ws,
                             ws_bytes, stream);
}

// Inference-mode BatchNorm (+ residual) (+ ReLU) over feature rows: y = relu?((x - mean) * invstd * gamma + beta (+ res)) with
// GIVEN statistics (the running estimates), one read and one write — nn.BatchNorm2d(eval) + nn.ReLU of the BEV backbone
// (reference base_bev_backbone.py:35-44,60-73) are two passes in torch.  Same row-strided output as the training kernels.
extern "C" int spx_bn_apply(const float* x, const float* res, int64_t n, const int64_t* d_n, int c, const float* mean,
                            const float* invstd, const float* gamma, const float* beta, int relu, float* y, int64_t y_ld,
                            spx_stream_t stream) {
  if (!x || !mean || !invstd || !gamma || !beta || !y || n < 0 || c <= 0) return SPX_ERR_INVALID_ARG;
  if (c % 4 != 0 || 1024 % c != 0) return SPX_ERR_UNSUPPORTED;
  if (y_ld == 0) y_ld = c;
  if (y_ld < c || y_ld % 4 != 0 || ((uintptr_t)y & 15) != 0) return SPX_ERR_INVALID_ARG;
  if (n == 0) return SPX_OK;
  hipLaunchKernelGGL(k_bn_apply, dim3(bn_blocks(n, c)), dim3(256), 0, spx_s(stream), x, mean, invstd, gamma, beta, n, d_n, c,
                     relu, res, y_ld, log2_of(c), y);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
