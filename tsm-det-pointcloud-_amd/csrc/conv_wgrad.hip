// conv_wgrad.hip — weight gradient of the sparse convolution.
//
//   dw[co][k][ci] = sum_o dout[o, co] * in[pair[k][o], ci]
//
// The reduction axis (rule pairs of one offset k) is the MFMA K axis, so it can be COMPACTED: each wave
// scans 64 rule entries at a time, ballot-compacts the valid (input row, output row) pairs into a
// per-wave LDS queue and feeds v_mfma_f32_16x16x4_f32 four valid pairs per instruction — invalid pairs
// cost nothing, which matters because 50-90 % of a submanifold rulebook is -1.
//   A[i = ci][kk = pair p] = in  [src_row(p)][ci]     (16 lanes read 64 contiguous bytes of a row)
//   B[kk = pair p][j = co] = dout[dst_row(p)][co]
// grid = (S row-chunks, K offsets), one wave per block; each block owns a [cin, cout] accumulator for its
// (chunk, offset) and writes it to a partial slab; a second kernel sums the S partials in fixed order
// (no float atomics: bitwise reproducible) and emits the reference parameter layout [Cout][K][Cin].
//
// Serves the autograd of reference call sites spconv_backbone.py:86-121, triggered by loss.backward()
// at tools/train_utils/train_utils.py:53.
#include "spx_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGroups = 4;  // 64-row rule groups fetched per iteration (independent loads in flight)

// row-chunks per offset: enough one-wave blocks to fill the chip (S*K >= ~4 per SIMD), bounded by the partial-slab
// traffic (<= 32 MiB) and by at least one full iteration (64*kGroups rows) per block
static inline int wgrad_splits(int64_t n, int cin, int cout, int kvol) {
  int64_t s = n / (64 * kGroups);
  int64_t by_slab = (int64_t(32) << 20) / ((int64_t)kvol * cin * cout * 4);
  if (s > by_slab) s = by_slab;
  int64_t by_blocks = 4096 / kvol;          // ~4 one-wave blocks per SIMD is plenty; more only lengthens the reduce
  if (by_blocks < 8) by_blocks = 8;
  if (s > by_blocks) s = by_blocks;
  if (s < 1) s = 1;
  return (int)s;
}

template <int MI, int NJ>
__global__ __launch_bounds__(64) void k_wgrad_mfma(const float* __restrict__ in, int cin,
                                                   const float* __restrict__ dout, int cout,
                                                   const int32_t* __restrict__ pair, int64_t ld, int64_t n,
                                                   const int64_t* d_n, int64_t chunk, int S, int K, float* __restrict__ slab) {
  __shared__ int32_t q_src[64 * kGroups];
  __shared__ int32_t q_dst[64 * kGroups];
  const int lane = threadIdx.x;
  const int c = lane & 15, q = lane >> 4;
  // XCD-aware block -> (chunk, offset) map: blocks b and b+8 share an XCD (and its L2).  The K blocks of one row chunk
  // all read the same dout rows and neighbouring input rows, so they are given ids of one residue class mod 8 and
  // consecutive positions inside it: one XCD fetches the chunk once instead of up to K times (PMC before: 158 MB
  // fetched per 64->64 launch against 38 MB compulsory).
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int k = j % K;
  const int chunk_id = xcd + 8 * (j / K);
  if (chunk_id >= S) return;
  const int64_t nlive = spx_live_n(d_n, n);
  int64_t r0 = (int64_t)chunk_id * chunk;
  int64_t r1 = r0 + chunk < nlive ? r0 + chunk : nlive;

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) acc[mi][nj] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int32_t* prow = pair + (int64_t)k * ld;
  for (int64_t base = r0; base < r1; base += 64 * kGroups) {
    int32_t id[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      int64_t row = base + 64 * g + lane;
      id[g] = row < r1 ? prow[row] : -1;
    }
    int nvalid = 0;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      unsigned long long mask = __ballot(id[g] >= 0);
      int rank = nvalid + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      if (id[g] >= 0) {
        q_src[rank] = id[g];
        q_dst[rank] = (int32_t)(base + 64 * g + lane);
      }
      nvalid += __popcll(mask);
    }
    if (nvalid == 0) continue;
    __builtin_amdgcn_wave_barrier();
    for (int st = 0; st * 4 < nvalid; ++st) {
      int p = 4 * st + q;
      bool has = p < nvalid;
      int32_t sid = has ? q_src[p] : 0;
      int32_t did = has ? q_dst[p] : 0;
      float a[MI], b[NJ];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int ci = 16 * mi + c;
        a[mi] = (has && ci < cin) ? in[(size_t)sid * cin + ci] : 0.f;
      }
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
        int co = 16 * nj + c;
        b[nj] = (has && co < cout) ? dout[(size_t)did * cout + co] : 0.f;
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }

  // partial[s][k][ci][co]; C layout: row(ci) = 16mi + 4q + e, col(co) = 16nj + c
  float* out = slab + ((size_t)chunk_id * K + k) * cin * cout;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int ci = 16 * mi + 4 * q + e, co = 16 * nj + c;
        if (ci < cin && co < cout) out[(size_t)ci * cout + co] = acc[mi][nj][e];
      }
}

// dw[co][k][ci] = sum_s slab[s][k][ci][co]   (fixed order over s).  Block = 64 consecutive elements x 4 S-slices:
// coalesced 256-B reads, 4 independent partial sums per element, combined through LDS in slice order.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slab, int S, int K, int cin, int cout,
                                                      float* __restrict__ dw) {
  __shared__ float part[4][64];
  const int per = cin * cout;
  const int total = K * per;
  const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + e;
  float s = 0.f;
  if (t < total)
    for (int j = sl; j < S; j += 4) s += slab[(size_t)j * total + t];
  part[sl][e] = s;
  __syncthreads();
  if (sl == 0 && t < total) {
    float v = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
    int co = t % cout, ci = (t / cout) % cin, k = t / per;
    dw[((size_t)co * K + k) * cin + ci] = v;
  }
}

}  // namespace

extern "C" size_t spx_conv_wgrad_ws_bytes(int cin, int cout, int kvol, int64_t n_out) {
  return spx_align((size_t)wgrad_splits(n_out, cin, cout, kvol) * kvol * cin * cout * sizeof(float));
}

#define SPX_WG_CASE(A, B)                                                                                          \
  if (MI == A && NJ == B) {                                                                                        \
    hipLaunchKernelGGL((k_wgrad_mfma<A, B>), dim3(8 * kvol * ((S + 7) / 8)), dim3(64), 0, s, in, cin, dout, cout, pair, pair_ld, \
                       n_out, d_n_out, chunk, S, kvol, slab);                                                                      \
    launched = true;                                                                                               \
  }

extern "C" int spx_conv_wgrad(const float* in, int cin, const float* dout, int cout, int kvol, const int32_t* pair,
                              int64_t pair_ld, int64_t n_out, const int64_t* d_n_out, float* dw, void* ws,
                              size_t ws_bytes, spx_stream_t stream) {
  if (!in || !dout || !pair || !dw || cin <= 0 || cout <= 0 || kvol <= 0 || kvol > SPX_MAX_KVOL || n_out < 0 ||
      pair_ld < n_out)
    return SPX_ERR_INVALID_ARG;
  if (cin > 128 || cout > 128) return SPX_ERR_UNSUPPORTED;
  if (n_out >= (int64_t(1) << 31)) return SPX_ERR_TOO_LARGE;
  if (!ws || ws_bytes < spx_conv_wgrad_ws_bytes(cin, cout, kvol, n_out)) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  if (n_out == 0) {
    spx_fill_async(dw, 0, sizeof(float) * (size_t)cout * kvol * cin, s);
    return SPX_OK;
  }
  int S = wgrad_splits(n_out, cin, cout, kvol);
  int64_t chunk = ((n_out + S - 1) / S + 63) / 64 * 64;
  float* slab = reinterpret_cast<float*>(ws);
  int MI = (cin + 15) / 16, NJ = (cout + 15) / 16;
  if (MI == 3) MI = 4;
  if (MI > 4) MI = 8;
  if (NJ == 3) NJ = 4;
  if (NJ > 4) NJ = 8;
  bool launched = false;
  SPX_WG_CASE(1, 1)
  SPX_WG_CASE(1, 2)
  SPX_WG_CASE(1, 4)
  SPX_WG_CASE(1, 8)
  SPX_WG_CASE(2, 1)
  SPX_WG_CASE(2, 2)
  SPX_WG_CASE(2, 4)
  SPX_WG_CASE(2, 8)
  SPX_WG_CASE(4, 1)
  SPX_WG_CASE(4, 2)
  SPX_WG_CASE(4, 4)
  SPX_WG_CASE(4, 8)
  SPX_WG_CASE(8, 1)
  SPX_WG_CASE(8, 2)
  SPX_WG_CASE(8, 4)
  SPX_WG_CASE(8, 8)
  if (!launched) return SPX_ERR_UNSUPPORTED;
  int total = kvol * cin * cout;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((total + 63) / 64), dim3(256), 0, s, slab, S, kvol, cin, cout, dw);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
