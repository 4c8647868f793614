// conv_balanced.hip — sparse convolution forward / dgrad with an MFMA-work-balanced, persistent schedule.
//
// Same arithmetic as conv_gemm.hip (output-stationary implicit GEMM, v_mfma_f32_16x16x4_f32, same packed weights, same
// gather map).  What changes is WHO does which work.  In-kernel stamps on k_conv_mfma (tools/conv_diag.py) showed that a
// launch of one 16-row tile per wave ends with a long low-occupancy tail: ~5 tiles per SIMD, tiles differ in their
// number of non-empty offsets ("units"), a wave left alone on its SIMD keeps the MFMA pipe only 33-50 % busy, and the
// launch takes twice its MFMA floor.  Here
//   1. spx_conv_plan counts the units of every tile (a 27-bit mask per tile), prefix-sums them, and cuts the flat list
//      of all units into W equal ranges, W = the number of wave slots of the chip (<= U / K so that a tile is cut at
//      most once);
//   2. k_conv_mfma_pb runs exactly one wave per slot; wave w walks the tiles of its unit range; a tile that lies
//      entirely inside the range is written to `dst` directly, the (at most two) cut tiles at its ends go to a scratch
//      buffer as partial sums;
//   3. k_conv_fixup adds the two partial sums of every cut tile (head part + tail part, a fixed order) and applies the
//      epilogue.  No atomics, no in-kernel hand-off: results are bitwise reproducible from run to run.
// The plan depends only on the rule table, so the python layer caches it per rulebook (forward, dgrad and the second
// layer of a submanifold pair all reuse it).
//
// Serves the same reference call sites as conv_gemm.hip (spconv_backbone.py:86-121 and their autograd).
#include <stdlib.h>

#include "spx_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 3072;        // wave slots used: 3 per SIMD (256 CUs x 4 SIMDs); the 64x64 kernel holds a whole
                                    // unit's weight fragments (64 VGPRs) + two row sets in registers (160 VGPRs)
constexpr int kHdr = 4;             // plan header: [0] active waves nw, [1] total units U, [2] tiles T, [3] reserved

// plan layout (int32): hdr[kHdr] | wstart[kWaves + 1] | mask[Tcap] | pre[Tcap + 1] | split[Tcap]
__host__ __device__ inline int64_t plan_off_wstart() { return kHdr; }
__host__ __device__ inline int64_t plan_off_mask() { return kHdr + kWaves + 1; }
__host__ __device__ inline int64_t plan_off_pre(int64_t tcap) { return plan_off_mask() + tcap; }
__host__ __device__ inline int64_t plan_off_split(int64_t tcap) { return plan_off_pre(tcap) + tcap + 1; }
__host__ __device__ inline int64_t plan_ints(int64_t tcap) { return plan_off_split(tcap) + tcap; }

// ---------------------------------------------------------------- plan, pass 1: unit mask per tile
// One wave looks at 4 tiles (lane = 16*sub + r): bit k of mask[t] = some row of tile t has a neighbour at table row k.
__global__ __launch_bounds__(256) void k_plan_mask(const int32_t* __restrict__ pair, int64_t ld, int K, int64_t n,
                                                   const int64_t* d_n, int64_t tcap, int32_t* __restrict__ plan) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, r = lane & 15;
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t T = (nlive + 15) / 16;
  const int64_t tile = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub;
  const int64_t row = tile * 16 + r;
  const bool ok = tile < T && row < nlive;
  uint32_t m = 0;
  for (int k0 = 0; k0 < K; k0 += 9) {
    int32_t id[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) id[u] = (ok && k0 + u < K) ? pair[(int64_t)(k0 + u) * ld + row] : -1;
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const unsigned long long b = __ballot(id[u] >= 0);
      if ((b >> (16 * sub)) & 0xFFFFull) m |= 1u << (k0 + u);
    }
  }
  if (r == 0 && tile < tcap) plan[plan_off_mask() + tile] = tile < T ? (int32_t)m : 0;
}

// ---------------------------------------------------------------- plan, pass 2 (one block): prefix, cuts, split flags
__global__ __launch_bounds__(1024) void k_plan_scan(int K, int64_t n, const int64_t* d_n, int64_t tcap,
                                                    int32_t* __restrict__ plan) {
  __shared__ int s_part[1024];
  __shared__ int s_total;
  const int tid = threadIdx.x;
  const int64_t nlive = spx_live_n(d_n, n);
  const int T = (int)((nlive + 15) / 16);
  const int32_t* mask = plan + plan_off_mask();
  int32_t* pre = plan + plan_off_pre(tcap);
  int32_t* split = plan + plan_off_split(tcap);
  int32_t* wstart = plan + plan_off_wstart();
  // exclusive prefix of popcount(mask) over tiles: each thread owns a contiguous chunk
  const int per = (T + 1023) / 1024;
  const int lo = tid * per, hi = lo + per < T ? lo + per : T;
  int sum = 0;
  for (int t = lo; t < hi; ++t) sum += __popc((unsigned)mask[t]);
  s_part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 1024; ++i) {
      int v = s_part[i];
      s_part[i] = run;
      run += v;
    }
    s_total = run;
  }
  __syncthreads();
  int run = s_part[tid];
  for (int t = lo; t < hi; ++t) {
    pre[t] = run;
    run += __popc((unsigned)mask[t]);
    split[t] = 0;
  }
  const int U = s_total;
  if (tid == 0) {
    pre[T] = U;
    int nw = U / (K > 0 ? K : 1);
    if (nw > kWaves) nw = kWaves;
    if (nw < 1) nw = U > 0 ? 1 : 0;
    plan[0] = nw;
    plan[1] = U;
    plan[2] = T;
    plan[3] = 0;
  }
  __syncthreads();   // pre[], split[] written by this block are visible to it after the barrier
  const int nw = plan[0];
  // wave w owns units [w*U/nw, (w+1)*U/nw); wstart[w] = first tile with pre[t+1] > u0 (the tile holding unit u0)
  for (int w = tid; w <= kWaves; w += 1024) {
    int t0 = T;
    if (w < nw) {
      const int u0 = (int)((int64_t)w * U / nw);
      int a = 0, b = T;                        // smallest t in [0, T) with pre[t + 1] > u0
      while (a < b) {
        int mid = (a + b) >> 1;
        if (pre[mid + 1] > u0) b = mid; else a = mid + 1;
      }
      t0 = a;
      if (w > 0 && u0 > pre[t0]) split[t0] = 1;   // the cut falls strictly inside tile t0
    }
    wstart[w] = t0;
  }
}

// ---------------------------------------------------------------- persistent balanced implicit GEMM
template <int CS, int CD>
__global__ __launch_bounds__(256) void k_conv_mfma_pb(const float* __restrict__ src, const float* __restrict__ wp,
                                                      const int32_t* __restrict__ pair, int64_t ld, int K, int flip,
                                                      int64_t n, const int64_t* d_n, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu,
                                                      const int32_t* __restrict__ plan, int64_t tcap,
                                                      float* __restrict__ dst, float* __restrict__ scratch) {
  constexpr int NT = CD / 16;
  constexpr int JG = CS / 16;
  extern __shared__ char occupancy_pad[];   // sized by the launch so that exactly 3 workgroups fit a CU
  (void)occupancy_pad;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = plan[0];
  if (w >= nw) return;
  const int U = plan[1], T = plan[2];
  const int64_t nlive = spx_live_n(d_n, n);
  const int32_t* mask = plan + plan_off_mask();
  const int32_t* pre = plan + plan_off_pre(tcap);
  const int u0 = (int)((int64_t)w * U / nw), u1 = (int)((int64_t)(w + 1) * U / nw);
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);

  for (int t = plan[plan_off_wstart() + w]; t < T; ++t) {
    const int p0 = pre[t];
    if (p0 >= u1) break;
    const uint32_t m = (uint32_t)mask[t];
    const int cnt = __popc(m);
    const int first = u0 > p0 ? u0 - p0 : 0;            // units [first, last) of this tile, in loop order
    const int last = u1 - p0 < cnt ? u1 - p0 : cnt;
    if (first >= last) continue;
    const int64_t row_base = (int64_t)t * 16;
    const int64_t row = row_base + r;

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // units [first, last) of this tile as a bit set over the LOOP index k (weights W_k, table row trow(k))
    uint32_t todo = 0;
    {
      int j = 0;
      for (int k = 0; k < K; ++k) {
        const int trow = flip ? K - 1 - k : k;
        if (!((m >> trow) & 1u)) continue;
        if (j >= first && j < last) todo |= 1u << k;
        ++j;
      }
    }
    const int64_t row_c = row < nlive ? row : nlive - 1;     // clamped: the load is unconditional
    auto load_id = [&](int k) -> int32_t {
      const int kk = k >= 0 ? k : 0;
      const int32_t v = pair[(int64_t)(flip ? K - 1 - kk : kk) * ld + row_c];
      return (k >= 0 && row < nlive) ? v : -1;
    };
    // unconditional (index clamped; lanes without a neighbour are zeroed when the rows are USED): a predicated load
    // is a branch, and behind a branch the compiler waits vmcnt(0), i.e. for the weight fragments as well
    auto gather = [&](int32_t id, f32x4 (&a)[JG]) {
      const float* p = src + (size_t)(id > 0 ? id : 0) * CS + 4 * q;
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) a[jg] = *reinterpret_cast<const f32x4*>(p + 16 * jg);
    };
    auto pop = [&]() -> int {          // next unit in loop order, -1 when none
      if (todo == 0) return -1;
      const int k = __ffs((int)todo) - 1;
      todo &= todo - 1;
      return k;
    };
    // Per unit the wave would otherwise wait out five memory round trips in sequence (rule entry, then weights + rows
    // for each of the four 16-channel groups; ~14,000 cycles per unit against 2,048 of MFMA, measured).  Here the rule
    // entries run two units ahead, the gathered rows one unit ahead, and the 16 weight fragments of the unit are
    // requested together, in front of a scheduling fence, so that only the first fragment's latency is exposed.
    int k_cur = pop();
    int k_nxt = pop();
    int32_t id_cur = load_id(k_cur), id_nxt = load_id(k_nxt);
    f32x4 a_cur[JG], a_nxt[JG];
    gather(id_cur, a_cur);
    while (k_cur >= 0) {
      f32x4 b[JG][NT];
#pragma unroll
      for (int jg = 0; jg < JG; ++jg)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[jg][nt] = wp4[((size_t)(k_cur * NT + nt) * JG + jg) * 64 + lane];
      const int k_nn = pop();
      const int32_t id_nn = load_id(k_nn);
      gather(id_nxt, a_nxt);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const float keep = id_cur >= 0 ? 1.f : 0.f;
#pragma unroll
      for (int jg = 0; jg < JG; ++jg)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float av = id_cur >= 0 ? a_cur[jg][e] : 0.f;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[jg][nt][e], acc[nt], 0, 0, 0);
        }
      (void)keep;
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) a_cur[jg] = a_nxt[jg];
      k_cur = k_nxt;
      k_nxt = k_nn;
      id_cur = id_nxt;
      id_nxt = id_nn;
    }

    // write-out.  C layout: col = lane&15, row = 4*(lane>>4) + e
    const bool whole = first == 0 && last == cnt;
    if (whole) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * nt + r;
        const float sc = scale ? scale[col] : 1.0f;
        const float sh = shift ? shift[col] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t orow = row_base + 4 * q + e;
          if (orow < nlive) {
            float v = acc[nt][e];
            if (scale || shift) v = v * sc + sh;
            if (relu) v = v > 0.f ? v : 0.f;
            dst[orow * CD + col] = v;
          }
        }
      }
    } else {
      // partial sum of a cut tile: slot 2t = head part (units from 0), slot 2t+1 = tail part (units up to cnt)
      float* out = scratch + ((size_t)2 * t + (first == 0 ? 0 : 1)) * (16 * CD);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) out[(4 * q + e) * CD + 16 * nt + r] = acc[nt][e];
    }
  }
}

// ---------------------------------------------------------------- cut tiles: head part + tail part, epilogue
template <int CD>
__global__ __launch_bounds__(256) void k_conv_fixup(const int32_t* __restrict__ plan, int64_t tcap, int64_t n,
                                                    const int64_t* d_n, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, int relu,
                                                    const float* __restrict__ scratch, float* __restrict__ dst) {
  constexpr int V = 16 * CD / 4;                       // float4 pieces per tile
  const int T = plan[2];
  const int t = blockIdx.x;
  if (t >= T) return;
  const int32_t m = plan[plan_off_mask() + t];
  const bool cut = plan[plan_off_split(tcap) + t] != 0;
  if (!cut && m != 0) return;                          // written whole by the main kernel
  const int64_t nlive = spx_live_n(d_n, n);
  const f32x4* h = reinterpret_cast<const f32x4*>(scratch + (size_t)2 * t * (16 * CD));
  const f32x4* tl = h + V;
  for (int i = threadIdx.x; i < V; i += 256) {
    const int rr = (4 * i) / CD, c0 = (4 * i) % CD;
    const int64_t orow = (int64_t)t * 16 + rr;
    if (orow >= nlive) continue;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (cut) {
      const f32x4 a = h[i], b = tl[i];
      v = a + b;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = v[e];
      if (scale || shift) x = x * (scale ? scale[c0 + e] : 1.0f) + (shift ? shift[c0 + e] : 0.0f);
      if (relu) x = x > 0.f ? x : 0.f;
      v[e] = x;
    }
    *reinterpret_cast<f32x4*>(dst + orow * CD + c0) = v;
  }
}

static inline int64_t tiles_cap(int64_t n) { return (n + 15) / 16 + 1; }

template <int CS, int CD>
static void launch_pb(const float* src, const float* wp, const int32_t* pair, int64_t ld, int K, int flip, int64_t n,
                      const int64_t* d_n, const float* scale, const float* shift, int relu, const int32_t* plan,
                      float* dst, float* scratch, hipStream_t s) {
  const int64_t tcap = tiles_cap(n);
  // 48 KiB of dynamic LDS per workgroup: three workgroups (12 waves, 3 per SIMD) fit a CU's 160 KiB, a fourth does not
  hipLaunchKernelGGL((k_conv_mfma_pb<CS, CD>), dim3(kWaves / 4), dim3(256), 48 * 1024, s, src, wp, pair, ld, K, flip, n, d_n,
                     scale, shift, relu, plan, tcap, dst, scratch);
  hipLaunchKernelGGL((k_conv_fixup<CD>), dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, plan, tcap, n, d_n, scale, shift,
                     relu, scratch, dst);
}

}  // namespace

extern "C" size_t spx_conv_plan_bytes(int64_t n_dst) {
  return spx_align((size_t)plan_ints(tiles_cap(n_dst < 0 ? 0 : n_dst)) * sizeof(int32_t));
}

extern "C" int spx_conv_plan(const int32_t* pair, int64_t pair_ld, int kvol, int64_t n_dst, const int64_t* d_n_dst,
                             int32_t* plan, spx_stream_t stream) {
  if (!pair || !plan || kvol <= 0 || kvol > SPX_MAX_KVOL || n_dst <= 0 || pair_ld < n_dst) return SPX_ERR_INVALID_ARG;
  if (n_dst >= (int64_t(1) << 31)) return SPX_ERR_TOO_LARGE;
  hipStream_t s = spx_s(stream);
  const int64_t tcap = tiles_cap(n_dst);
  hipLaunchKernelGGL(k_plan_mask, dim3((unsigned)((tcap + 15) / 16)), dim3(256), 0, s, pair, pair_ld, kvol, n_dst, d_n_dst,
                     tcap, plan);
  hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, s, kvol, n_dst, d_n_dst, tcap, plan);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

extern "C" size_t spx_conv_gemm_balanced_ws_bytes(int c_dst, int64_t n_dst) {
  if (c_dst <= 0 || n_dst <= 0) return 0;
  return spx_align((size_t)2 * tiles_cap(n_dst) * 16 * c_dst * sizeof(float));
}

#define SPX_PB_CASE(A, B)                                                                                            \
  if (c_src == A && c_dst == B) {                                                                                    \
    launch_pb<A, B>(src, w_packed, pair, pair_ld, kvol, flip_k, n_dst, d_n_dst, scale, shift, relu, plan, dst, scratch, \
                    s);                                                                                              \
    SPX_CHECK_LAUNCH();                                                                                              \
    return SPX_OK;                                                                                                   \
  }

extern "C" int spx_conv_gemm_balanced(const float* src, int c_src, const float* w_packed, int c_dst, int kvol, int flip_k,
                                      const int32_t* pair, int64_t pair_ld, int64_t n_dst, const int64_t* d_n_dst,
                                      const float* scale, const float* shift, int relu, const int32_t* plan, float* dst,
                                      void* ws, size_t ws_bytes, spx_stream_t stream) {
  if (!src || !w_packed || !pair || !dst || !plan || c_src <= 0 || c_dst <= 0 || kvol <= 0 || kvol > SPX_MAX_KVOL ||
      n_dst <= 0 || pair_ld < n_dst)
    return SPX_ERR_INVALID_ARG;
  if (n_dst >= (int64_t(1) << 31)) return SPX_ERR_TOO_LARGE;
  if (!ws || ws_bytes < spx_conv_gemm_balanced_ws_bytes(c_dst, n_dst)) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  float* scratch = reinterpret_cast<float*>(ws);
  SPX_PB_CASE(32, 32)
  SPX_PB_CASE(32, 64)
  SPX_PB_CASE(64, 32)
  SPX_PB_CASE(64, 64)
  return SPX_ERR_UNSUPPORTED;   // other channel pairs: use spx_conv_gemm
}
