// conv_gemm.hip — sparse convolution forward / dgrad as an output-stationary implicit GEMM.
//
//   dst[o, :] = sum_k  src[pair[k][o], :] . W_k          (W_k : [c_src, c_dst])
//
// One wave owns 16*MT destination rows and ALL c_dst columns, so there is no atomic, no scatter and
// no cross-wave reduction: each destination row is written exactly once, in a fixed summation order
// (bitwise reproducible).  For every kernel offset k the wave
//   1. reads its rows' source indices pair[k][rows] (coalesced 64 B),
//   2. skips the offset entirely when no row of the wave has a neighbour there (wave-uniform ballot),
//      and skips 16-row M-tiles that have none,
//   3. gathers the source rows straight into the MFMA A-operand registers: the four lanes (r, q=0..3) of row r
//      load the four 16-byte pieces of ONE 64-byte sector of that row per instruction (16 sectors per wave
//      instruction; an earlier mapping that gave each lane its own quarter of the row touched 64),
//   4. issues v_mfma_f32_16x16x4_f32 (exact fp32, the reference's precision: spconv fp32, no TF32).
// The GEMM K axis is PERMUTED so that each lane's operand values are contiguous in memory: k-step j = 4jg + e of lane
// quarter q is source channel 16jg + 4q + e.  The weights are pre-packed by spx_pack_weight in exactly that order,
// lane-linear, so a B fragment is one coalesced 1 KiB read.
//
// Serves reference call sites pcdet/models/backbones_3d/spconv_backbone.py:86,93,98-100,105-107,
// 112-114,121 (forward) and their autograd (dgrad) — spconv itself is not vendored.
#include <stdlib.h>

#include <type_traits>

#include "spx_common.h"

#ifdef SPX_CV_DIAG
// diagnostic build only (never shipped): per-wave cycle stamps, see tools/conv_diag.py
__device__ unsigned long long* g_cv_diag = nullptr;
extern "C" int spx_diag_set_conv(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cv_diag), &p, sizeof(p)); }
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline bool mfma_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128; }
static inline bool use_mfma(int c_src, int c_dst) { return mfma_ok(c_src) && mfma_ok(c_dst); }

// ---------------------------------------------------------------- weight packing
// mode 0 (forward): src channel = ci, dst channel = co ; mode 1 (dgrad): src = co, dst = ci
__device__ __forceinline__ float w_elem(const float* w, int cout, int K, int cin, int mode, int k, int cs, int cd) {
  return mode == 0 ? w[((size_t)cd * K + k) * cin + cs] : w[((size_t)cs * K + k) * cin + cd];
}

// MFMA order: [k][nt][jg][lane][e]; lane = 16q + c ; k-step j = 4jg + e ; cs = 16jg + 4q + e ; cd = 16nt + c
// mode 2: both orders in one launch, out[0, total) = forward operand, out[total, 2 total) = dgrad operand
__global__ void k_pack_mfma(const float* __restrict__ w, int cout, int K, int cin, int mode, int CS, int CD,
                            float* __restrict__ out) {
  int total = K * CS * CD;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (mode == 2) {
    if (t >= 2 * total) return;
    out += t >= total ? total : 0;
    mode = t >= total ? 1 : 0;
    t -= mode * total;
    CS = mode == 0 ? cin : cout;
    CD = mode == 0 ? cout : cin;
  }
  if (t >= total) return;
  int e = t & 3, lane = (t >> 2) & 63;
  int rest = t >> 8;
  int JG = CS / 16, NT = CD / 16;
  int jg = rest % JG;
  rest /= JG;
  int nt = rest % NT;
  int k = rest / NT;
  int q = lane >> 4, c = lane & 15;
  int cs = 16 * jg + 4 * q + e, cd = 16 * nt + c;
  out[t] = w_elem(w, cout, K, cin, mode, k, cs, cd);
}

// plain order [k][cs][cd]
__global__ void k_pack_plain(const float* __restrict__ w, int cout, int K, int cin, int mode, int CS, int CD,
                             float* __restrict__ out) {
  int total = K * CS * CD;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (mode == 2) {
    if (t >= 2 * total) return;
    out += t >= total ? total : 0;
    mode = t >= total ? 1 : 0;
    t -= mode * total;
    CS = mode == 0 ? cin : cout;
    CD = mode == 0 ? cout : cin;
  }
  if (t >= total) return;
  int cd = t % CD, cs = (t / CD) % CS, k = t / (CD * CS);
  out[t] = w_elem(w, cout, K, cin, mode, k, cs, cd);
}

// ---------------------------------------------------------------- MFMA implicit GEMM
template <int CS, int CD, int MT, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void k_conv_mfma(const float* __restrict__ src, const float* __restrict__ wp,
                                                   const int32_t* __restrict__ pair, int64_t ld, int K, int flip,
                                                   int64_t n, const int64_t* d_n, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, int relu,
                                                   float* __restrict__ dst) {
  constexpr int NT = CD / 16;
  constexpr int JG = CS / 16;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t row_base = ((int64_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * (16 * MT);
  if (row_base >= nlive) return;
#ifdef SPX_CV_DIAG
  const unsigned long long d_t0 = __builtin_amdgcn_s_memtime(), d_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long d_units = 0;
#endif

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);

  // NOTE (measured, profiles/r01_conv_experiments.md): explicit software pipelining of this loop (ids two offsets ahead,
  // gathered rows one ahead) was 20 % SLOWER on MI355X at KITTI sizes — the extra live registers cost more than the
  // hidden latency; 4-5 resident waves per SIMD already overlap the pair -> row -> MFMA chain.
  for (int k = 0; k < K; ++k) {
    const int32_t* prow = pair + (int64_t)(flip ? K - 1 - k : k) * ld;
    int32_t id[MT];
    bool mv[MT];
    bool any = false;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      int64_t row = row_base + 16 * m + r;
      id[m] = row < nlive ? prow[row] : -1;
      mv[m] = __ballot(id[m] >= 0) != 0ull;
      any |= mv[m];
    }
    if (!any) continue;  // wave-uniform
#ifdef SPX_CV_DIAG
    d_units += 1;
#endif
#pragma unroll
    for (int jg = 0; jg < JG; ++jg) {
      f32x4 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = wp4[((size_t)(k * NT + nt) * JG + jg) * 64 + lane];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (!mv[m]) continue;  // wave-uniform
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        if (id[m] >= 0)
          a = *reinterpret_cast<const f32x4*>(src + (size_t)id[m] * CS + 16 * jg + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[nt][e], acc[m][nt], 0, 0, 0);
      }
    }
  }

#ifdef SPX_CV_DIAG
  if (g_cv_diag && lane == 0) {
    unsigned long long* o = g_cv_diag + ((size_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * 4;
    o[0] = __builtin_amdgcn_s_memtime() - d_t0;        // wave lifetime before the epilogue, core cycles
    o[1] = d_units;                                    // (tile, offset) units executed
    o[2] = d_r0;                                       // start, 100 MHz ticks (global clock)
    o[3] = __builtin_amdgcn_s_memrealtime();           // end
  }
#endif
  // epilogue: C layout col = lane&15, row = 4*(lane>>4) + e ; optional y = acc*scale + shift, ReLU
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = 16 * nt + r;
    const float sc = scale ? scale[col] : 1.0f;
    const float sh = shift ? shift[col] : 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int64_t row = row_base + 16 * m + 4 * q + e;
        if (row < nlive) {
          float v = acc[m][nt][e];
          if (scale || shift) v = v * sc + sh;
          if (relu) v = v > 0.f ? v : 0.f;
          dst[row * CD + col] = v;
        }
      }
  }
}

// ---------------------------------------------------------------- MFMA implicit GEMM, low-channel layers (latency form)
// For c_src * c_dst <= 2048 a (tile, offset) unit is 4-32 MFMAs, and k_conv_mfma spends its time waiting: rule entry ->
// (weights + rows -> MFMA) per 16-channel group, each step a full memory round trip (the compiler keeps `load, wait
// vmcnt(0), multiply` per group, and the per-offset `continue` is a branch it cannot count waits across).  Here
//   phase 1: the wave reads all K rule entries of its 16 rows up front (K independent loads, no branch), keeps them in
//            LDS and derives the set of non-empty offsets by ballot;
//   phase 2: it walks only the non-empty offsets; the rows of the NEXT offset are gathered (unconditionally, index
//            clamped, masked when used) and the weight fragments of the current one requested before a scheduling fence,
//            so a unit costs one partially hidden round trip instead of up to five.
// Same summation order as k_conv_mfma (offsets ascending, channels in packed order): identical bits.
template <int CS, int CD>
__global__ __launch_bounds__(256) void k_conv_mfma_sm(const float* __restrict__ src, const float* __restrict__ wp,
                                                      const int32_t* __restrict__ pair, int64_t ld, int K, int flip,
                                                      int64_t n, const int64_t* d_n, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu,
                                                      float* __restrict__ dst) {
  constexpr int NT = CD / 16;
  constexpr int JG = CS / 16;
  __shared__ int32_t s_id[4][SPX_MAX_KVOL][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t row_base = ((int64_t)blockIdx.x * 4 + wave) * 16;
  if (row_base >= nlive) return;            // no barrier is used: waves are independent
  const int64_t row = row_base + r;
  const int64_t rowc = row < nlive ? row : nlive - 1;
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);

  // phase 1: rule entries of this tile for every offset (loop index k <-> table row K-1-k when flipped)
  uint32_t todo = 0;
  for (int k0 = 0; k0 < K; k0 += 9) {
    int32_t id[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int k = k0 + u < K ? k0 + u : K - 1;
      id[u] = pair[(int64_t)(flip ? K - 1 - k : k) * ld + rowc];
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int32_t v = (k0 + u < K && row < nlive) ? id[u] : -1;
      if (__ballot(v >= 0) != 0ull) todo |= 1u << (k0 + u);
      if (q == 0 && k0 + u < K) s_id[wave][k0 + u][r] = v;
    }
  }
  __builtin_amdgcn_wave_barrier();

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto gather = [&](int32_t id, f32x4 (&a)[JG]) {
    const float* p = src + (size_t)(id > 0 ? id : 0) * CS + 4 * q;
#pragma unroll
    for (int jg = 0; jg < JG; ++jg) a[jg] = *reinterpret_cast<const f32x4*>(p + 16 * jg);
  };
  auto pop = [&]() -> int {
    if (todo == 0) return -1;
    const int k = __ffs((int)todo) - 1;
    todo &= todo - 1;
    return k;
  };

  // phase 2
  int k_cur = pop(), k_nxt = pop();
  int32_t id_cur = k_cur >= 0 ? s_id[wave][k_cur][r] : -1;
  f32x4 a_cur[JG], a_nxt[JG];
  gather(id_cur, a_cur);
  while (k_cur >= 0) {
    f32x4 b[JG][NT];
#pragma unroll
    for (int jg = 0; jg < JG; ++jg)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[jg][nt] = wp4[((size_t)(k_cur * NT + nt) * JG + jg) * 64 + lane];
    const int32_t id_nxt = s_id[wave][k_nxt >= 0 ? k_nxt : 0][r];
    const int32_t idn = k_nxt >= 0 ? id_nxt : -1;
    gather(idn, a_nxt);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jg = 0; jg < JG; ++jg)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float av = id_cur >= 0 ? a_cur[jg][e] : 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[jg][nt][e], acc[nt], 0, 0, 0);
      }
#pragma unroll
    for (int jg = 0; jg < JG; ++jg) a_cur[jg] = a_nxt[jg];
    id_cur = idn;
    k_cur = k_nxt;
    k_nxt = pop();
  }

  // epilogue: C layout col = lane&15, row = 4*(lane>>4) + e ; optional y = acc*scale + shift, ReLU
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
}

// ---------------------------------------------------------------- MFMA implicit GEMM, weights through LDS
// In-kernel stamps on k_conv_mfma (tools/conv_diag.py, 64->64, 82k rows): a wave spends ~13,000 cycles per (tile, offset)
// unit against 2,048 cycles of MFMA; the time goes into waiting for memory, and the traffic that congests it is the
// WEIGHT fragments: every wave re-reads the 16 KB slice W_k from L2 for every unit (1.9 GB per launch against 0.46 GB of
// gathered rows; the 20 waves of a CU sit at different offsets, so the 32 KB L1 holds none of it).
// Here the WPB waves of a workgroup walk the offsets in lockstep: W_k is copied global -> LDS once per workgroup with
// global_load_lds_dwordx4 (LDS-DMA: no VGPR staging; the packed weight slice is already the lane-linear LDS image),
// double-buffered so the copy of W_k+1 is in flight while W_k is multiplied, and every wave reads its B fragments with
// ds_read_b128.  L2 -> CU weight traffic drops WPB-fold; the source rows are still gathered straight into the A operand
// registers.  One barrier per offset.  A wave whose 16 rows have no neighbour at k skips its MFMAs but not the barrier.
// Same summation order per output element as k_conv_mfma (offsets ascending, channels in packed order): identical bits.
template <int CS, int CD, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_conv_mfma_ls(const float* __restrict__ src, const float* __restrict__ wp,
                                                           const int32_t* __restrict__ pair, int64_t ld, int K, int flip,
                                                           int64_t n, const int64_t* d_n, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu,
                                                           float* __restrict__ dst) {
  constexpr int NT = CD / 16;
  constexpr int JG = CS / 16;
  constexpr int NF = NT * JG;              // 1 KiB fragments per offset
  __shared__ f32x4 sB[2][NF * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t row_base = ((int64_t)blockIdx.x * WPB + wave) * 16;
  const int64_t row = row_base + r;
  const bool in_range = row < nlive;       // no early return: every wave takes part in the copies and the barriers

#ifdef SPX_CV_DIAG
  const unsigned long long d_t0 = __builtin_amdgcn_s_memtime(), d_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long d_units = 0;
#endif
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);
  auto copy_w = [&](int k, int buf) {
#pragma unroll
    for (int i = 0; i < (NF + WPB - 1) / WPB; ++i) {
      const int f = wave + i * WPB;       // wave-uniform
      if (f < NF)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(wp4 + ((size_t)k * NF + f) * 64 + lane),
            (__attribute__((address_space(3))) void*)(&sB[buf][f * 64]), 16, 0, 0);
    }
  };

  auto load_id = [&](int k) -> int32_t {
#ifdef SPX_CV_NO_IDS
    return (in_range && k < K) ? (int32_t)row : -1;   // diag: no rule-table read, identity gather
#endif
    return (in_range && k < K) ? pair[(int64_t)(flip ? K - 1 - k : k) * ld + row] : -1;
  };
  auto gather = [&](int32_t id, f32x4 (&a)[JG]) {
#pragma unroll
    for (int jg = 0; jg < JG; ++jg) {
      a[jg] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef SPX_CV_NO_GATHER
      if (id >= 0) a[jg] = f32x4{1.f, 2.f, 3.f, (float)id};   // diag: no row read
#else
      if (id >= 0) a[jg] = *reinterpret_cast<const f32x4*>(src + (size_t)id * CS + 16 * jg + 4 * q);
#endif
    }
  };
  // Pipeline: while offset k is multiplied, the rows of offset k+1, the weight slice W_k+1 and the rule entries of
  // offset k+2 are in flight; the barrier at the top of each trip is the one point where they are waited for.
  int32_t id_cur = load_id(0);
  int32_t id_nxt = load_id(1);
  copy_w(0, 0);
  f32x4 a_cur[JG], a_nxt[JG];
  gather(id_cur, a_cur);
#ifdef SPX_CV_DIAG
  unsigned long long d_drain = 0, d_bar = 0, d_mma = 0;
#endif
  for (int k = 0; k < K; ++k) {
#ifdef SPX_CV_DIAG
    unsigned long long d_a = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long d_b = __builtin_amdgcn_s_memtime();
    d_drain += d_b - d_a;
#endif
    __syncthreads();   // W_k landed for every wave, a_cur / id_nxt arrived; buffer (k+1)&1 is free again
#ifdef SPX_CV_DIAG
    d_bar += __builtin_amdgcn_s_memtime() - d_b;
#endif
    const bool any = __ballot(id_cur >= 0) != 0ull;   // wave-uniform
    const int32_t id_nn = load_id(k + 2);
    gather(id_nxt, a_nxt);
#ifndef SPX_CV_NO_GLDS
    if (k + 1 < K) copy_w(k + 1, (k + 1) & 1);
#endif
    // everything above is only ISSUED here; keep it above the MFMAs (the scheduler would otherwise sink the gathers
    // next to their use in the next trip and serialise latency and arithmetic again)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#ifdef SPX_CV_DIAG
    unsigned long long d_c = __builtin_amdgcn_s_memtime();
#endif
    if (any) {
#ifdef SPX_CV_DIAG
      d_units += 1;
#endif
      const f32x4* B = sB[k & 1];
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) {
        f32x4 b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = B[(nt * JG + jg) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[jg][e], b[nt][e], acc[nt], 0, 0, 0);
      }
    }
#ifdef SPX_CV_DIAG
    d_mma += __builtin_amdgcn_s_memtime() - d_c;
#endif
#pragma unroll
    for (int jg = 0; jg < JG; ++jg) a_cur[jg] = a_nxt[jg];
    id_cur = id_nxt;
    id_nxt = id_nn;
  }

  if (row_base >= nlive) return;
#ifdef SPX_CV_DIAG
  if (g_cv_diag && lane == 0) {
    unsigned long long* o = g_cv_diag + ((size_t)blockIdx.x * WPB + wave) * 4;
    o[0] = __builtin_amdgcn_s_memtime() - d_t0;
    o[1] = d_units | ((d_drain >> 6) << 8) | ((d_bar >> 6) << 28) | ((d_mma >> 6) << 48);   // 64-cycle units, 20 bits each
    o[2] = d_r0;
    o[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = 16 * nt + r;
    const float sc = scale ? scale[col] : 1.0f;
    const float sh = shift ? shift[col] : 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int64_t orow = row_base + 4 * q + e;
      if (orow < nlive) {
        float v = acc[nt][e];
        if (scale || shift) v = v * sc + sh;
        if (relu) v = v > 0.f ? v : 0.f;
        dst[orow * CD + col] = v;
      }
    }
  }
}

// ---------------------------------------------------------------- MFMA implicit GEMM, pair-compacting variant
// For the MFMA-bound layers (c_dst >= 64) roughly half of a submanifold rulebook is -1, so the masked kernel above
// spends half its MFMAs on zero rows.  Here a wave owns R = 64 destination rows but only 16*NTW destination COLUMNS
// (the waves of a workgroup split the columns of the same rows, which restores the wave count that the larger row tile
// costs).  For every offset k the wave ballot-compacts its valid (source row, local dst row) pairs into a private LDS
// list, runs the MFMAs over 16-pair chunks of that list only, and adds each chunk's [16 x 16*NTW] result into a
// private LDS accumulator tile at the pairs' destination rows with ds_add_f32.  For one k every destination row has at
// most one pair and the wave walks k in order, so the summation order per output element is fixed: results are
// bitwise reproducible, no cross-wave traffic, no barrier.
template <int CS, int CD, int NTW>
__global__ __launch_bounds__(256) void k_conv_mfma_cp(const float* __restrict__ src, const float* __restrict__ wp,
                                                      const int32_t* __restrict__ pair, int64_t ld, int K, int flip,
                                                      int64_t n, const int64_t* d_n, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu,
                                                      float* __restrict__ dst) {
  constexpr int R = 64;                    // rows per wave
  constexpr int NT = CD / 16;
  constexpr int JG = CS / 16;
  constexpr int WPT = NT / NTW;            // waves per row tile (column split)
  constexpr int TPB = 4 / WPT;             // row tiles per 4-wave block
  constexpr int CW = 16 * NTW;             // columns per wave
  constexpr int LD = CW + 4;               // LDS row pitch (floats), keeps float4 alignment, spreads banks
  static_assert(NT % NTW == 0 && 4 % WPT == 0 && WPT <= 4, "bad column split");
  __shared__ float s_acc[4][R * LD];
  __shared__ int32_t s_src[4][R];
  __shared__ int32_t s_dst[4][R];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t row0 = ((int64_t)blockIdx.x * TPB + wave / WPT) * R;
  if (row0 >= nlive) return;               // whole wave; no barriers are used anywhere
  const int nt0 = (wave % WPT) * NTW;      // first 16-column tile of this wave
  float* acc = s_acc[wave];
  int32_t* lsrc = s_src[wave];
  int32_t* ldst = s_dst[wave];

  for (int i = lane; i < R * LD / 4; i += 64) reinterpret_cast<f32x4*>(acc)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);
  const int64_t myrow = row0 + lane;
  int32_t id_next = myrow < nlive ? pair[(int64_t)(flip ? K - 1 : 0) * ld + myrow] : -1;

  for (int k = 0; k < K; ++k) {
    const int32_t id = id_next;
    if (k + 1 < K) id_next = myrow < nlive ? pair[(int64_t)(flip ? K - 2 - k : k + 1) * ld + myrow] : -1;
    const unsigned long long mask = __ballot(id >= 0);
    if (mask == 0ull) continue;            // wave-uniform
    const int nvalid = __popcll(mask);
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    if (id >= 0) {
      lsrc[rank] = id;
      ldst[rank] = lane;
    }
    f32x4 b[NTW][JG];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) b[t][jg] = wp4[((size_t)(k * NT + nt0 + t) * JG + jg) * 64 + lane];
    __builtin_amdgcn_wave_barrier();

    const int nch = (nvalid + 15) >> 4;
    // software pipeline over chunks: the gather of chunk ch+1 is in flight while chunk ch runs its MFMAs
    f32x4 a_nx[JG];
    {
      const bool has = r < nvalid;
      const int32_t sid = has ? lsrc[r] : 0;
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) {
        a_nx[jg] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has) a_nx[jg] = *reinterpret_cast<const f32x4*>(src + (size_t)sid * CS + 16 * jg + 4 * q);
      }
    }
    for (int ch = 0; ch < nch; ++ch) {
      f32x4 a[JG];
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) a[jg] = a_nx[jg];
      if (ch + 1 < nch) {
        const int p = (ch + 1) * 16 + r;
        const bool has = p < nvalid;
        const int32_t sid = has ? lsrc[p] : 0;
#pragma unroll
        for (int jg = 0; jg < JG; ++jg) {
          a_nx[jg] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (has) a_nx[jg] = *reinterpret_cast<const f32x4*>(src + (size_t)sid * CS + 16 * jg + 4 * q);
        }
      }
      // two independent accumulation chains per column tile (MFMA dependent-issue latency 40 > issue 32 cycles)
      f32x4 c0[NTW], c1[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) c0[t] = c1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jg = 0; jg < JG; ++jg)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            if ((e & 1) == 0)
              c0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jg][e], b[t][jg][e], c0[t], 0, 0, 0);
            else
              c1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jg][e], b[t][jg][e], c1[t], 0, 0, 0);
          }
      // C layout: this lane holds rows i = 4q+e (pairs ch*16 + i), column r of each 16-wide tile.  Plain LDS
      // read-modify-write (NOT ds_add_f32: LDS float atomics serialise per lane, ~70 cycles per instruction): for one
      // offset k every destination row occurs at most once, and only this wave touches this tile.
      const int pbase = ch * 16 + 4 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (pbase + e < nvalid) {
          const int dl = ldst[pbase + e];
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[dl * LD + 16 * t + r] += c0[t][e] + c1[t][e];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // epilogue: 4 lanes per row, one float4 each per 16 columns -> 64 B contiguous per row per instruction
  __builtin_amdgcn_wave_barrier();
  const int col4 = (lane & 3) * 4;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int colg = 16 * (nt0 + t) + col4;
    f32x4 sc = f32x4{1.f, 1.f, 1.f, 1.f}, sh = f32x4{0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x4*>(scale + colg);
    if (shift) sh = *reinterpret_cast<const f32x4*>(shift + colg);
    for (int rr = lane >> 2; rr < R; rr += 16) {
      const int64_t row = row0 + rr;
      if (row >= nlive) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(acc + rr * LD + 16 * t + col4);
      if (scale || shift) v = v * sc + sh;
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(dst + row * CD + colg) = v;
    }
  }
}

// ---------------------------------------------------------------- MFMA implicit GEMM, block-cooperative variant
// A block owns R = 64 destination rows; its WPT waves split the destination COLUMNS (16*NTW each).  Per kernel offset k:
//   * wave 0 ballot-compacts the block's valid (source row, local dst row) pairs into an LDS list (two offsets ahead);
//   * ALL threads gather the listed source rows — 16 consecutive threads read one 256-B row, fully coalesced — into
//     registers one offset ahead and drop them into a shared LDS tile after the barrier (the gather of offset k+1 is in
//     flight while offset k runs its MFMAs);
//   * every wave reads its A fragments from that tile (ds_read_b128), multiplies with ITS OWN column slice of W_k
//     (register double-buffered, loaded one offset ahead) over 16-pair chunks of the list only — no MFMA is spent on
//     rows without a neighbour — and adds the [16 x 16*NTW] result into its private LDS accumulator at the pairs'
//     destination rows (plain read-modify-write: one pair per destination row per offset, one owner wave per column
//     slice, offsets in order => fixed summation order, bitwise reproducible).
// Two barriers per offset.  LDS: (CS+4)*64*4 + WPT*64*(16*NTW+4)*4 + lists  (39 KiB for 64->64: 4 blocks per CU).
template <int CS, int CD, int NTW, int R>
__global__ __launch_bounds__(64 * (CD / 16 / NTW)) void k_conv_mfma_bc(
    const float* __restrict__ src, const float* __restrict__ wp, const int32_t* __restrict__ pair, int64_t ld, int K,
    int flip, int64_t n, const int64_t* d_n, const float* __restrict__ scale, const float* __restrict__ shift, int relu,
    float* __restrict__ dst) {
  constexpr int NT = CD / 16, JG = CS / 16;
  constexpr int WPT = NT / NTW, NTHR = 64 * WPT;
  constexpr int CW = 16 * NTW, LDC = CW + 4, LDA = CS + 4;
  constexpr int PPR = CS / 4;                        // float4 pieces per source row
  constexpr int NPIECE = (R * PPR + NTHR - 1) / NTHR;  // pieces per thread when all 64 rows are valid
  static_assert(NT % NTW == 0 && WPT >= 1 && WPT <= 4, "bad column split");
  constexpr int LA = 3;                              // gather look-ahead (offsets in flight per block)
  constexpr int NS = LA + 2;                         // list slots: k (in use) .. k+LA (issued) and k+LA+1 (being built)
  __shared__ float s_a[R * LDA];
  __shared__ float s_acc[WPT][R * LDC];
  __shared__ int32_t s_src[NS][R];
  __shared__ int32_t s_dst[NS][R];
  __shared__ int32_t s_cnt[NS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t row0 = (int64_t)blockIdx.x * R;
  if (row0 >= nlive) return;   // uniform for the whole block
  const int nt0 = wave * NTW;
  float* acc = s_acc[wave];
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);
  const f32x4* src4 = reinterpret_cast<const f32x4*>(src);

  for (int i = lane; i < R * LDC / 4; i += 64) reinterpret_cast<f32x4*>(acc)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- rule entries and pair lists (wave 0 only; lane <-> destination row)
  const int64_t myrow = row0 + lane;
  auto load_id = [&](int k) -> int32_t {
    return (k < K && lane < R && myrow < nlive) ? pair[(int64_t)(flip ? K - 1 - k : k) * ld + myrow] : -1;
  };
  auto build_list = [&](int32_t id, int slot) {
    const unsigned long long mask = __ballot(id >= 0);
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    if (id >= 0) {
      s_src[slot][rank] = id;
      s_dst[slot][rank] = lane;
    }
    if (lane == 0) s_cnt[slot] = __popcll(mask);
  };
  int32_t id_a = -1, id_b = -1;   // entries of offsets k+LA+1 and k+LA+2 while offset k computes
  if (wave == 0) {
#pragma unroll
    for (int m = 0; m <= LA; ++m) build_list(load_id(m), m);
    id_a = load_id(LA + 1);
    id_b = load_id(LA + 2);
  }
  __syncthreads();

  // ---- gather: piece p of a list = (row p / PPR, float4 part p % PPR); 16 consecutive threads read one source row
  f32x4 g[LA][NPIECE];
  auto gather_issue = [&](int m, f32x4* gs) {        // rows of offset m -> registers
    const int slot = m % NS;
    const int cnt = m < K ? s_cnt[slot] : 0;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int p = tid + i * NTHR;
      gs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p < cnt * PPR) gs[i] = src4[(size_t)s_src[slot][p / PPR] * PPR + (p % PPR)];
    }
  };
  auto gather_commit = [&](int m, const f32x4* gs) { // registers -> shared tile
    const int cnt = m < K ? s_cnt[m % NS] : 0;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int p = tid + i * NTHR;
      if (p < cnt * PPR) *reinterpret_cast<f32x4*>(&s_a[(p / PPR) * LDA + (p % PPR) * 4]) = gs[i];
    }
  };
  auto load_b = [&](int k, f32x4 (*b)[JG]) {
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int jg = 0; jg < JG; ++jg)
        b[t][jg] = k < K ? wp4[((size_t)(k * NT + nt0 + t) * JG + jg) * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
  };

  f32x4 b_cur[NTW][JG], b_nx[NTW][JG];
  gather_issue(0, g[0]);
  gather_issue(1, g[1]);
  gather_issue(2, g[2]);
  load_b(0, b_cur);
  gather_commit(0, g[0]);
  __syncthreads();

  // one phase = one kernel offset.  J = k % LA selects the register set statically (the loop is unrolled by LA).
  auto phase = [&](int k, auto jc) {
    constexpr int J = decltype(jc)::value;
    const int slot = k % NS;
    const int n_k = s_cnt[slot];
    gather_issue(k + LA, g[J]);          // set J held offset k (already in the tile): reuse it for offset k+LA
    load_b(k + 1, b_nx);

    const int nch = (n_k + 15) >> 4;
    for (int ch = 0; ch < nch; ++ch) {
      const int p = ch * 16 + r;
      f32x4 a[JG];
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) {
        a[jg] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p < n_k) a[jg] = *reinterpret_cast<const f32x4*>(&s_a[p * LDA + 16 * jg + 4 * q]);
      }
      f32x4 c0[NTW], c1[NTW];   // two accumulation chains per tile (dependent-issue latency 40 > issue interval 32)
#pragma unroll
      for (int t = 0; t < NTW; ++t) c0[t] = c1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jg = 0; jg < JG; ++jg)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            if ((e & 1) == 0)
              c0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jg][e], b_cur[t][jg][e], c0[t], 0, 0, 0);
            else
              c1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jg][e], b_cur[t][jg][e], c1[t], 0, 0, 0);
          }
      const int pbase = ch * 16 + 4 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (pbase + e < n_k) {
          const int dl = s_dst[slot][pbase + e];
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[dl * LDC + 16 * t + r] += c0[t][e] + c1[t][e];
        }
      }
    }
    if (wave == 0) {             // list of offset k+LA+1 -> the slot offset k-1 used (free since the last barrier)
      if (k + LA + 1 < K) build_list(id_a, (k + LA + 1) % NS);
      id_a = id_b;
      id_b = load_id(k + LA + 3);
    }
    __syncthreads();             // everyone is done reading the tile / s_dst[slot]
    gather_commit(k + 1, g[(J + 1) % LA]);
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) b_cur[t][jg] = b_nx[t][jg];
    __syncthreads();             // tile of offset k+1 visible
  };
  for (int k = 0; k < K; k += LA) {
    phase(k, std::integral_constant<int, 0>{});
    if (k + 1 < K) phase(k + 1, std::integral_constant<int, 1>{});
    if (k + 2 < K) phase(k + 2, std::integral_constant<int, 2>{});
  }

  // epilogue: 4 lanes per row, one float4 each per 16 columns
  const int col4 = (lane & 3) * 4;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int colg = 16 * (nt0 + t) + col4;
    f32x4 sc = f32x4{1.f, 1.f, 1.f, 1.f}, sh = f32x4{0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x4*>(scale + colg);
    if (shift) sh = *reinterpret_cast<const f32x4*>(shift + colg);
    for (int rr = lane >> 2; rr < R; rr += 16) {
      const int64_t row = row0 + rr;
      if (row >= nlive) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(acc + rr * LDC + 16 * t + col4);
      if (scale || shift) v = v * sc + sh;
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(dst + row * CD + colg) = v;
    }
  }
}

// ---------------------------------------------------------------- VALU fallback (any channel counts)
// thread = (row, cd); weights in plain [k][cs][cd] order.  Used for conv_input (c_src = 4 or 5) and
// for channel counts the MFMA kernels do not tile.
__global__ void k_conv_valu(const float* __restrict__ src, int CS, const float* __restrict__ wp, int CD,
                            const int32_t* __restrict__ pair, int64_t ld, int K, int flip, int64_t n,
                            const int64_t* d_n, const float* __restrict__ scale, const float* __restrict__ shift,
                            int relu, float* __restrict__ dst) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = t / CD;
  int cd = (int)(t % CD);
  if (row >= spx_live_n(d_n, n)) return;
  float acc = 0.f;
  // rule entries nine at a time, unconditionally: one memory round trip per nine offsets instead of one per offset
  // (a load behind the `continue` of the previous offset is a dependent load: 27 latencies in a row, 37 us for conv_input)
  for (int k0 = 0; k0 < K; k0 += 9) {
    int32_t id[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int k = k0 + u < K ? k0 + u : K - 1;
      id[u] = pair[(int64_t)(flip ? K - 1 - k : k) * ld + row];
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int k = k0 + u;
      if (k >= K || id[u] < 0) continue;
      const float* x = src + (size_t)id[u] * CS;
      const float* ww = wp + (size_t)k * CS * CD + cd;
      for (int cs = 0; cs < CS; ++cs) acc = fmaf(x[cs], ww[(size_t)cs * CD], acc);
    }
  }
  if (scale || shift) acc = acc * (scale ? scale[cd] : 1.f) + (shift ? shift[cd] : 0.f);
  if (relu) acc = acc > 0.f ? acc : 0.f;
  dst[row * CD + cd] = acc;
}

template <int CS, int CD>
static int launch_mfma(const float* src, const float* wp, const int32_t* pair, int64_t ld, int K, int flip, int64_t n,
                       const int64_t* d_n, const float* scale, const float* shift, int relu, float* dst,
                       hipStream_t s) {
  // rows per wave = 16*MT: larger MT amortises the weight fragment over more rows, smaller MT gives
  // more waves.  Keep >= ~8 waves per CU (2048 waves) when the problem allows it.
  {
    // The pair-compacting kernel wins only on low-density strided layers (measured); it is kept selectable for
    // experiments (SPX_CONV_CP=1) and is NOT the default.
    int use_bc = 0;
    if (const char* e = getenv("SPX_CONV_BC")) use_bc = atoi(e);  // dev override
    if (use_bc && CD >= 32) {
      constexpr int NTW = CD >= 128 ? 2 : 1;
      constexpr int WPT = (CD / 16) / NTW;
      // rows per block: 64 compacts better (fewer padded MFMA rows), 32 gives twice the blocks and half the LDS.  Pick
      // the one with the smaller  (dispatch rounds) x (rows per block) / (compaction efficiency)  on this chip.
      auto lds_bytes = [&](int R) { return 4 * (R * (CS + 4) + WPT * R * (16 * NTW + 4)) + 5 * 2 * 4 * R + 64; };
      auto cost = [&](int R, double eff) {
        int per_cu = 160 * 1024 / lds_bytes(R);
        int by_waves = 32 / WPT;
        if (per_cu > by_waves) per_cu = by_waves;
        if (per_cu < 1) per_cu = 1;
        int64_t tiles = (n + R - 1) / R, slots = 256 * (int64_t)per_cu;
        return (double)((tiles + slots - 1) / slots) * R / eff;
      };
      int R = cost(32, 0.70) < cost(64, 0.80) ? 32 : 64;
      if (const char* e = getenv("SPX_CONV_BC_R")) R = atoi(e);
      if (R == 32)
        hipLaunchKernelGGL((k_conv_mfma_bc<CS, CD, NTW, 32>), dim3((unsigned)((n + 31) / 32)), dim3(64 * WPT), 0, s, src, wp,
                           pair, ld, K, flip, n, d_n, scale, shift, relu, dst);
      else
        hipLaunchKernelGGL((k_conv_mfma_bc<CS, CD, NTW, 64>), dim3((unsigned)((n + 63) / 64)), dim3(64 * WPT), 0, s, src, wp,
                           pair, ld, K, flip, n, d_n, scale, shift, relu, dst);
      return SPX_OK;
    }
    int use_ls = 0;
    if (const char* e = getenv("SPX_CONV_LS")) use_ls = atoi(e);  // dev override: weights through LDS, WPB = value
    if (use_ls && (CS / 16) * (CD / 16) <= 16) {                    // weight slice <= 16 KB: two LDS buffers, 5 waves per SIMD
      int64_t waves1 = (n + 15) / 16;
      if (use_ls == 5)
        hipLaunchKernelGGL((k_conv_mfma_ls<CS, CD, 5>), dim3((unsigned)((waves1 + 4) / 5)), dim3(320), 0, s, src, wp, pair, ld,
                           K, flip, n, d_n, scale, shift, relu, dst);
      else if (use_ls == 10)
        hipLaunchKernelGGL((k_conv_mfma_ls<CS, CD, 10>), dim3((unsigned)((waves1 + 9) / 10)), dim3(640), 0, s, src, wp, pair, ld,
                           K, flip, n, d_n, scale, shift, relu, dst);
      else if (use_ls == 8)
        hipLaunchKernelGGL((k_conv_mfma_ls<CS, CD, 8>), dim3((unsigned)((waves1 + 7) / 8)), dim3(512), 0, s, src, wp, pair, ld,
                           K, flip, n, d_n, scale, shift, relu, dst);
      else if (use_ls == 16)
        hipLaunchKernelGGL((k_conv_mfma_ls<CS, CD, 16>), dim3((unsigned)((waves1 + 15) / 16)), dim3(1024), 0, s, src, wp, pair,
                           ld, K, flip, n, d_n, scale, shift, relu, dst);
      else
        hipLaunchKernelGGL((k_conv_mfma_ls<CS, CD, 4>), dim3((unsigned)((waves1 + 3) / 4)), dim3(256), 0, s, src, wp, pair, ld,
                           K, flip, n, d_n, scale, shift, relu, dst);
      return SPX_OK;
    }
    int use_cp = 0;
    if (const char* e = getenv("SPX_CONV_CP")) use_cp = atoi(e);  // dev override
    if (use_cp) {
      constexpr int NTW = CD >= 128 ? 2 : 1;      // columns per wave: 16 (32 for 128-wide outputs)
      constexpr int WPT = (CD / 16) / NTW;        // waves sharing a 64-row tile
      constexpr int TPB = 4 / WPT;
      int64_t tiles = (n + 63) / 64;
      unsigned nb = (unsigned)((tiles + TPB - 1) / TPB);
      hipLaunchKernelGGL((k_conv_mfma_cp<CS, CD, NTW>), dim3(nb), dim3(256), 0, s, src, wp, pair, ld, K, flip, n, d_n,
                         scale, shift, relu, dst);
      return SPX_OK;
    }
  }
  {
    int use_sm = 1;
    if (const char* e = getenv("SPX_CONV_SM")) use_sm = atoi(e);    // dev override: 0 = one-load-per-group kernel
    if constexpr (CS * CD <= 2048) {
      if (use_sm && n < (int64_t(1) << 18)) {
        int64_t waves1 = (n + 15) / 16;
        hipLaunchKernelGGL((k_conv_mfma_sm<CS, CD>), dim3((unsigned)((waves1 + 3) / 4)), dim3(256), 0, s, src, wp, pair, ld, K,
                           flip, n, d_n, scale, shift, relu, dst);
        return SPX_OK;
      }
    }
  }
  // rows per wave = 16*MT.  Measured on MI355X (tools/kbench.py): at KITTI/Waymo sizes (<= ~260k rows) the chip is
  // under-filled and 16 rows per wave (most waves) is fastest for every layer; only very large inputs amortise the
  // weight fragment over more rows.
  int64_t waves4 = (n + 63) / 64, waves2 = (n + 31) / 32;
  int mt = n >= (int64_t(1) << 20) ? 4 : (n >= (int64_t(1) << 18) ? 2 : 1);
  if (const char* e = getenv("SPX_CONV_MT")) mt = atoi(e);  // dev override
  if (mt == 4 && CD <= 64) {
    unsigned nb = (unsigned)((waves4 + 3) / 4);
    hipLaunchKernelGGL((k_conv_mfma<CS, CD, 4>), dim3(nb), dim3(256), 0, s, src, wp, pair, ld, K, flip, n, d_n, scale,
                       shift, relu, dst);
  } else if (mt >= 2) {
    unsigned nb = (unsigned)((waves2 + 3) / 4);
    hipLaunchKernelGGL((k_conv_mfma<CS, CD, 2>), dim3(nb), dim3(256), 0, s, src, wp, pair, ld, K, flip, n, d_n, scale,
                       shift, relu, dst);
  } else {
    int64_t waves1 = (n + 15) / 16;
    int lds = 0;
    if (const char* e = getenv("SPX_CONV_LDS")) lds = atoi(e);  // dev: cap residency -> dynamic block dispatch
    if (lds > 0) {
      hipLaunchKernelGGL((k_conv_mfma<CS, CD, 1, 1>), dim3((unsigned)waves1), dim3(64), lds, s, src, wp, pair, ld, K, flip,
                         n, d_n, scale, shift, relu, dst);
    } else {
      unsigned nb = (unsigned)((waves1 + 3) / 4);
      hipLaunchKernelGGL((k_conv_mfma<CS, CD, 1>), dim3(nb), dim3(256), 0, s, src, wp, pair, ld, K, flip, n, d_n, scale,
                         shift, relu, dst);
    }
  }
  return SPX_OK;
}

}  // namespace

extern "C" int spx_pack_weight(const float* w, int cout, int kvol, int cin, int mode, float* packed,
                               spx_stream_t stream) {
  if (!w || !packed || cout <= 0 || cin <= 0 || kvol <= 0 || kvol > SPX_MAX_KVOL || mode < 0 || mode > 2)
    return SPX_ERR_INVALID_ARG;
  int CS = mode == 1 ? cout : cin, CD = mode == 1 ? cin : cout;
  int total = kvol * CS * CD;
  unsigned nb = (unsigned)(((mode == 2 ? 2 : 1) * total + 255) / 256);
  if (use_mfma(CS, CD))
    hipLaunchKernelGGL(k_pack_mfma, dim3(nb), dim3(256), 0, spx_s(stream), w, cout, kvol, cin, mode, CS, CD, packed);
  else
    hipLaunchKernelGGL(k_pack_plain, dim3(nb), dim3(256), 0, spx_s(stream), w, cout, kvol, cin, mode, CS, CD, packed);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

#define SPX_MFMA_CASE(A, B)                                                                                    \
  if (c_src == A && c_dst == B) {                                                                              \
    launch_mfma<A, B>(src, w_packed, pair, pair_ld, kvol, flip_k, n_dst, d_n_dst, scale, shift, relu, dst, s); \
    SPX_CHECK_LAUNCH();                                                                                        \
    return SPX_OK;                                                                                             \
  }

extern "C" int spx_conv_gemm(const float* src, int c_src, const float* w_packed, int c_dst, int kvol, int flip_k,
                             const int32_t* pair, int64_t pair_ld, int64_t n_dst, const int64_t* d_n_dst,
                             const float* scale, const float* shift, int relu, float* dst, spx_stream_t stream) {
  if (!src || !w_packed || !pair || !dst || c_src <= 0 || c_dst <= 0 || kvol <= 0 || kvol > SPX_MAX_KVOL ||
      n_dst < 0 || pair_ld < n_dst)
    return SPX_ERR_INVALID_ARG;
  if (n_dst >= (int64_t(1) << 31)) return SPX_ERR_TOO_LARGE;
  if (n_dst == 0) return SPX_OK;
  hipStream_t s = spx_s(stream);
  if (use_mfma(c_src, c_dst)) {
    SPX_MFMA_CASE(16, 16)
    SPX_MFMA_CASE(16, 32)
    SPX_MFMA_CASE(32, 16)
    SPX_MFMA_CASE(32, 32)
    SPX_MFMA_CASE(32, 64)
    SPX_MFMA_CASE(64, 32)
    SPX_MFMA_CASE(64, 64)
    SPX_MFMA_CASE(64, 128)
    SPX_MFMA_CASE(128, 64)
    SPX_MFMA_CASE(128, 128)
    SPX_MFMA_CASE(16, 64)
    SPX_MFMA_CASE(64, 16)
    SPX_MFMA_CASE(32, 128)
    SPX_MFMA_CASE(128, 32)
    SPX_MFMA_CASE(16, 128)
    SPX_MFMA_CASE(128, 16)
    return SPX_ERR_UNSUPPORTED;  // unreachable: mfma_ok() admits exactly the 16 pairs above
  }
  int64_t total = n_dst * c_dst;
  unsigned nb = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(k_conv_valu, dim3(nb), dim3(256), 0, s, src, c_src, w_packed, c_dst, pair, pair_ld, kvol, flip_k,
                     n_dst, d_n_dst, scale, shift, relu, dst);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
