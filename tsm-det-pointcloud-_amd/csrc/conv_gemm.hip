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

// Every convolution weight of a network in ONE launch (both operand orders each): desc[i] = {w, packed, cout, K, cin,
// first block} as six int64; block b belongs to the last i with first block <= b.  The weights change once per optimizer
// step, all of them, so one launch replaces 2 x (number of layers) k_pack_mfma launches per step.
__global__ __launch_bounds__(256) void k_pack_batched(const int64_t* __restrict__ desc, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[6 * mid + 5] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int64_t* d = desc + 6 * lo;
  const float* w = reinterpret_cast<const float*>(d[0]);
  float* out = reinterpret_cast<float*>(d[1]);
  const int cout = (int)d[2], K = (int)d[3], cin = (int)d[4];
  const int total = K * cin * cout;
  int t = (int)(blockIdx.x - d[5]) * 256 + threadIdx.x;
  if (t >= 2 * total) return;
  const int mode = t >= total ? 1 : 0;
  out += mode * total;
  t -= mode * total;
  const int CS = mode == 0 ? cin : cout, CD = mode == 0 ? cout : cin;
  if (mfma_ok(CS) && mfma_ok(CD)) {
    const int e = t & 3, lane = (t >> 2) & 63;
    int rest = t >> 8;
    const int JG = CS / 16, NT = CD / 16;
    const int jg = rest % JG;
    rest /= JG;
    const int nt = rest % NT;
    const int k = rest / NT;
    const int q = lane >> 4, c = lane & 15;
    out[t] = w_elem(w, cout, K, cin, mode, k, 16 * jg + 4 * q + e, 16 * nt + c);
  } else {
    const int cd = t % CD, cs = (t / CD) % CS, k = t / (CD * CS);
    out[t] = w_elem(w, cout, K, cin, mode, k, cs, cd);
  }
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

// ---------------------------------------------------------------- VALU, few source channels (the input layer: c_src = 4 / 5)
// One thread per destination ROW holding all CD outputs in registers: the K rule entries and the few source floats of a
// row are read once (k_conv_valu reads them once per output CHANNEL: 16 x the loads for conv_input), the [K][CS][CD]
// weights sit in LDS and every lane reads the same word (broadcast).  Same summation order as k_conv_valu (offsets
// ascending, source channels ascending, one fmaf chain per output): identical bits.
template <int CD>
__global__ __launch_bounds__(256) void k_conv_valu_rows(const float* __restrict__ src, int CS, const float* __restrict__ wp,
                                                        const int32_t* __restrict__ pair, int64_t ld, int K, int flip,
                                                        int64_t n, const int64_t* d_n, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int relu, float* __restrict__ dst) {
  extern __shared__ float s_w[];                     // [K][CS][CD]
  for (int i = threadIdx.x; i < K * CS * CD; i += 256) s_w[i] = wp[i];
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= spx_live_n(d_n, n)) return;
  float acc[CD];
#pragma unroll
  for (int c = 0; c < CD; ++c) acc[c] = 0.f;
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
      const float* ww = s_w + (size_t)k * CS * CD;
      for (int cs = 0; cs < CS; ++cs) {
        const float xv = x[cs];
#pragma unroll
        for (int c = 0; c < CD; ++c) acc[c] = fmaf(xv, ww[cs * CD + c], acc[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CD; c += 4) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = acc[c + e];
      if (scale || shift) a = a * (scale ? scale[c + e] : 1.f) + (shift ? shift[c + e] : 0.f);
      if (relu) a = a > 0.f ? a : 0.f;
      v[e] = a;
    }
    *reinterpret_cast<f32x4*>(dst + row * CD + c) = v;
  }
}

template <int CS, int CD>
static int launch_mfma(const float* src, const float* wp, const int32_t* pair, int64_t ld, int K, int flip, int64_t n,
                       const int64_t* d_n, const float* scale, const float* shift, int relu, float* dst,
                       hipStream_t s) {
  // (Round 2 tried a kernel that keeps ALL K weight slices of a 16/32-channel layer in LDS — one 16-wave workgroup per CU,
  // tiles in equal contiguous shares, rows of four offsets gathered together: 58 us against 51 us for k_conv_mfma_sm at
  // 32->32 / 83 k rows; ablation: 22 us skeleton, +8 us gathers, +28..50 us LDS-read + MFMA phase at four waves per SIMD.
  // Not kept; profiles/r02_conv_experiments.md.)
  if constexpr (CS * CD <= 2048) {
    if (n < (int64_t(1) << 19)) {        // low-channel layers: the latency form (measured against two tiles per wave at
                                         // 140 k - 300 k live rows, also when the launch is sized by a 320 k-row static capacity)
      int64_t waves1 = (n + 15) / 16;
      hipLaunchKernelGGL((k_conv_mfma_sm<CS, CD>), dim3((unsigned)((waves1 + 3) / 4)), dim3(256), 0, s, src, wp, pair, ld, K,
                         flip, n, d_n, scale, shift, relu, dst);
      return SPX_OK;
    }
  }
  // rows per wave = 16*MT.  Measured on MI355X (tools/kbench.py): at KITTI/Waymo sizes (<= ~260k rows) the chip is
  // under-filled and 16 rows per wave (most waves) is fastest for every layer; only very large inputs amortise the
  // weight fragment over more rows.
  int64_t waves4 = (n + 63) / 64, waves2 = (n + 31) / 32, waves1 = (n + 15) / 16;
  const int mt = n >= (int64_t(1) << 20) ? 4 : (n >= (int64_t(1) << 18) ? 2 : 1);
  if (mt == 4 && CD <= 64) {
    hipLaunchKernelGGL((k_conv_mfma<CS, CD, 4>), dim3((unsigned)((waves4 + 3) / 4)), dim3(256), 0, s, src, wp, pair, ld, K,
                       flip, n, d_n, scale, shift, relu, dst);
  } else if (mt >= 2) {
    hipLaunchKernelGGL((k_conv_mfma<CS, CD, 2>), dim3((unsigned)((waves2 + 3) / 4)), dim3(256), 0, s, src, wp, pair, ld, K,
                       flip, n, d_n, scale, shift, relu, dst);
  } else {
    hipLaunchKernelGGL((k_conv_mfma<CS, CD, 1>), dim3((unsigned)((waves1 + 3) / 4)), dim3(256), 0, s, src, wp, pair, ld, K,
                       flip, n, d_n, scale, shift, relu, dst);
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

extern "C" int spx_pack_weight_batched(const int64_t* d_desc, int n, int64_t total_blocks, spx_stream_t stream) {
  if (!d_desc || n <= 0 || total_blocks <= 0 || total_blocks >= (int64_t(1) << 31)) return SPX_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_pack_batched, dim3((unsigned)total_blocks), dim3(256), 0, spx_s(stream), d_desc, n);
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
  if (c_src <= 8 && (c_dst == 16 || c_dst == 32) && (size_t)kvol * c_src * c_dst * sizeof(float) <= 48 * 1024) {
    const unsigned nbr = (unsigned)((n_dst + 255) / 256);
    const size_t lds = (size_t)kvol * c_src * c_dst * sizeof(float);
    if (c_dst == 16)
      hipLaunchKernelGGL((k_conv_valu_rows<16>), dim3(nbr), dim3(256), lds, s, src, c_src, w_packed, pair, pair_ld, kvol,
                         flip_k, n_dst, d_n_dst, scale, shift, relu, dst);
    else
      hipLaunchKernelGGL((k_conv_valu_rows<32>), dim3(nbr), dim3(256), lds, s, src, c_src, w_packed, pair, pair_ld, kvol,
                         flip_k, n_dst, d_n_dst, scale, shift, relu, dst);
    SPX_CHECK_LAUNCH();
    return SPX_OK;
  }
  int64_t total = n_dst * c_dst;
  unsigned nb = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(k_conv_valu, dim3(nb), dim3(256), 0, s, src, c_src, w_packed, c_dst, pair, pair_ld, kvol, flip_k,
                     n_dst, d_n_dst, scale, shift, relu, dst);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
