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
// One wave per block; each block owns a [cin, cout] accumulator for one offset k and one row range and writes it to
// a slot of a partial slab; a second kernel sums the slots of every k in fixed order (no float atomics: bitwise
// reproducible) and emits the reference parameter layout [Cout][K][Cin].
// Work split (the kernel is MFMA-bound on every SIMD that holds two waves, so its duration is the LONGEST block: in-
// kernel stamps showed max block lifetime = 2x the mean with equal row ranges, because the centre offset of a
// submanifold rulebook holds every row and a corner offset a few per cent).  A counting pass writes the valid-pair count
// of every (offset, 64-row group); the 8 XCDs take the 8 eighths of the rows (the K blocks that read the same dout /
// input rows share one L2); inside an XCD its J blocks are dealt to the offsets in proportion to the offsets' pair
// counts, and the blocks of one offset cut its rows at equal PAIR counts (prefix over the group counts).  Every block
// and the reduce kernel derive the same split from the counts with integer arithmetic: the summation order is still a
// pure function of the rulebook.
//
// Serves the autograd of reference call sites spconv_backbone.py:86-121, triggered by loss.backward()
// at tools/train_utils/train_utils.py:53.
#include "spx_common.h"

#ifdef SPX_WG_DIAG
// diagnostic build only (never shipped): per-block cycle stamps, see tools/wgrad_diag.py
__device__ unsigned long long* g_wg_diag = nullptr;
extern "C" int spx_diag_set(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_diag), &p, sizeof(p)); }
#define DIAG_T() __builtin_amdgcn_s_memtime()
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef SPX_SB
#define SPX_SB __builtin_amdgcn_sched_barrier(0)
#endif
constexpr int kGroups = 4;
#ifndef SPX_WG_BURST
#define SPX_WG_BURST 4
#endif
constexpr int kBurst = SPX_WG_BURST;  // K-steps whose loads are in flight together  // 64-row rule groups fetched per iteration (independent loads in flight)

// row-chunks per offset: enough one-wave blocks to fill the chip (S*K >= ~4 per SIMD), bounded by the partial-slab
// traffic (<= 32 MiB) and by at least one full iteration (64*kGroups rows) per block
static inline int wgrad_splits(int64_t n, int cin, int cout, int kvol) {
  int64_t s = n / (64 * kGroups);
  constexpr int slab_mb = 32;
  int64_t by_slab = (int64_t(slab_mb) << 20) / ((int64_t)kvol * cin * cout * 4);
  if (s > by_slab) s = by_slab;
  constexpr int max_blocks = 2048;
  int64_t by_blocks = max_blocks / kvol;          // two one-wave blocks per SIMD keep its MFMA pipe busy; more only lengthens the reduce
  if (by_blocks < 8) by_blocks = 8;
  if (s > by_blocks) s = by_blocks;
  if (s < 1) s = 1;
  return (int)s;
}

// Blocks of one XCD dealt to the K offsets: S_k = 1 + (J-K) * cnt[k] / sum(cnt)  (J >= K), evaluated on counts
// pre-shifted so that the product fits 32 bits (sum of floors <= floor of sum keeps the total within J).  Called by a
// full wave; lane k < K returns its S_k, the exclusive prefix start_k and its count.
// (Round 2 tried the min-max split instead — S_k = max(1, ceil(cnt[k] / T)), T the smallest pairs-per-block that fits J
// blocks, by bisection: every layer got 9-27 us SLOWER (profiles/r02_conv_experiments.md); so did deeper bursts and more,
// smaller blocks.  The proportional split stays.)
__device__ __forceinline__ void wgrad_plan(const int32_t* __restrict__ cnt, int K, int J, int lane, int& sk, int& start,
                                           int& mycnt) {
  const uint32_t cval = lane < K ? (uint32_t)cnt[lane] : 0u;
  mycnt = (int)cval;
  uint64_t total = cval;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
  int sh = 0;
  while ((total >> sh) >= (1u << 19)) ++sh;          // (J-K) < 2^12, so the product below stays under 2^31
  const uint32_t tot_s = (uint32_t)(total >> sh);
  sk = 0;
  if (lane < K) sk = tot_s > 0 ? 1 + (int)(((uint32_t)(J - K) * (cval >> sh)) / tot_s) : J / K;
  int incl = sk;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  start = incl - sk;
}

// Pass 1: pair counts.  Block (x, k) walks the 64-row groups of row-eighth x for offset k: cnt64[k][g] = valid rule
// entries in rows [64g, 64g+64), cnt8[x][k] = their sum.  No atomics, nothing to zero beforehand.  16 waves per block,
// each with kCntUnroll independent loads in flight (a one-load-per-trip loop here cost 30 us of pure latency).
constexpr int kCntUnroll = 4;
__global__ __launch_bounds__(1024) void k_wgrad_count(const int32_t* __restrict__ pair, int64_t ld, int64_t n,
                                                      const int64_t* d_n, int K, int G, int32_t* __restrict__ cnt64,
                                                      int32_t* __restrict__ cnt8) {
  __shared__ int32_t part[16];
  const int x = blockIdx.x, k = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t n8 = ((nlive + 7) / 8 + 63) / 64 * 64;
  const int64_t g0 = x * (n8 / 64);
  int64_t g1 = g0 + n8 / 64;
  const int64_t glive = (nlive + 63) / 64;
  if (g1 > glive) g1 = glive;
  const int32_t* prow = pair + (int64_t)k * ld;
  int total = 0;
  for (int64_t g = g0 + w * kCntUnroll; g < g1; g += 16 * kCntUnroll) {
    int32_t id[kCntUnroll];
#pragma unroll
    for (int u = 0; u < kCntUnroll; ++u) {
      const int64_t row = 64 * (g + u) + lane;
      id[u] = (g + u < g1 && row < nlive) ? prow[row] : -1;
    }
#pragma unroll
    for (int u = 0; u < kCntUnroll; ++u) {
      const int c = __popcll(__ballot(id[u] >= 0));
      if (lane == 0 && g + u < g1) cnt64[(size_t)k * G + g + u] = c;
      total += c;
    }
  }
  if (lane == 0) part[w] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += part[i];
    cnt8[x * 32 + k] = t;
  }
}

// MI consecutive floats of one row as a single load (dword / dwordx2 / dwordx4 / 2 x dwordx4)
template <int W>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[W]) {
  if constexpr (W == 1) {
    v[0] = p[0];
  } else if constexpr (W == 2) {
    float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x, v[1] = t.y;
  } else {
#pragma unroll
    for (int i = 0; i < W; i += 4) {
      float4 t = *reinterpret_cast<const float4*>(p + i);
      v[i] = t.x, v[i + 1] = t.y, v[i + 2] = t.z, v[i + 3] = t.w;
    }
  }
}

template <int W>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&v)[W]) {
  if constexpr (W == 1) {
    p[0] = v[0];
  } else if constexpr (W == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
#pragma unroll
    for (int i = 0; i < W; i += 4) *reinterpret_cast<float4*>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  }
}

// EXACT: cin == 16*MI and cout == 16*NJ (every layer of the backbone except the 4-channel input conv).
// Operand fetch of one MFMA K-step (4 compacted pairs; this lane serves pair q).  The 16 MFMA rows of tile mi are the
// channels { MI*c + mi : c = 0..15 } (not 16*mi + c): a lane then needs MI CONSECUTIVE floats of its pair's row, one
// vector load instead of MI strided dwords; the permutation is undone when the accumulators are stored.
// FULL steps (all four pairs valid) are fetched without any predicate, so the compiler can keep the next step's loads
// in flight behind the current MFMAs with a counted s_waitcnt; the single partial step of a block (at most 3 pairs,
// carried from batch to batch) is fetched with `npairs` < 4.
template <int MI, int NJ, bool EXACT, bool FULL>
__device__ __forceinline__ void wgrad_fetch(const float* __restrict__ in, int cin, const float* __restrict__ dout,
                                            int cout, const int32_t* q_src, const int32_t* q_dst, int p, int npairs,
                                            int c, float (&a)[MI], float (&b)[NJ]) {
  const bool has = FULL || (p < npairs);
  const int32_t sid = has ? q_src[p] : 0;
  const int32_t did = has ? q_dst[p] : 0;
  if (EXACT && FULL) {
    load_vec<MI>(in + (size_t)sid * cin + MI * c, a);
    load_vec<NJ>(dout + (size_t)did * cout + NJ * c, b);
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int ci = MI * c + mi;
      a[mi] = (has && ci < cin) ? in[(size_t)sid * cin + ci] : 0.f;
    }
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
      int co = NJ * c + nj;
      b[nj] = (has && co < cout) ? dout[(size_t)did * cout + co] : 0.f;
    }
  }
}

template <int MI, int NJ>
__device__ __forceinline__ void wgrad_mfma_step(f32x4 (&acc)[MI][NJ], const float (&a)[MI], const float (&b)[NJ]) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
}

template <int MI, int NJ, bool EXACT>
__global__ __launch_bounds__(64) void k_wgrad_mfma(const float* __restrict__ in, int cin,
                                                   const float* __restrict__ dout, int cout,
                                                   const int32_t* __restrict__ pair, int64_t ld, int64_t n,
                                                   const int64_t* d_n, const int32_t* __restrict__ cnt64,
                                                   const int32_t* __restrict__ cnt8, int G, int J, int K,
                                                   float* __restrict__ slab) {
  __shared__ int32_t q_src[64 * kGroups + 4];
  __shared__ int32_t q_dst[64 * kGroups + 4];
  const int lane = threadIdx.x;
  const int c = lane & 15, q = lane >> 4;
  // blocks b and b+8 share an XCD (and its L2): XCD x works on row-eighth x = rows [x*n8, (x+1)*n8)
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  int sk, start, mycnt;
  wgrad_plan(cnt8 + xcd * 32, K, J, lane, sk, start, mycnt);
  const unsigned long long mine = __ballot(lane < K && j >= start && j < start + sk);
  if (mine == 0) return;                       // j beyond the dealt blocks: its slot is never read
  // wave-uniform by construction; readfirstlane tells the compiler so (scalar loop control, no exec-masked loops)
  const int k = __ffsll((long long)mine) - 1;
  const int sub = __builtin_amdgcn_readfirstlane(j - __shfl(start, k));
  const int nsub = __builtin_amdgcn_readfirstlane(__shfl(sk, k));
  const int64_t total = __builtin_amdgcn_readfirstlane(__shfl(mycnt, k));
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t n8 = ((nlive + 7) / 8 + 63) / 64 * 64;
  // this block's rows: the 64-row groups of the eighth whose exclusive pair prefix falls in [lo, hi) — equal PAIRS
  // (= equal MFMA steps) per block of an offset, not equal rows
  const int64_t lo = total * sub / nsub, hi = total * (sub + 1) / nsub;
  const int gx0 = (int)(xcd * (n8 / 64));
  int gx1 = gx0 + (int)(n8 / 64);
  const int glive = (int)((nlive + 63) / 64);
  if (gx1 > glive) gx1 = glive;
  int n_lo = 0, n_hi = 0;
  int64_t run = 0;
  for (int g = gx0; g < gx1; g += 64) {
    const int v = g + lane < gx1 ? cnt64[(size_t)k * G + g + lane] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    const int64_t excl = run + incl - v;
    const bool in = g + lane < gx1;
    n_lo += __popcll(__ballot(in && excl < lo));
    n_hi += __popcll(__ballot(in && excl < hi));
    run += __shfl(incl, 63);
  }
  int64_t r0 = 64 * (int64_t)(gx0 + n_lo);
  int64_t r1 = 64 * (int64_t)(gx0 + n_hi);
  if (r1 > nlive) r1 = nlive;
  if (r0 > r1) r0 = r1;

#ifdef SPX_WG_DIAG
  const unsigned long long d_t0 = DIAG_T(), d_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long d_steps = 0, d_loop = 0, d_nsteps = 0;
#endif
  f32x4 acc[MI][NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) acc[mi][nj] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int32_t* prow = pair + (int64_t)k * ld;
  int carry = 0;   // 0..3 pairs left over from the previous batch, kept at the front of the queue
  // rule entries of the NEXT batch are requested one batch ahead; the loads are unconditional (row clamped) and the
  // out-of-range test is applied when the value is consumed, so nothing waits on them before that
  int32_t idn[kGroups];
  const int64_t rlast = r1 > r0 ? r1 - 1 : 0;   // (an empty block harmlessly reads entry 0)
#pragma unroll
  for (int g = 0; g < kGroups; ++g) {
    int64_t row = r0 + 64 * g + lane;
    idn[g] = prow[row < rlast ? row : rlast];
  }
  for (int64_t base = r0; base < r1; base += 64 * kGroups) {
    int32_t id[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      id[g] = base + 64 * g + lane < r1 ? idn[g] : -1;
      int64_t row = base + 64 * (kGroups + g) + lane;
      idn[g] = prow[row < rlast ? row : rlast];
    }
    int nvalid = carry;
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
    const int full = nvalid >> 2;
    carry = nvalid & 3;
    if (full == 0) continue;
    __builtin_amdgcn_wave_barrier();
    // kBurst K-steps per loop trip: all their row loads are issued first, then the MFMA groups consume them in order
    // (loads return in order, so group g waits with vmcnt(2*(kBurst-1-g)) while the later rows are still in flight).
    // Nothing in flight is carried over the back-edge, so the compiler needs no register rotation and no vmcnt(0).
    // Steps past the last one repeat it with the dout operand scaled by 0.
    const int last = full - 1;
#ifdef SPX_WG_DIAG
    const unsigned long long d_a = DIAG_T();
    d_nsteps += full;
#endif
    for (int st = 0; st < full; st += kBurst) {
      float a[kBurst][MI], b[kBurst][NJ];
#pragma unroll
      for (int g = 0; g < kBurst; ++g) {
        const int sg = st + g < last ? st + g : last;
        wgrad_fetch<MI, NJ, EXACT, true>(in, cin, dout, cout, q_src, q_dst, 4 * sg + q, 4, c, a[g], b[g]);
      }
      // all loads above, all MFMAs below: without the fence the scheduler sinks every load next to its use to save
      // registers and each step waits out the full memory latency with vmcnt(0)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < kBurst; ++g) {
        if (g > 0) {
          const float keep = st + g < full ? 1.f : 0.f;
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj) b[g][nj] *= keep;
        }
        wgrad_mfma_step<MI, NJ>(acc, a[g], b[g]);
      }
    }
#ifdef SPX_WG_DIAG
    d_loop += DIAG_T() - d_a;
#endif
    __builtin_amdgcn_wave_barrier();
    if (carry) {   // move the 1..3 leftover pairs to the front
      int32_t ts = 0, td = 0;
      if (lane < carry) ts = q_src[4 * full + lane], td = q_dst[4 * full + lane];
      __builtin_amdgcn_wave_barrier();
      if (lane < carry) q_src[lane] = ts, q_dst[lane] = td;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the one partial step.  Written as a loop with an opaque trip count (0 or 1) so that the accumulators stay
  // loop-carried, i.e. in place: as a plain `if` the register allocator gives the step a second set of 16*MI*NJ/4
  // accumulator registers and the occupancy of the whole kernel drops.
  int ntail = carry ? 1 : 0;
  ntail = __builtin_amdgcn_readfirstlane(ntail);
  asm volatile("" : "+s"(ntail));
  for (int t = 0; t < ntail; ++t) {
    __builtin_amdgcn_wave_barrier();
    float a0[MI], b0[NJ];
    wgrad_fetch<MI, NJ, EXACT, false>(in, cin, dout, cout, q_src, q_dst, q, carry, c, a0, b0);
    wgrad_mfma_step<MI, NJ>(acc, a0, b0);
  }

#ifdef SPX_WG_DIAG
  if (g_wg_diag && lane == 0) {
    unsigned long long* o = g_wg_diag + (size_t)blockIdx.x * 8;
    const unsigned long long t1 = DIAG_T();
    o[0] = t1 - d_t0;                                   // block lifetime before the epilogue, core cycles
    o[1] = d_loop;                                      // cycles inside the K-step loops
    o[2] = d_nsteps;                                    // full K-steps executed
    o[3] = __builtin_amdgcn_s_memrealtime() - d_r0;     // same lifetime in 100 MHz ticks
    o[4] = k;
    o[5] = d_t0;
    o[6] = t1;
    o[7] = (unsigned long long)(r1 > r0 ? r1 - r0 : 0);
  }
#endif
  // partial[s][k][ci][co]; C layout: tile row = 4q + e -> ci = MI*(4q+e) + mi, tile col = c -> co = NJ*c + nj
  float* out = slab + ((size_t)xcd * J + j) * cin * cout;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ci = MI * (4 * q + e) + mi;
      if (EXACT) {
        float v[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) v[nj] = acc[mi][nj][e];
        store_vec<NJ>(out + (size_t)ci * cout + NJ * c, v);
      } else if (ci < cin) {
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
          int co = NJ * c + nj;
          if (co < cout) out[(size_t)ci * cout + co] = acc[mi][nj][e];
        }
      }
    }
}

// dw[co][k][ci] = sum over the slots of offset k, fixed order.  The slots of k (XCD-major, then sub-range) form one
// list; block = 64 consecutive elements x 16 slices, slice sl sums list entries sl, sl+16, ... with four independent
// accumulators (loads in flight instead of one dependent add per memory latency); the 64 partial sums of an element are
// combined through LDS in a fixed order.  Coalesced 256-B reads.
constexpr int kRedSlices = 16;
__global__ __launch_bounds__(64 * kRedSlices) void k_wgrad_reduce(const float* __restrict__ slab,
                                                                   const int32_t* __restrict__ cnt8, int J, int K, int cin,
                                                                   int cout, float* __restrict__ dw) {
  __shared__ float part[kRedSlices][64];
  __shared__ int p_off[9][32], p_start[8][32];   // p_off[x][k] = slots of k in XCDs < x
  {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (w < 8) {
      int sk, start, mycnt;
      wgrad_plan(cnt8 + w * 32, K, J, lane, sk, start, mycnt);
      if (lane < 32) p_off[w + 1][lane] = sk, p_start[w][lane] = start;
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    int run = 0;
    for (int x = 0; x < 8; ++x) {
      int v = p_off[x + 1][threadIdx.x];
      p_off[x][threadIdx.x] = run;
      run += v;
    }
    p_off[8][threadIdx.x] = run;
  }
  __syncthreads();
  const int per = cin * cout;
  const int total = K * per;
  const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + e;
  float acc4[4] = {0.f, 0.f, 0.f, 0.f};
  if (t < total) {
    const int k = t / per, el = t - k * per;
    const int len = p_off[8][k];
    int x = 0;
    for (int i = sl; i < len; i += 4 * kRedSlices) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ii = i + u * kRedSlices;
        if (ii < len) {
          while (ii >= p_off[x + 1][k]) ++x;
          acc4[u] += slab[((size_t)x * J + p_start[x][k] + (ii - p_off[x][k])) * per + el];
        }
      }
    }
  }
  part[sl][e] = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
  __syncthreads();
  if (sl == 0 && t < total) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < kRedSlices; ++i) v += part[i][e];
    int co = t % cout, ci = (t / cout) % cin, k = t / per;
    dw[((size_t)co * K + k) * cin + ci] = v;
  }
}

}  // namespace

// blocks (= slab slots) per XCD: the row split of wgrad_splits spread over 8 XCDs, at least one block per offset
static inline int wgrad_blocks_per_xcd(int64_t n, int cin, int cout, int kvol) {
  int J = (wgrad_splits(n, cin, cout, kvol) * kvol + 7) / 8;
  return J < kvol ? kvol : J;
}

// workspace = [slab: 8*J slots of cin*cout floats][cnt64: K * G ints, G = 64-row groups][cnt8: 8 x 32 ints]
static inline size_t wgrad_slab_bytes(int cin, int cout, int kvol, int64_t n) {
  return spx_align((size_t)8 * wgrad_blocks_per_xcd(n, cin, cout, kvol) * cin * cout * sizeof(float));
}
static inline size_t wgrad_cnt64_bytes(int kvol, int64_t n) {
  return spx_align((size_t)kvol * (size_t)((n + 63) / 64 + 1) * sizeof(int32_t));
}

extern "C" size_t spx_conv_wgrad_ws_bytes(int cin, int cout, int kvol, int64_t n_out) {
  return wgrad_slab_bytes(cin, cout, kvol, n_out) + wgrad_cnt64_bytes(kvol, n_out) + spx_align(8 * 32 * sizeof(int32_t));
}

#define SPX_WG_CASE(A, B)                                                                                          \
  if (MI == A && NJ == B) {                                                                                        \
    if (exact)                                                                                                     \
      hipLaunchKernelGGL((k_wgrad_mfma<A, B, true>), dim3(8 * J), dim3(64), 0, s, in, cin, dout, cout, pair, pair_ld, n_out, \
                         d_n_out, cnt64, cnt8, G, J, kvol, slab);                                                  \
    else                                                                                                           \
      hipLaunchKernelGGL((k_wgrad_mfma<A, B, false>), dim3(8 * J), dim3(64), 0, s, in, cin, dout, cout, pair, pair_ld, n_out, \
                         d_n_out, cnt64, cnt8, G, J, kvol, slab);                                                  \
    launched = true;                                                                                               \
  }

// Pair counts of a rule table (per offset and 64-row group, per offset and row-eighth): what the work split of
// spx_conv_wgrad is derived from.  They depend on the table only — not on the channel counts — so a caller computes them
// once per table (two layers share a submanifold table; the index stream builds them ahead) and passes them in.
extern "C" size_t spx_conv_wgrad_counts_bytes(int kvol, int64_t n_out) {
  return wgrad_cnt64_bytes(kvol, n_out < 0 ? 0 : n_out) + spx_align(8 * 32 * sizeof(int32_t));
}

extern "C" int spx_conv_wgrad_counts(const int32_t* pair, int64_t pair_ld, int kvol, int64_t n_out, const int64_t* d_n_out,
                                     int32_t* counts, spx_stream_t stream) {
  if (!pair || !counts || kvol <= 0 || kvol > SPX_MAX_KVOL || n_out <= 0 || pair_ld < n_out) return SPX_ERR_INVALID_ARG;
  if (n_out >= (int64_t(1) << 31)) return SPX_ERR_TOO_LARGE;
  const int G = (int)((n_out + 63) / 64 + 1);
  int32_t* cnt8 = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(counts) + wgrad_cnt64_bytes(kvol, n_out));
  hipLaunchKernelGGL(k_wgrad_count, dim3(8, kvol), dim3(1024), 0, spx_s(stream), pair, pair_ld, n_out, d_n_out, kvol, G,
                     counts, cnt8);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

extern "C" int spx_conv_wgrad(const float* in, int cin, const float* dout, int cout, int kvol, const int32_t* pair,
                              int64_t pair_ld, int64_t n_out, const int64_t* d_n_out, const int32_t* counts, float* dw,
                              void* ws, size_t ws_bytes, spx_stream_t stream) {
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
  const int J = wgrad_blocks_per_xcd(n_out, cin, cout, kvol);
  const int G = (int)((n_out + 63) / 64 + 1);
  const int32_t* cnt64 = counts;
  const int32_t* cnt8 = counts ? reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(counts) + wgrad_cnt64_bytes(kvol, n_out))
                               : nullptr;
  if (!counts) {      // not supplied: count into the workspace
    int32_t* c64 = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + wgrad_slab_bytes(cin, cout, kvol, n_out));
    int32_t* c8 = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(c64) + wgrad_cnt64_bytes(kvol, n_out));
    hipLaunchKernelGGL(k_wgrad_count, dim3(8, kvol), dim3(1024), 0, s, pair, pair_ld, n_out, d_n_out, kvol, G, c64, c8);
    cnt64 = c64;
    cnt8 = c8;
  }
  float* slab = reinterpret_cast<float*>(ws);
  int MI = (cin + 15) / 16, NJ = (cout + 15) / 16;
  if (MI == 3) MI = 4;
  if (MI > 4) MI = 8;
  if (NJ == 3) NJ = 4;
  if (NJ > 4) NJ = 8;
  bool launched = false;
  const bool exact = cin == 16 * MI && cout == 16 * NJ;
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
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((total + 63) / 64), dim3(64 * kRedSlices), 0, s, slab, cnt8, J, kvol, cin, cout, dw);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
