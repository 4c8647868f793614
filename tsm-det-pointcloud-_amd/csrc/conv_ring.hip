// conv_ring.hip — sparse convolution forward / dgrad, round 3: every 16-row tile belongs to ONE wave for the whole launch.
//
// Same arithmetic as conv_gemm.hip / conv_balanced.hip (output-stationary implicit GEMM on v_mfma_f32_16x16x4_f32, same
// packed weights, same K-permuted gather map).  What changed is the decomposition.  conv_balanced.hip deals (64-row
// super-tile, offset) units in equal ranges to 1024 workgroups that walk them in lockstep behind one barrier per unit; at
// the sizes of the reference's layers (40-90 k rows, spconv_backbone.py:105-114) a workgroup owns ~20 units = 1.3
// super-tiles, so nearly every super-tile is SPLIT between workgroups and combined through write-through slabs, tickets and
// a read-back (26 MB of slab traffic per call, two drain + ticket periods per workgroup), and the barrier drains the row
// prefetch once per unit.  Measured: MFMA pipe 55 % busy, 0.34 of the fp32 MFMA peak.
//
// Here one workgroup per CU (12 consumer waves, three per SIMD, + 4 loader waves; <= 128 VGPRs) keeps the accumulators of its
// tiles in registers from the first offset to the last (one or two tiles per wave and turn of the ring; more rows -> more turns):
//   * no tile is ever shared: no slabs, no tickets, no combine, every output row is written once by the wave that summed
//     it over k in ascending order (bitwise reproducible, and independent of the plan);
//   * the K weight slices W_k stream through an 8-slot LDS ring (16 KiB each at 64 x 64) filled by the loader waves with
//     LDS-DMA; consumers and loaders meet through two LDS words per slot (ready = which ring index the slot holds, done = how
//     many consumer signals it has received) — NO workgroup barrier in the main loop, waves drift up to 7 offsets apart, so a
//     wave whose tiles lack an offset simply runs ahead instead of waiting for its neighbours;
//   * ring_body1 (one tile per wave and turn): each wave walks its own (tile, offset) units with a register pipeline across
//     turns — rule entries of unit i+2, rows of unit i+1 (raw buffer loads: an entry of -1 is an out-of-range offset and reads
//     as zeros — no select, no clamp) and the MFMAs of unit i; the two row buffers rotate by NAME (the loop is unrolled by two),
//     so no in-flight register is ever copied and the compiler's counted vmcnt waits leave the newest gather in flight across
//     the whole MFMA block;
//   * ring_body2 (two tiles per wave and turn): a static loop over the ring indices, every load unconditional, one wait / one
//     signal per offset for both tiles — the layers of the reference's blocks (5 tiles per SIMD) then need ONE turn of the ring;
//   * spx_conv_ring_plan (cached per rule table like the grouped rows) cuts the rows into chunks (<= 4096 rows), gives every
//     XCD the same number of chunks chosen greedily by weight (the chunk's rows stay in that XCD's L2), orders an XCD's tiles by
//     their number of non-empty offsets and deals them round by round over its 128 SIMD bins, a row's heaviest tile to the bin
//     that carries the least — the launch lasts as long as its fullest SIMD; it also picks the body from the LIVE row count;
//   * optional epilogue: per-workgroup column sums of y and y*y (the statistics pass of the training-mode BatchNorm1d that
//     follows every sparse conv, reference spconv_backbone.py:26-27,81), consumed by spx_bn_relu_fwd_from_sums.
//
// Serves the reference call sites spconv_backbone.py:86-121 (forward) and their autograd (dgrad).
#include "spx_common.h"

#ifdef SPX_RING_DIAG
// diagnostic build only (never shipped): per-wave cycle stamps, see tools/ring_diag.py
__device__ unsigned long long* g_ring_diag = nullptr;
extern "C" int spx_diag_set_ring(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ring_diag), &p, sizeof(p)); }
static __device__ __forceinline__ unsigned long long rd_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define RD_STAMP() rd_stamp()
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kRG = 256;                  // workgroups: one per CU (LDS footprint admits one)
constexpr int kCW = 12;                   // consumer waves per workgroup: waves w, w + 4 and w + 8 share a SIMD
constexpr int kTMmax = 2;                 // 16-row tiles a wave holds per round (one turn of the ring): 1 (k_conv_ring) or 2 (k_conv_ring2)
constexpr int kSlots = 8;                 // weight ring depth
constexpr int kLW = 4;                    // loader waves: loader w fills the offsets g = w (mod kLW)
constexpr int kBins = 128;                // SIMD bins per XCD range: 32 workgroups x 4 SIMDs
constexpr int kWpS = kCW / 4;              // consumer waves per SIMD
constexpr int kPerRound = kBins * kWpS;   // tiles one range places per round and tile slot (x tm)
constexpr int kMaxCh = 96;                // chunks one XCD can be given (kMaxRows / 4096 / 8, rounded up)
constexpr int kHdr = 32 + 8 * kMaxCh;     // plan header ints
// header: [0] tiles T, [1] units U, [2] spin timeouts (debug), [3] chunks, [4] tm, [8..15] rounds of XCD x, [16..23] chunks of XCD x,
// [24..31] first position of XCD x in `sorted`, [32 + x * kMaxCh + j] = j-th chunk of XCD x
// chunks = contiguous pieces of <= 4096 rows (the windows spx_conv_group sorts in): at least eight, a multiple of eight
__host__ __device__ inline int ring_chunks(int64_t nlive) {
  int64_t w = (nlive + 4095) / 4096;
  if (w < 8) w = 8;
  return (int)((w + 7) / 8 * 8);
}
__host__ __device__ inline int chunk_begin(int c, int T, int W) { return (int)((int64_t)c * T / W); }
constexpr int kSpinLimit = 1 << 18;        // ~30 ms of polling: a protocol bug ends as plan[2] != 0, never as a hang

// plan layout (int32): hdr[kHdr] | mask[tcap] | sorted[tcap] | ent[rcap][kRG][kCW][tm][2]
__host__ __device__ inline int64_t ring_tcap(int64_t n) { return (n + 15) / 16 + 1; }
// rounds an XCD can need: it holds NC / 8 chunks of at most ceil(T / NC) tiles each
__host__ __device__ inline int64_t ring_rcap(int64_t tcap, int tm) {
  return ((tcap + 7) / 8 + kMaxCh + kPerRound * tm - 1) / (kPerRound * tm) + 1;
}
constexpr int64_t kMaxRows = (int64_t)60 * kPerRound * 8 * 16;   // rounds <= 64: a wave keeps its tiles on its 64 lanes (2.9 M rows)
__host__ __device__ inline int64_t off_mask() { return kHdr; }
__host__ __device__ inline int64_t off_sorted(int64_t tcap) { return off_mask() + (tcap + 3) / 4 * 4; }
__host__ __device__ inline int64_t off_ent(int64_t tcap) { return off_sorted(tcap) + (tcap + 3) / 4 * 4; }
__host__ __device__ inline int64_t plan_ints(int64_t tcap) {      // room for either tm
  const int64_t e1 = ring_rcap(tcap, 1), e2 = ring_rcap(tcap, 2) * 2;
  return off_ent(tcap) + (e1 > e2 ? e1 : e2) * (int64_t)kRG * kCW * 2;
}

// ---------------------------------------------------------------- plan 1: offset mask per 16-row tile
__global__ __launch_bounds__(256) void k_ring_mask(const int32_t* __restrict__ pair, int64_t ld, int K, int64_t n,
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
  if (r == 0 && tile < tcap) plan[off_mask() + tile] = tile < T ? (int32_t)m : 0;
}

// ---------------------------------------------------------------- plan 2 (one block per XCD): chunks -> XCDs, order, deal
// Chunks -> XCDs.  Eight contiguous ranges of equal weight (weight of a tile = its non-empty offsets + 1 for the epilogue) need
// very different tile counts (ground planes are heavy, upper levels light), and a range over its share of tiles costs its
// workgroups a whole extra turn of the ring; contiguous ranges of equal tile count differ by 20 % in weight.  So: NC / 8 whole
// chunks per XCD (equal tile counts, each chunk a compact piece of the row order: the XCD's L2 holds those pieces), chosen
// greedily by weight, heaviest chunk first to the XCD that carries the least.  Every block computes this (tiny, deterministic)
// assignment for itself from the masks — no separate launch, no prefix array.
__global__ __launch_bounds__(1024) void k_ring_assign(int64_t n, const int64_t* d_n, int64_t tcap, int K, int tm_want,
                                                      int32_t* __restrict__ plan) {
  __shared__ int s_hist[32];
  __shared__ int s_start[32];
  __shared__ int s_wcnt[16][32];
  __shared__ int s_cw[8 * kMaxCh];         // chunk weights
  __shared__ int s_ord[8 * kMaxCh];        // chunks by descending weight
  __shared__ int s_xof[8 * kMaxCh];        // chunk -> XCD
  __shared__ int s_units;
  __shared__ int s_cpre[kMaxCh + 1];
  __shared__ int s_cbeg[kMaxCh];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int x = blockIdx.x;
  const int64_t nlive = spx_live_n(d_n, n);
  const int T = (int)((nlive + 15) / 16);
  const int NC = ring_chunks(nlive);
  // tiles per wave and turn: every XCD gets NC / 8 chunks of at most ceil(T / NC) tiles.  One tile per wave keeps the row
  // pipeline two units deep and is the faster kernel while ONE turn of the ring holds everything (measured at 2.4 tiles per
  // SIMD: 64.6 vs 71.7 us); beyond that two tiles per wave halve the turns (5.0 tiles per SIMD: 106 vs 112 us)
  const int tm = tm_want ? tm_want : ((NC / 8) * ((T + NC - 1) / NC) > kPerRound ? 2 : 1);
  const int32_t* mask = plan + off_mask();
  int32_t* sorted = plan + off_sorted(tcap);
  int32_t* ent = plan + off_ent(tcap);
  for (int c = tid; c < NC; c += 1024) s_cw[c] = 0;
  if (tid == 0) s_units = 0;
  if (tid < 32) s_hist[tid] = 0;
  __syncthreads();
  {
    int units = 0;
    for (int t = tid; t < T; t += 1024) {
      const int pc = __popc((unsigned)mask[t]);
      units += pc;
      int c = (int)((int64_t)t * NC / T);
      while (c + 1 < NC && chunk_begin(c + 1, T, NC) <= t) ++c;
      while (c > 0 && chunk_begin(c, T, NC) > t) --c;
      atomicAdd(&s_cw[c], pc + 1);                      // integer sums: order-free
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) units += __shfl_xor(units, o);
    if (lane == 0) atomicAdd(&s_units, units);
  }
  __syncthreads();
  if (x == 0 && tid == 0) {
    plan[0] = T;
    plan[1] = s_units;
    plan[2] = 0;
    plan[3] = NC;
    plan[4] = tm;
  }
  for (int c = tid; c < NC; c += 1024) {
    const int my = s_cw[c];
    int rank = 0;
    for (int o = 0; o < NC; ++o) {
      const int v = s_cw[o];
      rank += (v > my || (v == my && o < c)) ? 1 : 0;
    }
    s_ord[rank] = c;
  }
  __syncthreads();
  if (tid == 0) {
    int load[8], cnt[8];
    for (int xx = 0; xx < 8; ++xx) load[xx] = 0, cnt[xx] = 0;
    const int each = NC / 8;
    for (int i = 0; i < NC; ++i) {
      const int c = s_ord[i];
      int best = -1;
      for (int xx = 0; xx < 8; ++xx)
        if (cnt[xx] < each && (best < 0 || load[xx] < load[best])) best = xx;
      s_xof[c] = best;
      cnt[best] += 1;
      load[best] += s_cw[c];
    }
    // this XCD's tiles: its chunks in chunk order (ascending rows), one after the other; local position i -> tile
    int run = 0, j = 0, base = 0;
    for (int c = 0; c < NC; ++c) {
      const int sz = chunk_begin(c + 1, T, NC) - chunk_begin(c, T, NC);
      if (s_xof[c] < x) base += sz;
      if (s_xof[c] == x) {
        s_cpre[j] = run;
        s_cbeg[j] = chunk_begin(c, T, NC);
        plan[32 + x * kMaxCh + j] = c;
        run += sz;
        ++j;
      }
    }
    s_cpre[j] = run;
    plan[16 + x] = j;
    plan[24 + x] = base;
    s_start[0] = j;                                     // hand nch / base to the block through LDS
    s_start[1] = base;
  }
  __syncthreads();
  const int nch = s_start[0];
  const int a = s_start[1];                             // this XCD's piece of `sorted`
  const int nx = s_cpre[nch];
  __syncthreads();
  auto tile_at = [&](int i) -> int {
    int j = 0;
    while (j + 1 < nch && s_cpre[j + 1] <= i) ++j;
    return s_cbeg[j] + (i - s_cpre[j]);
  };
  const int dealRows = kWpS * tm, perRound = kBins * dealRows;
  const int rounds = (nx + perRound - 1) / perRound;
  if (tid == 0) plan[8 + x] = rounds;
  for (int i = tid; i < nx; i += 1024) atomicAdd(&s_hist[__popc((unsigned)mask[tile_at(i)])], 1);   // integer counts: order-free
  __syncthreads();
  if (tid == 0) {                     // heaviest first
    int run = 0;
    for (int v = 31; v >= 0; --v) {
      s_start[v] = run;
      run += s_hist[v];
    }
  }
  __syncthreads();
  // stable placement: chunks of 1024 tiles in order; inside a chunk rank by (wave, lane) among equal keys
  for (int base = 0; base < nx; base += 1024) {
    const int i = base + tid;
    const int tl = i < nx ? tile_at(i) : -1;
    const int key = tl >= 0 ? __popc((unsigned)mask[tl]) : -1;
    int myrank = 0;
    for (int v = 0; v < 32; ++v) {
      const unsigned long long bal = __ballot(key == v);
      if (key == v) myrank = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) s_wcnt[wv][v] = __popcll(bal);
    }
    __syncthreads();
    if (key >= 0) {
      int off = s_start[key];
      for (int w = 0; w < wv; ++w) off += s_wcnt[w][key];
      sorted[a + off + myrank] = tl;
    }
    __syncthreads();
    if (tid < 32) {
      int add = 0;
      for (int w = 0; w < 16; ++w) add += s_wcnt[w][tid];
      s_start[tid] += add;
    }
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
  // entries of this range's 32 workgroups.  A round places dealRows = 3 * tm rows of kBins tiles (heaviest tiles first): one tile per
  // SIMD bin and row, the row's heaviest tile to the bin that carries the least so far IN THIS ROUND (the ring makes the
  // twelve waves of a workgroup walk a round together, so it is the per-round load of a SIMD that has to be level; a plain
  // snake over the sorted order left the fullest SIMD 14-32 % above the mean, and the launch lasts as long as the fullest
  // SIMD).  Row `h` of a round goes to wave simd + 4 * h.
  __shared__ int s_load[kBins];
  __shared__ int s_binof[kBins];
  __shared__ int s_rt[kWpS * kTMmax * kBins];   // the round's tiles (sorted order) and their masks: two dependent global
  __shared__ int s_rm[kWpS * kTMmax * kBins];   // reads once per round instead of once per row
  for (int rd = 0; rd < rounds; ++rd) {
    if (tid < kBins) s_load[tid] = 0;
    if (tid < perRound) {
      const int p = rd * perRound + tid;
      const int tile = p < nx ? sorted[a + p] : -1;
      s_rt[tid] = tile;
      s_rm[tid] = tile >= 0 ? mask[tile] : 0;
    }
    __syncthreads();
    // rows of this round; a last row that does not fill every bin is placed FIRST (its tiles are the round's lightest): the
    // bins that end up with one tile more than the others must get lighter tiles from the full rows, and a greedy deal can
    // only arrange that if it sees the extra tiles before it places the heavy ones
    const int left = nx - rd * perRound;
    const int nrows = left >= perRound ? dealRows : (left + kBins - 1) / kBins;
    const bool partial = left < perRound && (left % kBins) != 0;
    for (int step = 0; step < nrows; ++step) {
      const int row = partial ? (step == 0 ? nrows - 1 : step - 1) : step;
      {
        const int bin = tid >> 3, part = tid & 7;        // eight threads per bin, sixteen comparisons each
        const int my = s_load[bin];
        int rank = 0;
        for (int o = 16 * part; o < 16 * part + 16; ++o) {
          const int v = s_load[o];
          rank += (v < my || (v == my && o < bin)) ? 1 : 0;
        }
        rank += __shfl_xor(rank, 1);
        rank += __shfl_xor(rank, 2);
        rank += __shfl_xor(rank, 4);
        if (part == 0) s_binof[rank] = bin;
      }
      __syncthreads();
      if (tid < kBins) {
        const int bin = s_binof[tid];
        const int tile = s_rt[row * kBins + tid];
        const int wgl = bin >> 2, simd = bin & 3;
        const int wave = simd + 4 * (row % kWpS), t = row / kWpS;
        const int wg = wgl * 8 + x;                     // blockIdx & 7 == x: the workgroups one XCD gets
        int32_t* o = ent + ((((int64_t)rd * kRG + wg) * kCW + wave) * tm + t) * 2;
        const int m = s_rm[row * kBins + tid];
        o[0] = tile;
        o[1] = m;
        if (tile >= 0) s_load[bin] += __popc((unsigned)m) + 1;
      }
      __syncthreads();
    }
    // rows of the round that hold no tile at all: empty entries
    for (int row = nrows; row < dealRows; ++row) {
      if (tid < kBins) {
        const int wgl = tid >> 2, simd = tid & 3;
        const int wave = simd + 4 * (row % kWpS), t = row / kWpS;
        int32_t* o = ent + ((((int64_t)rd * kRG + (wgl * 8 + x)) * kCW + wave) * tm + t) * 2;
        o[0] = -1;
        o[1] = 0;
      }
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------- the weight ring (shared by both kernels)
// ready[slot] = g + 1 once the slot holds the weights of ring index g; done[slot] = consumer signals the slot has received in
// all its visits; abort = set by the first spin that gave up (every later wait returns at once).
struct RingCtl {
  f32x4* ring;
  int* ready;
  int* done;
  int* abort;
  int32_t* plan;
  int32_t* status;   // device status word (nullable): receives SPX_ERR_RING_STALL
};

__device__ __forceinline__ void ring_give_up(const RingCtl& c, int lane) {
  if (lane == 0) {
    atomicAdd(&c.plan[2], 1);
    if (c.status) atomicMin(c.status, (int32_t)SPX_ERR_RING_STALL);
    __hip_atomic_store(c.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// Loader waves: W_k of ring index g -> slot g % kSlots.  One fill is serial (wait for the slot, NF x 1 KiB LDS-DMA, wait for
// them to land ~1.1 us, publish), so a single loader delivers a slice per ~1.3 us — slower than a round of light tiles consumes
// them (first version: the launch was loader-bound).  kLW loaders take the indices round-robin: four fills in flight at a time.
template <int NF>
__device__ __forceinline__ void ring_fill(const RingCtl& c, const float* wp, int first, int total, int K, int lane) {
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);
  for (int g = first; g < total; g += kLW) {
    const int slot = g & (kSlots - 1);
    const int k = g % K;
    if (g >= kSlots) {                        // every consumer is through with the slot's previous index
      const int target = kCW * (g / kSlots);
      int spins = 0;
      while (__hip_atomic_load(&c.done[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != target) {
        if (spins < 8) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(6);
        if (++spins > kSpinLimit || __hip_atomic_load(c.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
          ring_give_up(c, lane);
          break;
        }
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int f = 0; f < NF; ++f)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp4 + ((size_t)k * NF + f) * 64 + lane),
                                       (__attribute__((address_space(3))) void*)(&c.ring[(slot * NF + f) * 64]), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&c.ready[slot], g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// Consumer side.  A consumer signals a ring index only after it has SEEN the slot hold it (also indices it has no unit at):
// then no signal of a slot's next visit can arrive before all signals of the current one (the loaders publish g only after
// every consumer signalled g - kSlots), and the per-slot counters never mix visits.
// One poll reads all kSlots words at once: ghz = first ring index NOT known to be published; indices below it need no
// further LDS round trip (a slot cannot change before this wave has signalled it).  The wave that lags — the one every
// other wave is waiting for — therefore runs its units without a single poll on its path (first version: two LDS round
// trips of ~700 cycles per unit, on the critical wave).
__device__ __forceinline__ void ring_wait(const RingCtl& c, int g, int& ghz, int lane) {
#ifdef SPX_RING_NO_PROTO                       // ablation (dev builds): nobody waits, nobody signals; wrong results
  return;
#endif
  if (g < ghz) return;
  int spins = 0;
  for (;;) {
    const int j = lane & (kSlots - 1);
    const int v = __hip_atomic_load(&c.ready[(g + j) & (kSlots - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned m = (unsigned)__ballot(v == g + j + 1) & ((1u << kSlots) - 1u);
    ghz = g + __builtin_ctz(~m | (1u << kSlots));
    if (g < ghz) break;
    if (spins < 8) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(8);
    if (++spins > kSpinLimit || __hip_atomic_load(c.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
      ring_give_up(c, lane);
      ghz = g + 1;
      break;
    }
  }
}
__device__ __forceinline__ void ring_signal(const RingCtl& c, int g, int lane) {
#ifndef SPX_RING_NO_PROTO
  if (lane == 0) __hip_atomic_fetch_add(&c.done[g & (kSlots - 1)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

// ---------------------------------------------------------------- the convolution
// LDS of a workgroup, one object: ring | statistics scratch | control words (ready[kSlots], done[kSlots], abort, padding)
template <int CS, int CD>
constexpr int ring_smem_f4() { return kSlots * (CD / 16) * (CS / 16) * 64 + kCW * 2 * CD / 4 + 2 * kSlots / 4 + 4; }

// ---- one tile per wave and turn of the ring (plans with tm = 1)
template <int CS, int CD>
__device__ __forceinline__ void ring_body1(
    const float* __restrict__ src, int64_t n_src, const float* __restrict__ wp, const int32_t* __restrict__ pair, int64_t ld,
    int K, int flip, int64_t n, const int64_t* d_n, const float* __restrict__ scale, const float* __restrict__ shift,
    int relu, int32_t* __restrict__ plan, int64_t tcap, const int32_t* __restrict__ perm, float* __restrict__ dst,
    float* __restrict__ stats, int32_t* status, f32x4* smem) {
  constexpr int NT = CD / 16;
  constexpr int JG = CS / 16;
  constexpr int NF = NT * JG;                 // 1 KiB weight fragments per offset
  constexpr int kStatF4 = kCW * 2 * CD / 4;   // cross-wave statistics scratch, in f32x4
  f32x4* ring = smem;
  float* s_stat = reinterpret_cast<float*>(smem + kSlots * NF * 64);
  int* s_ready = reinterpret_cast<int*>(smem + kSlots * NF * 64 + kStatF4);
  int* s_done = s_ready + kSlots;
  int* s_abort = s_done + kSlots;             // set by the first spin that gave up: every later wait returns at once
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int x = blockIdx.x & 7;
  const int R = plan[8 + x];
  if (threadIdx.x < 2 * kSlots + 1) s_ready[threadIdx.x] = 0;
  __syncthreads();                            // the only workgroup barrier before the statistics epilogue

  const RingCtl ctl{ring, s_ready, s_done, s_abort, plan, status};
  // ------------------------------------------------------------ loader waves
  if (wave >= kCW) {
#ifdef SPX_RING_NO_PROTO
    const int total = kSlots < R * K ? kSlots : R * K;
#else
    const int total = R * K;
#endif
    ring_fill<NF>(ctl, wp, wave - kCW, total, K, lane);
    if (stats) __syncthreads();               // matches the consumers' barrier of the statistics epilogue
    return;
  }

  // ------------------------------------------------------------ consumer waves
#ifdef SPX_RING_DIAG
  const unsigned long long d_t0 = RD_STAMP(), d_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long d_proto = 0, d_mma = 0, d_issue = 0, d_units = 0, d_round0 = 0, d_epi = 0, d_pass = 0, d_first = 0, d_last = 0;
#endif
  const int nlive = (int)spx_live_n(d_n, n);
  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(n_src * CS * (int64_t)sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_pair =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(pair), 0, (int)((int64_t)K * ld * (int64_t)sizeof(int32_t)), 0x00020000);
  const int ldb = (int)(ld * 4), ld_i = (int)ld;
  const int G = R * K;                         // ring indices of this workgroup: g = round * K + offset

  // column sums of what this wave writes (wave-private rows of s_stat: no barrier until the end)
  if (stats && q == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      s_stat[(wave * 2 + 0) * CD + 16 * nt + r] = 0.f;
      s_stat[(wave * 2 + 1) * CD + 16 * nt + r] = 0.f;
    }
  }

  // this wave's tiles: round i on lane i (R <= 64, checked by the plan)
  int tileL = -1;
  uint32_t maskL = 0;
  if (lane < R) {
    const int2 e = *reinterpret_cast<const int2*>(plan + off_ent(tcap) + (((int64_t)lane * kRG + blockIdx.x) * kCW + wave) * 2);
    tileL = e.x;
    maskL = (uint32_t)e.y;                     // bit = table row
  }

  // ---- unit iterator over all rounds: the wave's (tile, offset) units in ring order.  A unit = (g, k, rd)
  int it_rd = -1, it_row = 0;
  uint32_t it_rem = 0;
  auto next_unit = [&](int& ug, int& uk, int& urd) {
    while (it_rem == 0 && it_rd + 1 < R) {
      ++it_rd;
      const int tl = __builtin_amdgcn_readlane(tileL, it_rd);
      const uint32_t mt = (uint32_t)__builtin_amdgcn_readlane((int)maskL, it_rd);
      it_rem = tl >= 0 ? (flip ? (__brev(mt) >> (32 - K)) : mt) : 0u;     // offsets in LOOP order (weights W_k)
      const int row = (tl > 0 ? tl : 0) * 16 + r;
      it_row = row < ld_i ? row : ld_i - 1;
    }
    if (it_rem == 0) {
      ug = G, uk = K - 1, urd = R;             // past the end
      return;
    }
    uk = __ffs((int)it_rem) - 1;
    it_rem &= it_rem - 1;
    urd = it_rd;
    ug = it_rd * K + uk;
  };
  // rule entries of the unit the iterator has just produced: lane (r, q) reads the entry of row r of its tile (the four q
  // copies coalesce); the raw value: -1 = no pair.  (Rows beyond the live count of the LAST tile read whatever the table
  // holds there; their accumulator rows are never stored, and every gather is bounds-checked by the buffer hardware.)
  auto load_id = [&](int k) -> int32_t {
    const int tr = flip ? K - 1 - k : k;
    return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_pair, it_row * 4, __builtin_amdgcn_readfirstlane(tr * ldb), 0);
  };
  // rows of a unit: entry -1 = an offset beyond the records -> zeros
  auto gather = [&](int32_t id, f32x4 (&a)[JG]) {
    const int voff = id * (CS * 4) + 16 * q;
#ifdef SPX_RING_NO_GATHER                      // ablation (dev builds): no row traffic, operands from the entry value
#pragma unroll
    for (int jg = 0; jg < JG; ++jg) a[jg] = f32x4{(float)voff, 1.f, (float)jg, 2.f};
#else
#pragma unroll
    for (int jg = 0; jg < JG; ++jg)
      a[jg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, voff + 64 * jg, 0, 0));
#endif
  };
  int ghz = 0;
  auto wait_ready = [&](int g) { ring_wait(ctl, g, ghz, lane); };
  auto pass_indices = [&](int from, int to) {   // ring indices [from, to): seen, then signalled, in order
    for (int g = from; g < to; ++g) {
      wait_ready(g);
      asm volatile("" ::: "memory");
      ring_signal(ctl, g, lane);
    }
  };

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // write the tile of round rd from `acc` (epilogue, statistics), then clear acc.  C layout of the MFMA: col = lane & 15,
  // row = 4 * (lane >> 4) + e.  No branch per element: a row beyond the live count gets a store offset beyond the records,
  // which the buffer hardware drops.
  const __amdgpu_buffer_rsrc_t rs_dst =
      __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(n * CD * (int64_t)sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_perm =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(perm), 0, perm ? (int)(n * (int64_t)sizeof(int32_t)) : 0, 0x00020000);
  auto write_tile = [&](int rd) {
    const int tile = __builtin_amdgcn_readlane(tileL, rd);
    if (tile >= 0) {
      const int orow0 = tile * 16 + 4 * q;
      int drow[4];
      if (perm) {
        const u32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(rs_perm, orow0 * 4, 0, 0);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) drow[e4] = (int)pv[e4];
      } else {
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) drow[e4] = orow0 + e4;
      }
      float live[4];
      int voff[4];
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const bool ok = orow0 + e4 < nlive;
        live[e4] = ok ? 1.0f : 0.0f;
        voff[e4] = ok ? drow[e4] * (CD * 4) + 4 * r : (int)0xFFFFFF00u;
      }
      const bool affine = scale != nullptr || shift != nullptr;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float sc = 1.0f, sh = 0.0f;
        if (affine) {
          sc = scale ? scale[16 * nt + r] : 1.0f;
          sh = shift ? shift[16 * nt + r] : 0.0f;
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          float v = acc[nt][e4];
          if (affine) v = v * sc + sh;
          if (relu) v = v > 0.f ? v : 0.f;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_dst, voff[e4] + 64 * nt, 0, 0);
          const float vl = v * live[e4];
          s1 += vl;
          s2 += vl * v;
        }
        if (stats) {                             // the four q groups of a column: lanes r, r + 16, r + 32, r + 48
          s1 += __shfl_xor(s1, 16);
          s1 += __shfl_xor(s1, 32);
          s2 += __shfl_xor(s2, 16);
          s2 += __shfl_xor(s2, 32);
          if (q == 0) {
            s_stat[(wave * 2 + 0) * CD + 16 * nt + r] += s1;
            s_stat[(wave * 2 + 1) * CD + 16 * nt + r] += s2;
          }
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  int g0, k0, r0, g1, k1, r1, g2, k2, r2;
  int32_t idX, idY;
  f32x4 aX[JG], aY[JG];
  next_unit(g0, k0, r0);
  idX = load_id(k0);
  next_unit(g1, k1, r1);
  idY = load_id(k1);
  gather(idX, aX);
  next_unit(g2, k2, r2);
  int gsig = 0;                                // ring indices already signalled
  // tiles of the rounds before the first unit that have no offset at all (backward tables): written as epilogue(0)
  for (int rr = 0; rr < (r0 < R ? r0 : R); ++rr) write_tile(rr);

#ifdef SPX_RING_DIAG
  unsigned long long d_a = 0, d_b = 0, d_c = 0, d_p = 0;
#define RD_A() { d_a = RD_STAMP(); if (d_first == 0) d_first = d_a - d_t0; }
#define RD_B() { d_b = RD_STAMP(); d_issue += d_b - d_a; }
#define RD_P() { d_p = RD_STAMP(); d_pass += d_p - d_b; }
#define RD_C() { d_c = RD_STAMP(); d_proto += d_c - d_p; }
#define RD_D() { d_last = RD_STAMP(); d_mma += d_last - d_c; d_units += 1; }
#define RD_E0() const unsigned long long d_e = RD_STAMP()
#define RD_E1() { d_epi += RD_STAMP() - d_e; if (epi_from == 0) d_round0 = RD_STAMP() - d_t0; }
#else
#define RD_A()
#define RD_B()
#define RD_P()
#define RD_C()
#define RD_D()
#define RD_E0()
#define RD_E1()
#endif

  // The weight fragments of a unit are a STREAM of NF 1 KiB reads through a ring of eight fragment registers: read j goes to
  // bq[j & 7], six reads run ahead of the MFMAs that consume them (two reads = one group of eight MFMAs on two accumulators,
  // each accumulator every other MFMA).  First version: four reads, wait for all, 16 MFMAs, next four reads — the LDS latency
  // (~600 cycles with twelve waves reading) was exposed four times per unit on the wave everyone else waits for.
#ifdef SPX_RING_NO_BREAD                       // ablation (dev builds): fragments from registers, no LDS reads
#define RB(B, o) f32x4{(float)(o), 1.f, (float)lane, 3.f}
#else
#define RB(B, o) (B)[o]
#endif
  constexpr int NP = NT / 2;                   // accumulator pairs
  constexpr int NG = JG * NP;                  // MFMA groups per unit, two fragment reads each
  constexpr int kAhead = 6 < NF ? 6 : NF;      // reads in flight ahead of their group
  // (tried, no gain: the next unit's first four reads issued before this unit's last two groups; wave priorities 3/2/1)
  static_assert(NT % 2 == 0, "pairs of accumulators");
  f32x4 bq[8];

  // one step: unit (g0, k0, r0) with rows in ACUR; loads the entries of the unit two ahead (the one the iterator has just
  // produced) into IDLOAD and gathers the rows of the next unit (entries IDUSE, loaded one step ago) into ANXT — both stay in
  // flight across this step's MFMAs.  The pipeline runs across tiles: the next tile's rows are on their way while this one is
  // written.
#define SPX_RING_STEP(ACUR, ANXT, IDLOAD, IDUSE)                                                                       \
  {                                                                                                                     \
    if (g0 >= G) goto ring_done;                                                                                        \
    RD_A();                                                                                                             \
    /* leaving ring indices [gsig, g0): this wave's reads of their slots are complete (their MFMAs were issued) */      \
    asm volatile("" ::: "memory");                                                                                      \
    pass_indices(gsig, g0);                                                                                             \
    gsig = g0;                                                                                                          \
    RD_P();                                                                                                             \
    wait_ready(g0);                                                                                                     \
    asm volatile("" ::: "memory");                                                                                      \
    RD_C();                                                                                                             \
    const f32x4* B = ring + (size_t)(g0 & (kSlots - 1)) * NF * 64 + lane;                                               \
    _Pragma("unroll") for (int j = 0; j < kAhead; ++j) {                                                                \
      const int gi = j >> 1, jg = gi / NP, nt = 2 * (gi % NP) + (j & 1);                                                \
      bq[j & 7] = RB(B, (nt * JG + jg) * 64);                                                                           \
    }                                                                                                                   \
    IDLOAD = load_id(k2);                                                                                               \
    gather(IDUSE, ANXT);                                                                                                \
    /* everything above is only ISSUED here; the scheduler would otherwise sink the gather next to its use in the next */ \
    /* step and serialise its latency with the arithmetic */                                                            \
    asm volatile("" ::: "memory");                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    RD_B();                                                                                                             \
    _Pragma("unroll") for (int gi = 0; gi < NG; ++gi) {                                                                 \
      const int jg = gi / NP, n0 = 2 * (gi % NP);                                                                       \
      _Pragma("unroll") for (int j = 2 * gi + kAhead; j < 2 * gi + kAhead + 2; ++j) {                                   \
        if (j < NF) {                                                                                                   \
          const int gj = j >> 1, jj = gj / NP, nn = 2 * (gj % NP) + (j & 1);                                            \
          bq[j & 7] = RB(B, (nn * JG + jj) * 64);                                                                       \
        }                                                                                                               \
      }                                                                                                                 \
      _Pragma("unroll") for (int ee = 0; ee < 4; ++ee) {                                                                \
        acc[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ACUR[jg][ee], bq[(2 * gi) & 7][ee], acc[n0], 0, 0, 0);         \
        acc[n0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ACUR[jg][ee], bq[(2 * gi + 1) & 7][ee], acc[n0 + 1], 0, 0, 0); \
      }                                                                                                                 \
    }                                                                                                                   \
    RD_D();                                                                                                             \
    if (r1 != r0) epi_from = r0, epi_to = r1 < R ? r1 : R;   /* last unit of its tile: write it (below) */              \
    g0 = g1, k0 = k1, r0 = r1, g1 = g2, k1 = k2, r1 = r2;                                                               \
    next_unit(g2, k2, r2);                                                                                              \
  }

  // the two steps swap the buffer NAMES; a tile's end leaves the rotation, writes the tile at ONE place in the code and
  // re-enters at the other phase
  {
    int phase = 0, epi_from = -1, epi_to = -1;
    for (;;) {
      if (epi_from >= 0) {
        RD_E0();
        for (int rr = epi_from; rr < epi_to; ++rr) write_tile(rr);   // the tile, and any after it without an offset
        RD_E1();
        epi_from = -1;
      }
      if (phase == 0) {
        SPX_RING_STEP(aX, aY, idX, idY)
        if (epi_from >= 0) {
          phase = 1;
          continue;
        }
      }
      SPX_RING_STEP(aY, aX, idY, idX)
      phase = 0;
    }
  }
ring_done:
#undef SPX_RING_STEP
#undef RB
#ifdef SPX_RING_DIAG
  const unsigned long long d_e0 = RD_STAMP();
#endif
  // the ring indices this wave has not signalled yet (its last one, and any it had no unit at)
  asm volatile("" ::: "memory");
  pass_indices(gsig, G);
#ifdef SPX_RING_DIAG
  const unsigned long long d_tail = RD_STAMP() - d_e0;
  if (g_ring_diag && lane == 0) {
    unsigned long long* o = g_ring_diag + ((size_t)blockIdx.x * kCW + wave) * 16;
    o[0] = RD_STAMP() - d_t0;                      // wave lifetime, core cycles
    o[1] = d_units;
    o[2] = d_proto;                                // ring protocol: pass_indices + wait_ready (+ the tail after the last unit)
    o[3] = d_mma;                                  // LDS reads + MFMAs of the units
    o[4] = d_issue;                                // entry load + gather issue (incl. the wait for the entries)
    o[5] = d_r0;
    o[6] = __builtin_amdgcn_s_memrealtime();
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    o[7] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    o[8] = d_round0;
    o[9] = d_epi;
    o[10] = d_pass;                                // signalling of the indices left behind (own previous one, skipped ones)
    o[11] = d_tail;                                // after the last unit: the rest of the ring
    o[12] = d_first;                               // wave start -> first step
    o[13] = d_last - d_t0;                         // wave start -> end of the last unit's MFMAs
  }
#endif
#undef RD_A
#undef RD_B
#undef RD_P
#undef RD_C
#undef RD_D
#undef RD_E0
#undef RD_E1

  // ------------------------------------------------------------ statistics: column sums over the rows this workgroup wrote
  if (stats) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * CD; i += 64 * kCW) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < kCW; ++w) s += s_stat[w * 2 * CD + i];      // fixed order
      stats[(size_t)blockIdx.x * 2 * CD + i] = s;
    }
  }
}


// ---------------------------------------------------------------- the convolution, two tiles per wave (plans with tm = 2)
// Same ring, same loaders, same plan; a consumer wave holds TWO tiles per round (accumulators of both in registers) and walks
// the ring indices in a STATIC loop: at index g = round * K + k it multiplies the rows of both tiles with W_k — the two
// units share every weight fragment read from LDS (half the LDS reads per MFMA of k_conv_ring) and one wait / one signal of
// the ring protocol.  The layers of the reference's blocks (40-90 k rows = 2.4-5.5 tiles per SIMD) then need ONE turn of the
// ring instead of two; k_conv_ring paid ~15 us per turn in ring fill, drain and protocol.
// No unit iterator: every load is unconditional (a tile that lacks offset k has entries of -1 there, which gather as
// out-of-range zeros without memory traffic) and only the MFMA block is skipped, so the order and number of vector memory
// operations per index is fixed and every s_waitcnt is a counted one.  One row buffer per tile: the 16 bytes of K-slice jg are
// re-fetched for index g + 1 right after the MFMAs of g that read them were issued (a full index of head start), entries run
// two indices ahead.
template <int CS, int CD>
__device__ __forceinline__ void ring_body2(
    const float* __restrict__ src, int64_t n_src, const float* __restrict__ wp, const int32_t* __restrict__ pair, int64_t ld,
    int K, int flip, int64_t n, const int64_t* d_n, const float* __restrict__ scale, const float* __restrict__ shift,
    int relu, int32_t* __restrict__ plan, int64_t tcap, const int32_t* __restrict__ perm, float* __restrict__ dst,
    float* __restrict__ stats, int32_t* status, f32x4* smem) {
  constexpr int NT = CD / 16;
  constexpr int JG = CS / 16;
  constexpr int NF = NT * JG;
  constexpr int kStatF4 = kCW * 2 * CD / 4;
  f32x4* ring = smem;
  float* s_stat = reinterpret_cast<float*>(smem + kSlots * NF * 64);
  int* s_ready = reinterpret_cast<int*>(smem + kSlots * NF * 64 + kStatF4);
  int* s_done = s_ready + kSlots;
  int* s_abort = s_done + kSlots;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int x = blockIdx.x & 7;
  const int R = plan[8 + x];
  if (threadIdx.x < 2 * kSlots + 1) s_ready[threadIdx.x] = 0;
  __syncthreads();
  const RingCtl ctl{ring, s_ready, s_done, s_abort, plan, status};
  const int G = R * K;
  if (wave >= kCW) {
    ring_fill<NF>(ctl, wp, wave - kCW, G, K, lane);
    if (stats) __syncthreads();
    return;
  }

  const int nlive = (int)spx_live_n(d_n, n);
  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(n_src * CS * (int64_t)sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_pair =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(pair), 0, (int)((int64_t)K * ld * (int64_t)sizeof(int32_t)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dst =
      __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(n * CD * (int64_t)sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_perm =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(perm), 0, perm ? (int)(n * (int64_t)sizeof(int32_t)) : 0, 0x00020000);
  const int ldb = (int)(ld * 4), ld_i = (int)ld;

  if (stats && q == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      s_stat[(wave * 2 + 0) * CD + 16 * nt + r] = 0.f;
      s_stat[(wave * 2 + 1) * CD + 16 * nt + r] = 0.f;
    }
  }

  // this wave's tiles: slot t of round i on lane i (R <= 64, checked by the plan)
  int tileAL = -1, tileBL = -1;
  uint32_t maskAL = 0, maskBL = 0;
  if (lane < R) {
    const int4 e = *reinterpret_cast<const int4*>(plan + off_ent(tcap) + (((int64_t)lane * kRG + blockIdx.x) * kCW + wave) * 4);
    tileAL = e.x, maskAL = (uint32_t)e.y, tileBL = e.z, maskBL = (uint32_t)e.w;      // mask bit = table row
  }
  auto tile_of = [&](int tl, int rd) -> int { return rd < R ? __builtin_amdgcn_readlane(tl, rd < R ? rd : 0) : -1; };
  auto row_of = [&](int tile) -> int {         // the lane's table row of a tile, as a byte offset (an absent tile: row 0)
    const int row = (tile > 0 ? tile : 0) * 16 + r;
    return (row < ld_i ? row : ld_i - 1) * 4;
  };
  // rule entries of (tile, k): lane (r, q) reads the entry of row r (the four q copies coalesce).  -1 = no pair; an absent
  // tile reads row 0 and is overridden with -1.  (Rows beyond the live count of the LAST tile read whatever the table holds
  // there; their accumulator rows are never stored, and every gather is bounds-checked by the buffer hardware.)
  auto load_id = [&](int rowb, int k, bool present) -> int32_t {
    const int tr = flip ? K - 1 - k : k;
    const int32_t v = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_pair, rowb, __builtin_amdgcn_readfirstlane(tr * ldb), 0);
    return present ? v : -1;
  };

  f32x4 accA[NT], accB[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // epilogue of one tile (see k_conv_ring): C layout col = lane & 15, row = 4 * (lane >> 4) + e; rows beyond the live count
  // get a store offset beyond the records, which the buffer hardware drops
  auto write_tile = [&](f32x4 (&acc)[NT], int tile) {
    if (tile >= 0) {
      const int orow0 = tile * 16 + 4 * q;
      int drow[4];
      if (perm) {
        const u32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(rs_perm, orow0 * 4, 0, 0);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) drow[e4] = (int)pv[e4];
      } else {
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) drow[e4] = orow0 + e4;
      }
      float live[4];
      int voff[4];
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const bool ok = orow0 + e4 < nlive;
        live[e4] = ok ? 1.0f : 0.0f;
        voff[e4] = ok ? drow[e4] * (CD * 4) + 4 * r : (int)0xFFFFFF00u;
      }
      const bool affine = scale != nullptr || shift != nullptr;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float sc = 1.0f, sh = 0.0f;
        if (affine) {
          sc = scale ? scale[16 * nt + r] : 1.0f;
          sh = shift ? shift[16 * nt + r] : 0.0f;
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          float v = acc[nt][e4];
          if (affine) v = v * sc + sh;
          if (relu) v = v > 0.f ? v : 0.f;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_dst, voff[e4] + 64 * nt, 0, 0);
          const float vl = v * live[e4];
          s1 += vl;
          s2 += vl * v;
        }
        if (stats) {
          s1 += __shfl_xor(s1, 16);
          s1 += __shfl_xor(s1, 32);
          s2 += __shfl_xor(s2, 16);
          s2 += __shfl_xor(s2, 32);
          if (q == 0) {
            s_stat[(wave * 2 + 0) * CD + 16 * nt + r] += s1;
            s_stat[(wave * 2 + 1) * CD + 16 * nt + r] += s2;
          }
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  constexpr int NP = NT / 2;
  constexpr int kQ = 8;                         // fragment registers: reads run kAhead = kQ - 2 ahead of their group
  constexpr int kAhead = kQ - 2 < NF ? kQ - 2 : NF;
  static_assert(NT % 2 == 0, "pairs of accumulators");
  f32x4 rA[JG], rB[JG], bq[kQ];

  // ---- prologue: rows of index 0, entries of index 1
  int tA = tile_of(tileAL, 0), tB = tile_of(tileBL, 0);
  uint32_t mA = R > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)maskAL, 0) : 0u;
  uint32_t mB = R > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)maskBL, 0) : 0u;
  int rd1 = 0, k1 = 0;                          // (round, offset) of index g + 1, then g + 2 below
  int t1A = tA, t1B = tB, rowA = row_of(tA), rowB = row_of(tB);
  int32_t idA = load_id(rowA, 0, tA >= 0), idB = load_id(rowB, 0, tB >= 0);
  {
    const int vA = idA * (CS * 4) + 16 * q, vB = idB * (CS * 4) + 16 * q;
#pragma unroll
    for (int jg = 0; jg < JG; ++jg) {
      rA[jg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, vA + 64 * jg, 0, 0));
      rB[jg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, vB + 64 * jg, 0, 0));
    }
  }
  auto advance = [&]() {                        // (rd1, k1) -> the next index; refreshes the tiles / rows it reads entries of
    if (++k1 == K) {
      k1 = 0;
      ++rd1;
      t1A = tile_of(tileAL, rd1);
      t1B = tile_of(tileBL, rd1);
      rowA = row_of(t1A);
      rowB = row_of(t1B);
    }
  };
  advance();                                    // index 1
  idA = load_id(rowA, k1, t1A >= 0);
  idB = load_id(rowB, k1, t1B >= 0);
  advance();                                    // index 2

  // the MFMAs of one tile at the current index: a stream of NF fragment reads through kQ registers, two reads (one group of
  // eight MFMAs on two accumulators) ahead
#define SPX_DUO_UNIT(ACC, ROWS)                                                                                        \
  {                                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < kAhead; ++j) {                                                                \
      const int gi = j >> 1, jg = gi / NP, nt = 2 * (gi % NP) + (j & 1);                                                \
      bq[j % kQ] = Bf[(nt * JG + jg) * 64];                                                                             \
    }                                                                                                                   \
    _Pragma("unroll") for (int gi = 0; gi < JG * NP; ++gi) {                                                            \
      const int jg = gi / NP, n0 = 2 * (gi % NP);                                                                       \
      _Pragma("unroll") for (int j = 2 * gi + kAhead; j < 2 * gi + kAhead + 2; ++j) {                                   \
        if (j < NF) {                                                                                                   \
          const int gj = j >> 1, jj = gj / NP, nn = 2 * (gj % NP) + (j & 1);                                            \
          bq[j % kQ] = Bf[(nn * JG + jj) * 64];                                                                         \
        }                                                                                                               \
      }                                                                                                                 \
      _Pragma("unroll") for (int ee = 0; ee < 4; ++ee) {                                                                \
        ACC[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ROWS[jg][ee], bq[(2 * gi) % kQ][ee], ACC[n0], 0, 0, 0);          \
        ACC[n0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ROWS[jg][ee], bq[(2 * gi + 1) % kQ][ee], ACC[n0 + 1], 0, 0, 0); \
      }                                                                                                                 \
    }                                                                                                                   \
  }
#define SPX_DUO_FETCH(ROWS, V)                                                                                         \
  {                                                                                                                     \
    _Pragma("unroll") for (int jg = 0; jg < JG; ++jg)                                                                   \
      ROWS[jg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, V + 64 * jg, 0, 0));           \
    asm volatile("" ::: "memory");                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  }

  int ghz = 0, k = 0, rd = 0;
  for (int g = 0; g < G; ++g) {
    // entries of index g + 2 (older than this index's gathers: complete, without a wait of their own, when g + 1 needs them)
    const int32_t idA2 = load_id(rowA, k1, t1A >= 0), idB2 = load_id(rowB, k1, t1B >= 0);
    const int vA = idA * (CS * 4) + 16 * q, vB = idB * (CS * 4) + 16 * q;     // rows of index g + 1
    const int tr = flip ? K - 1 - k : k;
    const bool pA = tA >= 0 && ((mA >> tr) & 1u), pB = tB >= 0 && ((mB >> tr) & 1u);
    ring_wait(ctl, g, ghz, lane);
    asm volatile("" ::: "memory");
    const f32x4* Bf = ring + (size_t)(g & (kSlots - 1)) * NF * 64 + lane;
    // tile A, then its rows of index g + 1 (in flight across tile B's MFMAs); tile B likewise (in flight across tile A's of g + 1)
    if (pA) SPX_DUO_UNIT(accA, rA)
    SPX_DUO_FETCH(rA, vA)
    if (pB) SPX_DUO_UNIT(accB, rB)
    SPX_DUO_FETCH(rB, vB)
    // this wave's reads of the slot are complete (the MFMAs that consume them were issued)
    asm volatile("" ::: "memory");
    ring_signal(ctl, g, lane);
    idA = idA2;
    idB = idB2;
    advance();
    if (++k == K) {                              // the round's last offset: write both tiles, next round's tiles
      write_tile(accA, tA);
      write_tile(accB, tB);
      k = 0;
      ++rd;
      tA = tile_of(tileAL, rd);
      tB = tile_of(tileBL, rd);
      mA = rd < R ? (uint32_t)__builtin_amdgcn_readlane((int)maskAL, rd < R ? rd : 0) : 0u;
      mB = rd < R ? (uint32_t)__builtin_amdgcn_readlane((int)maskBL, rd < R ? rd : 0) : 0u;
    }
  }
#undef SPX_DUO_UNIT
#undef SPX_DUO_FETCH

  if (stats) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * CD; i += 64 * kCW) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < kCW; ++w) s += s_stat[w * 2 * CD + i];      // fixed order
      stats[(size_t)blockIdx.x * 2 * CD + i] = s;
    }
  }
}

// One launch serves both decompositions: the plan says which one its entries were dealt for (plan[4], chosen by
// k_ring_assign from the LIVE row count, which the host does not know under static capacities).
template <int CS, int CD>
__global__ __launch_bounds__(64 * (kCW + kLW)) void k_conv_ring(
    const float* __restrict__ src, int64_t n_src, const float* __restrict__ wp, const int32_t* __restrict__ pair, int64_t ld,
    int K, int flip, int64_t n, const int64_t* d_n, const float* __restrict__ scale, const float* __restrict__ shift,
    int relu, int32_t* __restrict__ plan, int64_t tcap, const int32_t* __restrict__ perm, float* __restrict__ dst,
    float* __restrict__ stats, int32_t* status) {
  __shared__ f32x4 smem[ring_smem_f4<CS, CD>()];
  if (plan[4] == 2)
    ring_body2<CS, CD>(src, n_src, wp, pair, ld, K, flip, n, d_n, scale, shift, relu, plan, tcap, perm, dst, stats, status, smem);
  else
    ring_body1<CS, CD>(src, n_src, wp, pair, ld, K, flip, n, d_n, scale, shift, relu, plan, tcap, perm, dst, stats, status, smem);
}

template <int CS, int CD>
static void launch_ring(const float* src, int64_t n_src, const float* wp, const int32_t* pair, int64_t ld, int K, int flip,
                        int64_t n, const int64_t* d_n, const float* scale, const float* shift, int relu, int32_t* plan,
                        const int32_t* perm, float* dst, float* stats, int32_t* status, hipStream_t s) {
  hipLaunchKernelGGL((k_conv_ring<CS, CD>), dim3(kRG), dim3(64 * (kCW + kLW)), 0, s, src, n_src, wp, pair, ld, K, flip, n, d_n,
                     scale, shift, relu, plan, ring_tcap(n), perm, dst, stats, status);
}

// tiles per wave and turn of the ring of the plans this library deals: 0 = by size (two as soon as one tile per wave would
// need a second turn), 1 / 2 = always that (SPX_RING_TM in the environment, or spx_conv_ring_tiles_per_wave(); tests, A/B runs)
int g_ring_tm = -1;
int ring_tm() {
  if (g_ring_tm < 0) {
    const char* e = getenv("SPX_RING_TM");
    g_ring_tm = e && (e[0] == '1' || e[0] == '2') ? e[0] - '0' : 0;
  }
  return g_ring_tm;
}

}  // namespace

extern "C" size_t spx_conv_ring_plan_bytes(int64_t n_dst) {
  return spx_align((size_t)plan_ints(ring_tcap(n_dst < 0 ? 0 : n_dst)) * sizeof(int32_t));
}

extern "C" int spx_conv_ring_stat_rows(void) { return kRG; }

extern "C" int spx_conv_ring_tiles_per_wave(int set) {
  if (set >= 0 && set <= 2) g_ring_tm = set;
  return ring_tm();
}

extern "C" int spx_conv_ring_plan(const int32_t* pair, int64_t pair_ld, int kvol, int64_t n_dst, const int64_t* d_n_dst,
                                  int32_t* plan, spx_stream_t stream) {
  if (!pair || !plan || kvol <= 0 || kvol > 31 || n_dst <= 0 || pair_ld < n_dst) return SPX_ERR_INVALID_ARG;
  if (n_dst > kMaxRows) return SPX_ERR_TOO_LARGE;
  hipStream_t s = spx_s(stream);
  const int64_t tcap = ring_tcap(n_dst);
  hipLaunchKernelGGL(k_ring_mask, dim3((unsigned)((tcap + 15) / 16)), dim3(256), 0, s, pair, pair_ld, kvol, n_dst, d_n_dst, tcap,
                     plan);
  hipLaunchKernelGGL(k_ring_assign, dim3(8), dim3(1024), 0, s, n_dst, d_n_dst, tcap, kvol, ring_tm(), plan);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

#define SPX_RING_CASE(A, B)                                                                                          \
  if (c_src == A && c_dst == B) {                                                                                    \
    launch_ring<A, B>(src, n_src, w_packed, pair, pair_ld, kvol, flip_k, n_dst, d_n_dst, scale, shift, relu, plan, perm, \
                      dst, stats, d_status, s);                                                                      \
    SPX_CHECK_LAUNCH();                                                                                              \
    return SPX_OK;                                                                                                   \
  }

extern "C" int spx_conv_gemm_ring(const float* src, int64_t n_src, int c_src, const float* w_packed, int c_dst, int kvol,
                                  int flip_k, const int32_t* pair, int64_t pair_ld, int64_t n_dst, const int64_t* d_n_dst,
                                  const float* scale, const float* shift, int relu, int32_t* plan, const int32_t* perm,
                                  float* dst, float* stats, int32_t* d_status, spx_stream_t stream) {
  if (!src || !w_packed || !pair || !dst || !plan || c_src <= 0 || c_dst <= 0 || kvol <= 0 || kvol > 31 || n_dst <= 0 ||
      n_src <= 0 || pair_ld < n_dst)
    return SPX_ERR_INVALID_ARG;
  if (n_dst > kMaxRows) return SPX_ERR_TOO_LARGE;
  // raw buffer descriptors carry 32-bit byte counts; an entry of -1 must land beyond the source records
  if (n_src * (int64_t)c_src * 4 >= (int64_t(1) << 32) - 4096 || (int64_t)kvol * pair_ld * 4 >= (int64_t(1) << 31))
    return SPX_ERR_TOO_LARGE;
  hipStream_t s = spx_s(stream);
  SPX_RING_CASE(64, 64)
  SPX_RING_CASE(32, 64)
  SPX_RING_CASE(64, 32)
  SPX_RING_CASE(32, 32)
  return SPX_ERR_UNSUPPORTED;   // other channel pairs: spx_conv_gemm / spx_conv_gemm_balanced
}
