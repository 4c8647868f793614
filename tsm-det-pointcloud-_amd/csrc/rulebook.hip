// rulebook.hip — index arithmetic of the sparse-conv hot path (integer work, HBM/latency bound).
//
//  * submanifold rulebook  (SURVEY.md §8a row a7): open-addressing hash of 64-bit linear voxel keys,
//    one probe per (output row, kernel offset), rows of one offset written coalesced, per-offset pair
//    counts by wave ballot + one atomic per wave.
//  * regular / strided rulebook (row a8): NO hash and NO sort.  The output grid is small (stride >= 1
//    of an already sparse grid), so candidates are marked in a BITMAP over output cells; a popcount
//    prefix-sum over the bitmap words gives every active cell its rank, and rank order IS the
//    canonical ascending-linear-key order.  Pair tables are then filled input-driven:
//    rank(o_key(i,k)) is one word load + one popcount.
//
// Replaces the indice-pair builders inside spconv.pytorch.{SubMConv3d,SparseConv3d}.forward, reference
// call sites pcdet/models/backbones_3d/spconv_backbone.py:86-122 (spconv itself is not vendored).
#include "spx_common.h"

namespace {

constexpr uint64_t kEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------ hash (subm)

struct HashTable {
  uint64_t* keys;
  int32_t* vals;
  int log2size;
  int32_t* status;   // device status word (nullable): receives SPX_ERR_TABLE_FULL when a probe sequence finds no end
};

// Every probe sequence is bounded by the slot count.  A table sized by hash_slots() and cleared by the library always has
// >= 50 % empty slots, so the bound is only ever reached on a workspace the caller declared pre-cleared
// (SPX_WS_PRECLEARED) but which holds stale keys: the row is then dropped and the status word says so, instead of a wave
// that spins for ever.
__device__ __forceinline__ void table_full(int32_t* status) {
  if (status) atomicMin(status, (int32_t)SPX_ERR_TABLE_FULL);
}

static inline int64_t hash_slots(int64_t n) {
  int64_t s = 1024;
  while (s < 2 * n) s <<= 1;
  return s;
}
static inline int ilog2(int64_t s) {
  int l = 0;
  while ((int64_t(1) << l) < s) ++l;
  return l;
}

__global__ void k_hash_insert(const int32_t* __restrict__ idx, int64_t n, const int64_t* d_n, int batch, Int3 shape,
                              HashTable t) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= spx_live_n(d_n, n)) return;
  int4 c = reinterpret_cast<const int4*>(idx)[i];
  if ((unsigned)c.x >= (unsigned)batch || (unsigned)c.y >= (unsigned)shape.v[0] ||
      (unsigned)c.z >= (unsigned)shape.v[1] || (unsigned)c.w >= (unsigned)shape.v[2])
    return;  // out-of-grid rows never become neighbours
  uint64_t key = (uint64_t)spx_lin_key(c.x, c.y, c.z, c.w, shape);
  uint64_t mask = (1ull << t.log2size) - 1;
  uint64_t slot = spx_hash64(key) >> (64 - t.log2size);
  for (uint64_t probe = 0; probe <= mask; ++probe) {
    unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&t.keys[slot]), kEmpty, key);
    if (old == kEmpty || old == key) {
      atomicMin(&t.vals[slot], (int32_t)i);  // duplicates: smallest row wins (deterministic)
      return;
    }
    slot = (slot + 1) & mask;
  }
  table_full(t.status);
}

__device__ __forceinline__ int32_t hash_find(const HashTable& t, uint64_t key) {
  uint64_t mask = (1ull << t.log2size) - 1;
  uint64_t slot = spx_hash64(key) >> (64 - t.log2size);
  for (uint64_t probe = 0; probe <= mask; ++probe) {
    uint64_t k = t.keys[slot];
    if (k == key) return t.vals[slot];
    if (k == kEmpty) return -1;
    slot = (slot + 1) & mask;
  }
  table_full(t.status);
  return -1;
}

// grid (ceil(n/256), K): one probe per (row, offset); rows of one offset are contiguous -> coalesced stores
__global__ void k_subm_probe(const int32_t* __restrict__ idx, int64_t n, const int64_t* d_n, Int3 shape, Int3 ks,
                             Int3 dil, HashTable t, int32_t* __restrict__ pair, int64_t ld, int32_t* cnt) {
  int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int k = blockIdx.y;
  int kx = k % ks.v[2], ky = (k / ks.v[2]) % ks.v[1], kz = k / (ks.v[2] * ks.v[1]);
  int32_t r = -1;
  bool live = o < spx_live_n(d_n, n);
  if (live) {
    int4 c = reinterpret_cast<const int4*>(idx)[o];
    int z = c.y + (kz - ks.v[0] / 2) * dil.v[0];
    int y = c.z + (ky - ks.v[1] / 2) * dil.v[1];
    int x = c.w + (kx - ks.v[2] / 2) * dil.v[2];
    if ((unsigned)z < (unsigned)shape.v[0] && (unsigned)y < (unsigned)shape.v[1] &&
        (unsigned)x < (unsigned)shape.v[2])
      r = hash_find(t, (uint64_t)spx_lin_key(c.x, z, y, x, shape));
    pair[(int64_t)k * ld + o] = r;
  }
  if (cnt != nullptr) {
    unsigned long long m = __ballot(r >= 0);
    if (spx_lane() == 0 && m) atomicAdd(&cnt[k], __popcll(m));
  }
}

__device__ __forceinline__ bool in_grid(const int4& c, int batch, const Int3& s);

// Rows known to be UNIQUE (one row per cell: what a voxeliser emits and what spconv requires of SparseConvTensor.indices): a
// submanifold table is symmetric — row r is the neighbour of row o at offset k exactly when o is the neighbour of r at the
// opposite offset K-1-k (odd kernel sizes) — so only the first (K-1)/2 offsets are probed and every hit is written twice.  The
// centre is the row itself.  Halves the random reads of the hash table, which are the whole cost of the kernel (8.1 M probes =
// 0.5 GB of 64-byte sectors at 300 k voxels).  grid (ceil(n/256), (K-1)/2); the upper half of the table is pre-filled with -1.
// A row outside the grid is in no cell: nothing mirrors into it and it mirrors into nothing, it probes both offsets itself.
__global__ void k_subm_probe_sym(const int32_t* __restrict__ idx, int64_t n, const int64_t* d_n, int batch, Int3 shape,
                                 Int3 ks, Int3 dil, HashTable t, int32_t* __restrict__ pair, int64_t ld) {
  const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int K = ks.v[0] * ks.v[1] * ks.v[2];
  const int k = blockIdx.y, km = K - 1 - k;
  const int kx = k % ks.v[2], ky = (k / ks.v[2]) % ks.v[1], kz = k / (ks.v[2] * ks.v[1]);
  if (o >= spx_live_n(d_n, n)) return;
  const int4 c = reinterpret_cast<const int4*>(idx)[o];
  const bool ing = in_grid(c, batch, shape);
  const int dz = (kz - ks.v[0] / 2) * dil.v[0], dy = (ky - ks.v[1] / 2) * dil.v[1], dx = (kx - ks.v[2] / 2) * dil.v[2];
  auto probe = [&](int z, int y, int x) -> int32_t {
    if ((unsigned)z < (unsigned)shape.v[0] && (unsigned)y < (unsigned)shape.v[1] && (unsigned)x < (unsigned)shape.v[2] &&
        (unsigned)c.x < (unsigned)batch)
      return hash_find(t, (uint64_t)spx_lin_key(c.x, z, y, x, shape));
    return -1;
  };
  const int32_t r = probe(c.y + dz, c.z + dy, c.w + dx);
  pair[(int64_t)k * ld + o] = r;
  if (ing) {
    if (r >= 0) pair[(int64_t)km * ld + r] = (int32_t)o;
  } else {
    pair[(int64_t)km * ld + o] = probe(c.y - dz, c.z - dy, c.w - dx);
  }
  if (k == 0) pair[(int64_t)(K / 2) * ld + o] = ing ? (int32_t)o : -1;
}

// ------------------------------------------------------------------------------------ bitmap rank (strided)

struct ConvGeom {
  Int3 in_shape, out_shape, ks, stride, pad, dil;
  Int3 ls;   // log2(stride) per axis, or -1: a run-time integer division is ~40 instructions, and the index kernels
             // below are bound by exactly that arithmetic (every stride of the reference's backbones is 1 or 2)
};

// o = n / stride if divisible (n >= 0), else -1
__device__ __forceinline__ int div_stride(int n, int s, int ls) {
  if (ls >= 0) {
    const int o = n >> ls;
    return (o << ls) == n ? o : -1;
  }
  const int o = n / s;
  return o * s == n ? o : -1;
}

// output coordinate along one axis of (input coordinate c, kernel index k), or -1
__device__ __forceinline__ int cand_axis(int c, int k, const ConvGeom& g, int ax) {
  const int n = c + g.pad.v[ax] - k * g.dil.v[ax];
  if (n < 0) return -1;
  const int o = div_stride(n, g.stride.v[ax], g.ls.v[ax]);
  return o < g.out_shape.v[ax] ? o : -1;
}

// output linear key of (input voxel c, offset k) or -1
__device__ __forceinline__ int64_t cand_key(const int4& c, int k, const ConvGeom& g) {
  int kx = k % g.ks.v[2], ky = (k / g.ks.v[2]) % g.ks.v[1], kz = k / (g.ks.v[2] * g.ks.v[1]);
  const int oz = cand_axis(c.y, kz, g, 0), oy = cand_axis(c.z, ky, g, 1), ox = cand_axis(c.w, kx, g, 2);
  if ((oz | oy | ox) < 0) return -1;
  return spx_lin_key(c.x, oz, oy, ox, g.out_shape);
}

__device__ __forceinline__ bool in_grid(const int4& c, int batch, const Int3& s) {
  return (unsigned)c.x < (unsigned)batch && (unsigned)c.y < (unsigned)s.v[0] && (unsigned)c.z < (unsigned)s.v[1] &&
         (unsigned)c.w < (unsigned)s.v[2];
}

// one thread per (input voxel, kz): the per-axis candidates are tested with shifts, and only the (ky, kx) combinations
// that land on an output cell reach the bitmap (3.4 of 27 on average at stride 2).
// Round 3: the ORs of a block are COMBINED in LDS before they reach memory.  Neighbouring voxels mark the same 64-cell words
// — same-address atomics, which the L2 serialises: 56 us for ~1 M atomics at 300 k voxels, the kernel was bound by exactly
// that (a plain look before the atomic only added a dependent load: measured slower in round 2).  Per (kz, ky) a thread merges
// its kx candidates that share a word and ORs the mask into a block-private open-addressing table (word -> mask, 2 048 slots
// for at most 1 536 insertions: never full, probes bounded by the slot count); afterwards every used slot issues ONE global
// atomic.  Works for any row order: sorted levels (x-neighbours of a row in one wave) and the first level's voxeliser order
// alike.  OR is idempotent and commutative, so the bitmap does not depend on which block or slot carried a bit.
constexpr int kMarkSlots = 2048;
constexpr unsigned long long kMarkEmpty = ~0ull;

__device__ __forceinline__ void mark_insert(unsigned long long* s_key, unsigned long long* s_mask, unsigned long long w,
                                            unsigned long long m, uint64_t* bits) {
  unsigned h = (unsigned)((w * 0x9E3779B97F4A7C15ull) >> 40) & (kMarkSlots - 1);
  for (int probe = 0; probe < kMarkSlots; ++probe) {
    const unsigned long long prev = atomicCAS(&s_key[h], kMarkEmpty, w);
    if (prev == kMarkEmpty || prev == w) {
      atomicOr(&s_mask[h], m);
      return;
    }
    h = (h + 1) & (kMarkSlots - 1);
  }
  atomicOr(reinterpret_cast<unsigned long long*>(&bits[w]), m);     // unreachable (table never full); kept as the safe exit
}

__global__ __launch_bounds__(kBlock) void k_mark(const int32_t* __restrict__ idx, int64_t n, const int64_t* d_n, int batch,
                                                 ConvGeom g, uint64_t* bits) {
  __shared__ unsigned long long s_key[kMarkSlots];
  __shared__ unsigned long long s_mask[kMarkSlots];
  for (int j = threadIdx.x; j < kMarkSlots; j += kBlock) {
    s_key[j] = kMarkEmpty;
    s_mask[j] = 0ull;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = i < spx_live_n(d_n, n);
  int4 c = make_int4(-1, -1, -1, -1);
  if (live) c = reinterpret_cast<const int4*>(idx)[i];
  const int kz = blockIdx.y;
  const int oz = (live && in_grid(c, batch, g.in_shape)) ? cand_axis(c.y, kz, g, 0) : -1;
  if (oz >= 0) {
    for (int ky = 0; ky < g.ks.v[1]; ++ky) {
      const int oy = cand_axis(c.z, ky, g, 1);
      if (oy < 0) continue;
      unsigned long long W = kMarkEmpty, M = 0ull;
      for (int kx = 0; kx < g.ks.v[2]; ++kx) {
        const int ox = cand_axis(c.w, kx, g, 2);
        if (ox < 0) continue;
        const int64_t key = spx_lin_key(c.x, oz, oy, ox, g.out_shape);
        const unsigned long long w = (unsigned long long)(key >> 6), b = 1ull << (key & 63);
        if (W == kMarkEmpty) W = w;
        if (w == W) M |= b;
        else mark_insert(s_key, s_mask, w, b, bits);                  // this row's candidates straddle two words
      }
      if (M != 0ull) mark_insert(s_key, s_mask, W, M, bits);
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < kMarkSlots; j += kBlock) {
    const unsigned long long w = s_key[j];
    if (w != kMarkEmpty) atomicOr(reinterpret_cast<unsigned long long*>(&bits[w]), s_mask[j]);
  }
}

constexpr int kWordsPerThread = 1;   // one 64-cell word per thread: coalesced 8-byte loads, and the small dense grids of the deep levels (2-12 blocks at 8 words per thread, each thread decoding up to 8 x 64 cells serially: 30-48 us) spread over the chip
constexpr int kWordsPerBlock = kBlock * kWordsPerThread;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
  // 256 threads = 4 waves.  wave-level inclusive scan by shuffles, then combine through LDS.
  __shared__ uint32_t wsum[4];
  int lane = spx_lane(), wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return base + inc - v;
}

// exclusive scan of blocksum[0..nblk) in place by ONE block, total -> *d_n_out (+ capacity check)
__device__ __forceinline__ void scan_top_body(uint32_t* blocksum, int64_t nblk, int64_t* d_n_out, int64_t cap,
                                              int32_t* status) {
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nblk; base += kBlock) {
    int64_t j = base + threadIdx.x;
    uint32_t v = j < nblk ? blocksum[j] : 0;
    uint32_t total;
    uint32_t ex = block_exclusive_scan(v, &total);
    uint32_t carry = carry_s;
    if (j < nblk) blocksum[j] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *d_n_out = (int64_t)carry_s;
    // more active cells than the caller's row capacity: rows beyond it are dropped — never silently
    if (status && (int64_t)carry_s > cap) atomicMin(status, (int32_t)SPX_ERR_CAPACITY);
  }
}

__global__ void k_scan_blocksum(const uint64_t* __restrict__ bits, int64_t nwords, uint32_t* blocksum) {
  int64_t w0 = (int64_t)blockIdx.x * kWordsPerBlock + (int64_t)threadIdx.x * kWordsPerThread;
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < kWordsPerThread; ++j)
    if (w0 + j < nwords) s += __popcll(bits[w0 + j]);
  uint32_t total;
  block_exclusive_scan(s, &total);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
}

// single block: exclusive scan of blocksum in place, total -> d_n_out.  (Folding this into k_scan_blocksum through a
// last-arriver ticket was tried: one ticket word drawn by ~3000 blocks serialises at the L2 atomic unit, 15.5 us against
// 5 + 5.6 for the two launches.)
__global__ void k_scan_top(uint32_t* blocksum, int64_t nblk, int64_t* d_n_out, int64_t cap, int32_t* status) {
  scan_top_body(blocksum, nblk, d_n_out, cap, status);
}

__global__ void k_scan_expand(const uint64_t* __restrict__ bits, int64_t nwords, const uint32_t* __restrict__ blocksum,
                              uint32_t* __restrict__ prefix, Int3 out_shape, int32_t* __restrict__ out_idx,
                              int64_t cap) {
  int64_t w0 = (int64_t)blockIdx.x * kWordsPerBlock + (int64_t)threadIdx.x * kWordsPerThread;
  uint64_t wv[kWordsPerThread];
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < kWordsPerThread; ++j) {
    wv[j] = (w0 + j < nwords) ? bits[w0 + j] : 0ull;
    s += __popcll(wv[j]);
  }
  uint32_t run = blocksum[blockIdx.x] + block_exclusive_scan(s, nullptr);
#pragma unroll
  for (int j = 0; j < kWordsPerThread; ++j) {
    if (w0 + j >= nwords) break;
    prefix[w0 + j] = run;
    uint64_t m = wv[j];
    if (m == 0ull) continue;
    // decode the word's first cell ONCE (64-bit divisions are ~100 instructions each on the GPU), then walk the set
    // bits with 32-bit carries: the 64 cells of a word are consecutive in x and wrap into y / z / batch
    int64_t key0 = (w0 + j) << 6;
    int x0, y0, z0, b0;
    if ((nwords << 6) < (int64_t(1) << 31)) {        // grid-uniform: 32-bit divisions (~5x cheaper) when the keys fit
      const uint32_t k32 = (uint32_t)key0;
      const uint32_t sx = (uint32_t)out_shape.v[2], sy = (uint32_t)out_shape.v[1], sz = (uint32_t)out_shape.v[0];
      uint32_t t = k32 / sx;
      x0 = (int)(k32 - t * sx);
      uint32_t t2 = t / sy;
      y0 = (int)(t - t2 * sy);
      uint32_t t3 = t2 / sz;
      z0 = (int)(t2 - t3 * sz);
      b0 = (int)t3;
    } else {
      x0 = (int)(key0 % out_shape.v[2]);
      int64_t t = key0 / out_shape.v[2];
      y0 = (int)(t % out_shape.v[1]);
      t /= out_shape.v[1];
      z0 = (int)(t % out_shape.v[0]);
      b0 = (int)(t / out_shape.v[0]);
    }
    while (m) {
      int b = __ffsll((unsigned long long)m) - 1;
      m &= m - 1;
      if ((int64_t)run < cap) {
        int x = x0 + b, y = y0, z = z0, bb = b0;
        while (x >= out_shape.v[2]) {
          x -= out_shape.v[2];
          if (++y == out_shape.v[1]) {
            y = 0;
            if (++z == out_shape.v[0]) {
              z = 0;
              ++bb;
            }
          }
        }
        reinterpret_cast<int4*>(out_idx)[run] = make_int4(bb, z, y, x);
      }
      ++run;
    }
  }
}

// rank of the active output cell `key` (row of the output tensor), or -1
__device__ __forceinline__ int32_t rank_of(int64_t key, const uint64_t* __restrict__ bits, const uint32_t* __restrict__ prefix,
                                           int64_t cap) {
  const uint64_t w = bits[key >> 6];
  const uint64_t bit = 1ull << (key & 63);
  if (!(w & bit)) return -1;
  const uint32_t r = prefix[key >> 6] + __popcll(w & (bit - 1));
  return (int64_t)r < cap ? (int32_t)r : -1;
}

// pair_fwd[k][0..n_out) = -1 over the LIVE rows (grid-stride: the capacity can be several times the live count)
__global__ void k_fill_neg1(int32_t* pair, int64_t ld, const int64_t* d_n_out) {
  int64_t n = *d_n_out;
  if (n > ld) n = ld;
  for (int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x; o < n; o += (int64_t)gridDim.x * kBlock)
    pair[(int64_t)blockIdx.y * ld + o] = -1;
}

// Both tables of the strided conv, input-driven: pair_bwd[k][i] = output row fed by input i at offset k (coalesced along i)
// and the valid entries of pair_fwd[k][row] = i scattered over the -1 rows.  One thread per (input voxel, kz) walking the
// (ky, kx) plane like k_mark (round 3; before: one thread per (voxel, offset) = 27 x the index loads and three run-time integer
// divisions per thread to split the offset — the arithmetic the index kernels are bound by): the per-axis tests are shared, and
// only the 3.4 of 27 candidates that land on an output cell touch the bitmap.
__global__ void k_conv_pairs(const int32_t* __restrict__ idx, int64_t n, const int64_t* d_n, int batch, ConvGeom g,
                             const uint64_t* __restrict__ bits, const uint32_t* __restrict__ prefix,
                             int32_t* __restrict__ pair_fwd, int64_t cap, int32_t* __restrict__ pair_bwd,
                             int32_t* cnt) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int kz = blockIdx.y;
  const bool live = i < spx_live_n(d_n, n);
  int4 c = make_int4(-1, -1, -1, -1);
  if (live) c = reinterpret_cast<const int4*>(idx)[i];
  const int oz = (live && in_grid(c, batch, g.in_shape)) ? cand_axis(c.y, kz, g, 0) : -1;
  for (int ky = 0; ky < g.ks.v[1]; ++ky) {
    const int oy = oz >= 0 ? cand_axis(c.z, ky, g, 1) : -1;
    for (int kx = 0; kx < g.ks.v[2]; ++kx) {
      const int k = (kz * g.ks.v[1] + ky) * g.ks.v[2] + kx;
      int32_t row = -1;
      if (oy >= 0) {
        const int ox = cand_axis(c.w, kx, g, 2);
        if (ox >= 0) {
          const int64_t key = spx_lin_key(c.x, oz, oy, ox, g.out_shape);
          const uint64_t w = bits[key >> 6];               // the cell is marked by construction: both loads are independent
          const uint32_t r = prefix[key >> 6] + __popcll(w & ((1ull << (key & 63)) - 1));
          if ((int64_t)r < cap) {
            row = (int32_t)r;
            pair_fwd[(int64_t)k * cap + r] = (int32_t)i;
          }
        }
      }
      if (live) pair_bwd[(int64_t)k * n + i] = row;
      if (cnt != nullptr) {
        const unsigned long long m = __ballot(row >= 0);
        if (spx_lane() == 0 && m) atomicAdd(&cnt[k], __popcll(m));
      }
    }
  }
}

// Submanifold table of the OUTPUT level from the same bitmap: its rows are in rank order (= canonical key order), so the
// neighbour at offset k of row o is rank_of(key(coord(o) + (k - centre) * dil)) — one word load + one popcount instead of
// a hash probe, and no hash build at all.  grid (ceil(cap / 256), KZ): a thread resolves the KY x KX neighbours of one kz plane;
// the KX neighbours of a line sit in one bitmap word (two at a word boundary); rows of one offset contiguous -> coalesced stores.
__global__ void k_subm_from_bitmap(const int32_t* __restrict__ out_idx, const int64_t* __restrict__ d_n_out, int64_t cap,
                                   Int3 shape, Int3 ks, Int3 dil, const uint64_t* __restrict__ bits,
                                   const uint32_t* __restrict__ prefix, int32_t* __restrict__ pair, int64_t ld,
                                   int32_t* cnt) {
  const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int kz = blockIdx.y;                 // round 3: a thread walks the KY lines of its kz plane (one index load for all)
  int64_t nlive = *d_n_out;
  if (nlive > cap) nlive = cap;
  const bool live = o < nlive;
  int4 c = make_int4(0, 0, 0, 0);
  if (live) c = reinterpret_cast<const int4*>(out_idx)[o];
  const int z = c.y + (kz - ks.v[0] / 2) * dil.v[0];
  const bool plane_ok = live && (unsigned)z < (unsigned)shape.v[0];
  for (int ky = 0; ky < ks.v[1]; ++ky) {
    const int y = c.z + (ky - ks.v[1] / 2) * dil.v[1];
    const bool line_ok = plane_ok && (unsigned)y < (unsigned)shape.v[1];
    const int64_t line_key = line_ok ? spx_lin_key(c.x, z, y, 0, shape) : 0;
    int64_t w_cached = -1;
    uint64_t w_bits = 0;
    uint32_t w_pre = 0;
    for (int kx = 0; kx < ks.v[2]; ++kx) {
      const int k = (kz * ks.v[1] + ky) * ks.v[2] + kx;
      const int x = c.w + (kx - ks.v[2] / 2) * dil.v[2];
      int32_t r = -1;
      if (line_ok && (unsigned)x < (unsigned)shape.v[2]) {
        const int64_t key = line_key + x;
        if ((key >> 6) != w_cached) {
          w_cached = key >> 6;
          w_bits = bits[w_cached];
          w_pre = prefix[w_cached];
        }
        const uint64_t bit = 1ull << (key & 63);
        if (w_bits & bit) {
          const uint32_t rr = w_pre + __popcll(w_bits & (bit - 1));
          if ((int64_t)rr < cap) r = (int32_t)rr;
        }
      }
      if (live) pair[(int64_t)k * ld + o] = r;
      if (cnt != nullptr) {
        const unsigned long long m = __ballot(r >= 0);
        if (spx_lane() == 0 && m) atomicAdd(&cnt[k], __popcll(m));
      }
    }
  }
}

static inline int64_t cells_of(int batch, const int32_t* s) { return (int64_t)batch * s[0] * s[1] * s[2]; }

}  // namespace

// ------------------------------------------------------------------------------------ C ABI

extern "C" size_t spx_subm_rulebook_ws_bytes(int64_t n) {
  int64_t s = hash_slots(n < 1 ? 1 : n);
  return spx_align((size_t)s * 8) + spx_align((size_t)s * 4);
}

extern "C" int spx_subm_rulebook(const int32_t* idx, int64_t n, const int64_t* d_n, int batch, const int32_t* shape,
                                 const int32_t* ksize, const int32_t* dil, int32_t* pair, int64_t pair_ld,
                                 int32_t* cnt, int flags, int32_t* d_status, void* ws, size_t ws_bytes,
                                 spx_stream_t stream) {
  if ((!idx && n > 0) || !shape || !ksize || !dil || (!pair && n > 0) || n < 0 || batch <= 0 || pair_ld < n ||
      (flags & ~(SPX_WS_PRECLEARED | SPX_ROWS_UNIQUE)))
    return SPX_ERR_INVALID_ARG;
  int K = ksize[0] * ksize[1] * ksize[2];
  if (K <= 0 || K > SPX_MAX_KVOL) return SPX_ERR_INVALID_ARG;
  if (n >= (int64_t(1) << 31)) return SPX_ERR_TOO_LARGE;
  if (!ws || ws_bytes < spx_subm_rulebook_ws_bytes(n)) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  if (cnt) spx_fill_async(cnt, 0, sizeof(int32_t) * K, s);
  if (n == 0) return SPX_OK;
  int64_t slots = hash_slots(n);
  HashTable t;
  t.keys = reinterpret_cast<uint64_t*>(ws);
  t.vals = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + spx_align((size_t)slots * 8));
  t.log2size = ilog2(slots);
  t.status = d_status;
  if (!(flags & SPX_WS_PRECLEARED)) {
    spx_fill_async(t.keys, 0xFF, (size_t)slots * 8, s);
    spx_fill_async(t.vals, 0x7F, (size_t)slots * 4, s);
  }
  unsigned nb = (unsigned)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_hash_insert, dim3(nb), dim3(kBlock), 0, s, idx, n, d_n, batch, spx_i3(shape), t);
  const bool odd = (ksize[0] & 1) && (ksize[1] & 1) && (ksize[2] & 1);
  if ((flags & SPX_ROWS_UNIQUE) && odd && K >= 3 && !cnt) {
    // symmetric form: offsets K/2+1 .. K-1 start as "no pair" and receive the mirrored hits
    spx_fill_async(pair + (int64_t)(K / 2 + 1) * pair_ld, 0xFF, sizeof(int32_t) * (size_t)(K / 2) * (size_t)pair_ld, s);
    hipLaunchKernelGGL(k_subm_probe_sym, dim3(nb, (unsigned)(K / 2)), dim3(kBlock), 0, s, idx, n, d_n, batch, spx_i3(shape),
                       spx_i3(ksize), spx_i3(dil), t, pair, pair_ld);
  } else {
    hipLaunchKernelGGL(k_subm_probe, dim3(nb, K), dim3(kBlock), 0, s, idx, n, d_n, spx_i3(shape), spx_i3(ksize),
                       spx_i3(dil), t, pair, pair_ld, cnt);
  }
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

extern "C" int64_t spx_conv_out_cap(int64_t n_in, int batch, const int32_t* out_shape, const int32_t* ksize,
                                    const int32_t* stride) {
  int64_t per = 1;
  for (int j = 0; j < 3; ++j) per *= (ksize[j] + stride[j] - 1) / stride[j];
  int64_t a = per * n_in, b = cells_of(batch, out_shape);
  int64_t c = a < b ? a : b;
  return c < 1 ? 1 : c;
}

namespace {
struct RbWs {
  uint64_t* bits;       // [nwords]  (+ the scan ticket right behind it: one fill clears both)
  uint32_t* ticket;
  uint32_t* prefix;
  uint32_t* blocksum;
  int64_t nwords, nblk;
  size_t total, clear_bytes;
};
static RbWs rb_layout(void* ws, int batch, const int32_t* out_shape) {
  RbWs r;
  int64_t cells = cells_of(batch, out_shape);
  r.nwords = (cells + 63) / 64;
  r.nblk = (r.nwords + kWordsPerBlock - 1) / kWordsPerBlock;
  char* p = reinterpret_cast<char*>(ws);
  size_t o = 0;
  r.bits = reinterpret_cast<uint64_t*>(p + o);
  r.ticket = reinterpret_cast<uint32_t*>(p + o + (size_t)r.nwords * 8);
  r.clear_bytes = (size_t)r.nwords * 8 + 16;
  o += spx_align(r.clear_bytes);
  r.prefix = reinterpret_cast<uint32_t*>(p + o);
  o += spx_align((size_t)r.nwords * 4);
  r.blocksum = reinterpret_cast<uint32_t*>(p + o);
  o += spx_align((size_t)(r.nblk + 1) * 4);
  r.total = o;
  return r;
}
static inline int log2_or_neg(int v) {
  for (int l = 0; l < 31; ++l)
    if ((1 << l) == v) return l;
  return -1;
}
}  // namespace

extern "C" size_t spx_conv_rulebook_ws_bytes(int64_t n_in, int batch, const int32_t* out_shape) {
  (void)n_in;
  return rb_layout(nullptr, batch, out_shape).total;
}

extern "C" int spx_conv_rulebook(const int32_t* idx, int64_t n_in, const int64_t* d_n_in, int batch,
                                 const int32_t* in_shape, const int32_t* out_shape, const int32_t* ksize,
                                 const int32_t* stride, const int32_t* pad, const int32_t* dil, int32_t* out_idx,
                                 int32_t* pair_fwd, int32_t* pair_bwd, int32_t* cnt, int64_t* d_n_out, int64_t cap,
                                 const int32_t* subm_ksize, const int32_t* subm_dil, int32_t* subm_pair, int32_t* subm_cnt,
                                 int32_t* d_status, void* ws, size_t ws_bytes, spx_stream_t stream) {
  if ((!idx && n_in > 0) || !in_shape || !out_shape || !ksize || !stride || !pad || !dil || !out_idx || !pair_fwd || !pair_bwd ||
      !d_n_out || n_in < 0 || batch <= 0 || cap <= 0)
    return SPX_ERR_INVALID_ARG;
  int K = ksize[0] * ksize[1] * ksize[2];
  if (K <= 0 || K > SPX_MAX_KVOL) return SPX_ERR_INVALID_ARG;
  for (int j = 0; j < 3; ++j) {
    if (stride[j] <= 0 || dil[j] <= 0 || pad[j] < 0 || in_shape[j] <= 0) return SPX_ERR_INVALID_ARG;
    int expect = (in_shape[j] + 2 * pad[j] - dil[j] * (ksize[j] - 1) - 1) / stride[j] + 1;
    if (out_shape[j] != expect || expect <= 0) return SPX_ERR_INVALID_ARG;
  }
  int Ks = 0;
  if (subm_pair) {
    if (!subm_ksize || !subm_dil) return SPX_ERR_INVALID_ARG;
    Ks = subm_ksize[0] * subm_ksize[1] * subm_ksize[2];
    if (Ks <= 0 || Ks > SPX_MAX_KVOL || subm_dil[0] <= 0 || subm_dil[1] <= 0 || subm_dil[2] <= 0) return SPX_ERR_INVALID_ARG;
  }
  if (n_in >= (int64_t(1) << 31) || cap >= (int64_t(1) << 31)) return SPX_ERR_TOO_LARGE;
  if (cells_of(batch, out_shape) >= (int64_t(1) << 40)) return SPX_ERR_TOO_LARGE;
  // cap below spx_conv_out_cap() is allowed (static-capacity / graph mode): rows beyond cap are dropped and the true
  // count still lands in *d_n_out, so the caller detects overflow as *d_n_out > cap.
  if (!ws || ws_bytes < spx_conv_rulebook_ws_bytes(n_in, batch, out_shape)) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  RbWs w = rb_layout(ws, batch, out_shape);
  ConvGeom g;
  g.in_shape = spx_i3(in_shape);
  g.out_shape = spx_i3(out_shape);
  g.ks = spx_i3(ksize);
  g.stride = spx_i3(stride);
  g.pad = spx_i3(pad);
  g.dil = spx_i3(dil);
  for (int j = 0; j < 3; ++j) g.ls.v[j] = log2_or_neg(stride[j]);
  // 8 launches for the strided tables AND the submanifold table of the output level (round 1: 8 + 4):
  //   clear bitmap (+ counters) ; mark ; block sums ; top scan ; ranks + out indices ; -1 over the live rows of the
  //   forward table ; pair tables ; submanifold table (no hash build, no probe kernel for that level)
  if (cnt) spx_fill_async(cnt, 0, sizeof(int32_t) * K, s);
  if (subm_cnt) spx_fill_async(subm_cnt, 0, sizeof(int32_t) * Ks, s);
  spx_fill_async(w.bits, 0, w.clear_bytes, s);
  unsigned nb_in = (unsigned)((n_in + kBlock - 1) / kBlock);
  if (n_in > 0)
    hipLaunchKernelGGL(k_mark, dim3(nb_in, (unsigned)ksize[0]), dim3(kBlock), 0, s, idx, n_in, d_n_in, batch, g, w.bits);
  hipLaunchKernelGGL(k_scan_blocksum, dim3((unsigned)w.nblk), dim3(kBlock), 0, s, w.bits, w.nwords, w.blocksum);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kBlock), 0, s, w.blocksum, w.nblk, d_n_out, cap, d_status);
  hipLaunchKernelGGL(k_scan_expand, dim3((unsigned)w.nblk), dim3(kBlock), 0, s, w.bits, w.nwords, w.blocksum, w.prefix,
                     g.out_shape, out_idx, cap);
  unsigned nb_cap = (unsigned)((cap + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_fill_neg1, dim3(nb_cap > 256 ? 256 : nb_cap, K), dim3(kBlock), 0, s, pair_fwd, cap, d_n_out);
  if (n_in > 0)
    hipLaunchKernelGGL(k_conv_pairs, dim3(nb_in, (unsigned)ksize[0]), dim3(kBlock), 0, s, idx, n_in, d_n_in, batch, g, w.bits, w.prefix,
                       pair_fwd, cap, pair_bwd, cnt);
  if (subm_pair)
    hipLaunchKernelGGL(k_subm_from_bitmap, dim3((unsigned)((cap + kBlock - 1) / kBlock), (unsigned)subm_ksize[0]),
                       dim3(kBlock), 0, s, out_idx,
                       d_n_out, cap, g.out_shape, spx_i3(subm_ksize), spx_i3(subm_dil), w.bits, w.prefix, subm_pair, cap,
                       subm_cnt);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

// ================================================================================================
// Dynamic voxelisation + mean (SURVEY.md §8a row a5'): replaces DynamicMeanVFE.forward, reference
// pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:38-76 — every in-range point is kept (no per-voxel or per-frame
// caps), voxels are the UNIQUE cells sorted by the key ((b*X + cx)*Y + cy)*Z + cz (what torch.unique returns there),
// the feature of a voxel is the mean of its points' columns [xyz_col, xyz_col + C), coords come back as (b, z, y, x).
// No sort: the cells are marked in a bitmap laid out in key order and ranked by the popcount scan used for the strided
// rulebook.  The reference sums with float atomics (torch_scatter); here the points of a voxel are bucketed (counting
// sort), ordered by point index inside the bucket and summed in that order: bitwise reproducible.
namespace {

struct DynGeom {
  float lo[3], vs[3];
  int32_t grid[3];   // x, y, z
};

__device__ __forceinline__ int64_t dyn_key(const float* __restrict__ p, int batch_col, int xyz_col, int batch,
                                           const DynGeom& g) {
  // floor((p - lo) / vs) in fp32, like torch.floor((points[:, 1:4] - range[0:3]) / voxel_size).int()
  int c[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float f = floorf(__fdiv_rn(__fsub_rn(p[xyz_col + j], g.lo[j]), g.vs[j]));
    if (!(f >= 0.f) || !(f < (float)g.grid[j])) return -1;
    c[j] = (int)f;
  }
  const int b = batch_col >= 0 ? (int)p[batch_col] : 0;
  if (b < 0 || b >= batch) return -1;
  return (((int64_t)b * g.grid[0] + c[0]) * g.grid[1] + c[1]) * g.grid[2] + c[2];
}

__global__ void k_dyn_mark(const float* __restrict__ pts, int64_t n, int stride, int batch_col, int xyz_col, int batch,
                           DynGeom g, uint64_t* __restrict__ bits, int64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t key = dyn_key(pts + i * stride, batch_col, xyz_col, batch, g);
  keys[i] = key;
  if (key >= 0) atomicOr(reinterpret_cast<unsigned long long*>(&bits[key >> 6]), 1ull << (key & 63));
}

// rank of every point's voxel (-1 for dropped points) and the number of points per voxel
__global__ void k_dyn_rank(const int64_t* __restrict__ keys, int64_t n, const uint64_t* __restrict__ bits,
                           const uint32_t* __restrict__ prefix, int64_t cap, int32_t* __restrict__ inv,
                           int32_t* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t key = keys[i];
  int32_t rk = -1;
  if (key >= 0) {
    const uint64_t w = bits[key >> 6];
    const int64_t r = (int64_t)prefix[key >> 6] + __popcll(w & ((1ull << (key & 63)) - 1ull));
    if (r < cap) {
      rk = (int32_t)r;
      atomicAdd(&count[rk], 1);
    }
  }
  inv[i] = rk;
}

// single block: offsets = exclusive scan of count[0..nv)
__global__ void k_dyn_offsets(const int32_t* __restrict__ count, const int64_t* __restrict__ d_nv, int64_t cap,
                              int32_t* __restrict__ offset) {
  __shared__ uint32_t carry_s;
  int64_t nv = *d_nv;
  if (nv > cap) nv = cap;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nv; base += kBlock) {
    const int64_t j = base + threadIdx.x;
    const uint32_t v = j < nv ? (uint32_t)count[j] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(v, &total);
    const uint32_t carry = carry_s;
    if (j < nv) offset[j] = (int32_t)(carry + ex);
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
}

__global__ void k_dyn_bucket(const int32_t* __restrict__ inv, int64_t n, const int32_t* __restrict__ offset,
                             int32_t* __restrict__ cursor, int32_t* __restrict__ bucket) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t rk = inv[i];
  if (rk < 0) return;
  bucket[offset[rk] + atomicAdd(&cursor[rk], 1)] = (int32_t)i;
}

// one thread per voxel: order its bucket by point index (insertion sort; buckets are a handful of points), sum in
// that order, divide; coords (b, x, y, z) -> (b, z, y, x)
__global__ void k_dyn_mean(const float* __restrict__ pts, int stride, int xyz_col, int C, const int64_t* __restrict__ d_nv,
                           int64_t cap, const int32_t* __restrict__ offset, const int32_t* __restrict__ count,
                           int32_t* __restrict__ bucket, int32_t* __restrict__ coords, float* __restrict__ feats) {
  const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int64_t nv = *d_nv;
  if (nv > cap) nv = cap;
  if (v >= nv) return;
  int32_t* b = bucket + offset[v];
  const int m = count[v];
  for (int i = 1; i < m; ++i) {
    const int32_t x = b[i];
    int j = i - 1;
    while (j >= 0 && b[j] > x) {
      b[j + 1] = b[j];
      --j;
    }
    b[j + 1] = x;
  }
  const float inv_m = 1.0f / (float)m;
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
    for (int i = 0; i < m; ++i) s += pts[(int64_t)b[i] * stride + xyz_col + c];
    feats[v * C + c] = s * inv_m;
  }
  const int4 o = reinterpret_cast<const int4*>(coords)[v];      // (b, x, y, z) from k_scan_expand
  reinterpret_cast<int4*>(coords)[v] = make_int4(o.x, o.w, o.z, o.y);
}

struct DynWs {
  uint64_t* bits;
  uint32_t *prefix, *blocksum;
  int64_t* keys;
  int32_t *count, *offset, *cursor, *bucket;
  int64_t nwords, nblk;
  size_t total;
};

static DynWs dyn_layout(void* ws, int64_t n, int batch, const int32_t* grid, int64_t cap) {
  DynWs r;
  const int64_t cells = (int64_t)batch * grid[0] * grid[1] * grid[2];
  r.nwords = (cells + 63) / 64;
  r.nblk = (r.nwords + kWordsPerBlock - 1) / kWordsPerBlock;
  char* p = reinterpret_cast<char*>(ws);
  size_t o = 0;
  r.bits = reinterpret_cast<uint64_t*>(p + o);
  o += spx_align((size_t)r.nwords * 8 + 16);          // + the scan ticket
  r.prefix = reinterpret_cast<uint32_t*>(p + o);
  o += spx_align((size_t)r.nwords * 4);
  r.blocksum = reinterpret_cast<uint32_t*>(p + o);
  o += spx_align((size_t)(r.nblk + 1) * 4);
  r.keys = reinterpret_cast<int64_t*>(p + o);
  o += spx_align((size_t)(n > 0 ? n : 1) * 8);
  r.count = reinterpret_cast<int32_t*>(p + o);      // count and cursor are adjacent: one fill clears both
  o += spx_align((size_t)cap * 4);
  r.cursor = reinterpret_cast<int32_t*>(p + o);
  o += spx_align((size_t)cap * 4);
  r.offset = reinterpret_cast<int32_t*>(p + o);
  o += spx_align((size_t)cap * 4);
  r.bucket = reinterpret_cast<int32_t*>(p + o);
  o += spx_align((size_t)(n > 0 ? n : 1) * 4);
  r.total = o;
  return r;
}

}  // namespace

extern "C" size_t spx_dynamic_voxelize_ws_bytes(int64_t n_points, int batch, const int32_t* grid, int64_t cap) {
  if (!grid || batch <= 0 || cap <= 0) return 0;
  return dyn_layout(nullptr, n_points, batch, grid, cap).total;
}

extern "C" int spx_dynamic_voxelize(const float* points, int64_t n_points, int stride, int batch_col, int xyz_col,
                                    int num_features, const float* range6, const float* voxel_size3, const int32_t* grid3,
                                    int batch, float* voxel_features, int32_t* voxel_coords, int32_t* point_to_voxel,
                                    int64_t* d_num_voxels, int64_t cap, void* ws, size_t ws_bytes, spx_stream_t stream) {
  if ((!points && n_points > 0) || !range6 || !voxel_size3 || !grid3 || !voxel_features || !voxel_coords ||
      !point_to_voxel || !d_num_voxels || n_points < 0 || stride <= 0 || xyz_col < 0 || num_features < 3 ||
      xyz_col + num_features > stride || batch_col >= stride || batch <= 0 || cap <= 0)
    return SPX_ERR_INVALID_ARG;
  for (int j = 0; j < 3; ++j)
    if (grid3[j] <= 0 || !(voxel_size3[j] > 0.f)) return SPX_ERR_INVALID_ARG;
  if (n_points >= (int64_t(1) << 31) || cap >= (int64_t(1) << 31)) return SPX_ERR_TOO_LARGE;
  if ((int64_t)batch * grid3[0] * grid3[1] * grid3[2] >= (int64_t(1) << 40)) return SPX_ERR_TOO_LARGE;
  if (!ws || ws_bytes < spx_dynamic_voxelize_ws_bytes(n_points, batch, grid3, cap)) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  DynWs w = dyn_layout(ws, n_points, batch, grid3, cap);
  DynGeom g;
  for (int j = 0; j < 3; ++j) {
    g.lo[j] = range6[j];
    g.vs[j] = voxel_size3[j];
    g.grid[j] = grid3[j];
  }
  spx_fill_async(w.bits, 0, (size_t)w.nwords * 8 + 16, s);                     // bitmap + scan ticket
  spx_fill_async(w.count, 0, (size_t)((char*)w.offset - (char*)w.count), s);   // count + cursor
  const unsigned nbp = (unsigned)((n_points + kBlock - 1) / kBlock);
  if (n_points > 0)
    hipLaunchKernelGGL(k_dyn_mark, dim3(nbp), dim3(kBlock), 0, s, points, n_points, stride, batch_col, xyz_col, batch, g,
                       w.bits, w.keys);
  hipLaunchKernelGGL(k_scan_blocksum, dim3((unsigned)w.nblk), dim3(kBlock), 0, s, w.bits, w.nwords, w.blocksum);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kBlock), 0, s, w.blocksum, w.nblk, d_num_voxels, (int64_t(1) << 62),
                     (int32_t*)nullptr);
  Int3 shape;
  shape.v[0] = grid3[0], shape.v[1] = grid3[1], shape.v[2] = grid3[2];
  hipLaunchKernelGGL(k_scan_expand, dim3((unsigned)w.nblk), dim3(kBlock), 0, s, w.bits, w.nwords, w.blocksum, w.prefix, shape,
                     voxel_coords, cap);
  if (n_points > 0) {
    hipLaunchKernelGGL(k_dyn_rank, dim3(nbp), dim3(kBlock), 0, s, w.keys, n_points, w.bits, w.prefix, cap, point_to_voxel,
                       w.count);
    hipLaunchKernelGGL(k_dyn_offsets, dim3(1), dim3(kBlock), 0, s, w.count, d_num_voxels, cap, w.offset);
    hipLaunchKernelGGL(k_dyn_bucket, dim3(nbp), dim3(kBlock), 0, s, point_to_voxel, n_points, w.offset, w.cursor, w.bucket);
    hipLaunchKernelGGL(k_dyn_mean, dim3((unsigned)((cap + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, points, stride, xyz_col,
                       num_features, d_num_voxels, cap, w.offset, w.count, w.bucket, voxel_coords, voxel_features);
  }
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
