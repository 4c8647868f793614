// conv_group.hip — row order for the output-stationary conv kernels: rows grouped by their offset pattern.
//
// A 16-row MFMA tile issues the multiplications of offset k when ANY of its rows has a neighbour at k; rows that lack it
// multiply zeros.  In the canonical (b, z, y, x) row order only 44-66 % of the MFMAs issued for the submanifold layers of
// VoxelBackBone8x (reference spconv_backbone.py:99-114) are useful (tools/tile_waste_probe.py); with the rows of a tile
// sharing the same set of offsets it is 74-94 %.  spx_conv_group orders the destination rows of a rule table by
//     (window, group key)        stable: rows with equal keys keep the canonical order
// and writes
//   perm[j]            = table row processed at position j
//   pair_grouped[k][j] = pair[k][perm[j]]          (-1 beyond the live rows)
// spx_conv_plan / spx_conv_gemm_balanced then run over pair_grouped and scatter position j to dst row perm[j].  Every
// output row is still the same sum over k in ascending order: values do not depend on the order of the rows.
//   * group key (11 bits for the 3x3x3 kernel): the 9 in-plane offsets bit by bit + "any offset in the plane below" +
//     "any in the plane above" — keeps the effect of sorting by the full 27-bit mask (20 712 vs 23 365 -> see DESIGN.md)
//     and fits ONE counting-sort pass;
//   * windows: contiguous row ranges of <= 4 096 rows, at least eight of them (a multiple of eight: the XCD count): the balanced kernel walks
//     its work list XCD by XCD in contiguous eighths, so with windowed groups every XCD's private L2 gathers from its own
//     part of the feature matrix instead of from all of it (fetch 2 x 106 MB -> 2 x 36 MB per 82 k-row launch).
// The sort is hand-written (round 1 used rocPRIM's radix sort: 59 of the call's 77 us, ten launches): ONE workgroup per
// window sorts its rows entirely in LDS — per-wave histograms over contiguous row segments, one prefix over
// (key, wave), stable placement with wave-level peer matching.  Deterministic: no float, no order-dependent atomic.
#include "spx_common.h"

namespace {

constexpr int kGR = 256;          // rows per workgroup of the staging / permute kernels
constexpr int kRowInts = 32;      // row-major staging copy: one 128-byte line per row
constexpr int kKeyBits = 11;
constexpr int kBins = 1 << kKeyBits;
constexpr int kSortWaves = 8;     // waves of the sorting workgroup (per-wave histograms: 8 x 2048 x 4 B = 64 KiB of LDS)
constexpr int kWinRows = 4096;    // rows per window at most (their 16-bit keys: 8 KiB of LDS)
constexpr uint32_t kDead = 0xFFFFu;

// 11-bit group key of an offset mask
__device__ __forceinline__ uint32_t group_key(uint32_t m, int K) {
  if (K <= kKeyBits) return m;
  if (K % 3 == 0 && K / 3 <= 9) {
    const int P = K / 3;
    const uint32_t low = m & ((1u << P) - 1u), mid = (m >> P) & ((1u << P) - 1u), up = m >> (2 * P);
    return mid | (low ? 1u << 9 : 0u) | (up ? 1u << 10 : 0u);
  }
  return m >> (K - kKeyBits);
}

// windows: W = max(8, ceil(nlive / kWinRows)) rounded up to a multiple of 8, equal sizes
__host__ __device__ inline int64_t window_count(int64_t nlive) {
  int64_t w = (nlive + kWinRows - 1) / kWinRows;
  if (w < 8) w = 8;
  return (w + 7) / 8 * 8;
}
__host__ __device__ inline int64_t window_rows(int64_t nlive) {
  const int64_t w = window_count(nlive);
  return (nlive + w - 1) / w;
}

// pass 1: key16[row] = group key of the row's offset mask and a row-major copy of the table, written through LDS so that
// both the k-major reads and the row-major writes are coalesced.
__global__ __launch_bounds__(kGR) void k_group_keys(const int32_t* __restrict__ pair, int64_t ld, int K, int64_t n,
                                                    const int64_t* d_n, uint16_t* __restrict__ key,
                                                    int32_t* __restrict__ rowmajor) {
  __shared__ int32_t s[kRowInts][kGR + 1];
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t row0 = (int64_t)blockIdx.x * kGR, row = row0 + threadIdx.x;
  uint32_t m = 0;
  for (int k = 0; k < kRowInts; ++k) {
    int32_t id = -1;
    if (k < K && row < nlive) id = pair[(int64_t)k * ld + row];
    if (id >= 0) m |= 1u << k;
    s[k][threadIdx.x] = id;
  }
  if (row < n) key[row] = row < nlive ? (uint16_t)group_key(m, K) : (uint16_t)kDead;
  __syncthreads();
  for (int i = 0; i < kRowInts; ++i) {
    const int e = threadIdx.x + kGR * i, rr = e / kRowInts, c = e % kRowInts;
    if (row0 + rr < n) rowmajor[(row0 + rr) * kRowInts + c] = s[c][rr];
  }
}

// pass 2: one workgroup per window: stable counting sort of the window's rows by key, in LDS
__global__ __launch_bounds__(64 * kSortWaves) void k_group_sort(const uint16_t* __restrict__ key, int64_t n,
                                                                const int64_t* d_n, int32_t* __restrict__ perm) {
  __shared__ uint32_t s_hist[kSortWaves][kBins];       // per-wave counts, then per-wave write cursors
  __shared__ uint16_t s_key[kWinRows];
  __shared__ uint32_t s_scan[64 * kSortWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t W = window_count(nlive), wsz = window_rows(nlive);
  // rows beyond the live count keep their place (identity): the first workgroups write them
  for (int64_t j = nlive + (int64_t)blockIdx.x * (64 * kSortWaves) + tid; j < n; j += (int64_t)gridDim.x * (64 * kSortWaves))
    perm[j] = (int32_t)j;
  if ((int64_t)blockIdx.x >= W) return;
  const int64_t r0 = (int64_t)blockIdx.x * wsz;
  int64_t r1 = r0 + wsz;
  if (r1 > nlive) r1 = nlive;
  const int cnt = r1 > r0 ? (int)(r1 - r0) : 0;           // <= kWinRows by construction
  for (int i = tid; i < kSortWaves * kBins; i += 64 * kSortWaves) (&s_hist[0][0])[i] = 0u;
  __syncthreads();
  // every wave owns a contiguous segment of the window (multiples of 64 rows): counts of its segment
  const int seg = ((cnt + kSortWaves - 1) / kSortWaves + 63) / 64 * 64;
  const int a0 = wave * seg < cnt ? wave * seg : cnt, a1 = (wave + 1) * seg < cnt ? (wave + 1) * seg : cnt;
  for (int i = a0 + lane; i < a1; i += 64 * 8) {          // eight independent loads in flight per lane
    uint32_t kk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) kk[u] = i + 64 * u < a1 ? (uint32_t)key[r0 + i + 64 * u] : kDead;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i + 64 * u < a1) {
        s_key[i + 64 * u] = (uint16_t)kk[u];
        atomicAdd(&s_hist[wave][kk[u] & (kBins - 1)], 1u);   // integer counts: order-independent
      }
  }
  __syncthreads();
  // exclusive prefix over (key, wave): thread t owns keys [4t, 4t + 4) (kBins = 4 x 512)
  constexpr int kPer = kBins / (64 * kSortWaves);
  uint32_t tot = 0;
#pragma unroll
  for (int b = 0; b < kPer; ++b)
    for (int w = 0; w < kSortWaves; ++w) tot += s_hist[w][tid * kPer + b];
  // block-wide exclusive scan of tot
  uint32_t incl = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  uint32_t base = incl - tot;
  for (int w = 0; w < wave; ++w) base += s_scan[w];
  __syncthreads();
#pragma unroll
  for (int b = 0; b < kPer; ++b)
    for (int w = 0; w < kSortWaves; ++w) {
      const uint32_t c = s_hist[w][tid * kPer + b];
      s_hist[w][tid * kPer + b] = base;                   // cursor of (key, wave)
      base += c;
    }
  __syncthreads();
  // stable placement: a wave walks its segment in order, 64 rows at a time.  Rank of a row among the rows of its group
  // with the same key = equal keys in lower lanes (64 readlane compares, no loop over distinct keys: canonical row order
  // puts up to 64 different keys into one group); the highest lane of every key advances that key's cursor.
  uint32_t* cur = s_hist[wave];
  for (int i0 = a0; i0 < a1; i0 += 64) {
    const int i = i0 + lane;
    const bool live = i < a1;
    const uint32_t kk = live ? (uint32_t)s_key[i] & (kBins - 1) : 0xFFFFFFFFu;
    uint32_t below = 0;
    bool above = false;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)kk, j);
      const bool eq = kj == kk;
      below += (eq && j < lane) ? 1u : 0u;
      above |= eq && j > lane;
    }
    uint32_t pos = 0;
    if (live) pos = cur[kk] + below;
    __builtin_amdgcn_wave_barrier();
    if (live && !above) cur[kk] = pos + 1;                   // one lane per distinct key
    __builtin_amdgcn_wave_barrier();
    if (live) perm[r0 + pos] = (int32_t)(r0 + i);
  }
}

// pass 3: pair_grouped[k][j] = rowmajor[perm[j]][k]; each thread pulls its row's line (eight 16-byte pieces), the writes
// are coalesced along j.
__global__ __launch_bounds__(kGR) void k_group_permute(const int32_t* __restrict__ rowmajor, const int32_t* __restrict__ perm,
                                                       int K, int64_t n, const int64_t* d_n, int64_t ld_out,
                                                       int32_t* __restrict__ grouped) {
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t j = (int64_t)blockIdx.x * kGR + threadIdx.x;
  if (j >= n) return;
  int32_t v[kRowInts];
  if (j < nlive) {
    const int4* line = reinterpret_cast<const int4*>(rowmajor + (int64_t)perm[j] * kRowInts);
#pragma unroll
    for (int i = 0; i < kRowInts / 4; ++i) {
      const int4 t = line[i];
      v[4 * i] = t.x, v[4 * i + 1] = t.y, v[4 * i + 2] = t.z, v[4 * i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < kRowInts; ++i) v[i] = -1;
  }
#pragma unroll
  for (int k = 0; k < kRowInts; ++k)
    if (k < K) grouped[(int64_t)k * ld_out + j] = v[k];
}

struct GroupWs {
  uint16_t* key;
  int32_t* rowmajor;
  size_t total;
};

static GroupWs group_layout(void* ws, int64_t n) {
  GroupWs L{};
  size_t off = 0;
  char* base = reinterpret_cast<char*>(ws);
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += spx_align(bytes);
    return p;
  };
  L.key = reinterpret_cast<uint16_t*>(take((size_t)n * 2));
  L.rowmajor = reinterpret_cast<int32_t*>(take((size_t)n * kRowInts * 4));
  L.total = off;
  return L;
}

}  // namespace

extern "C" size_t spx_conv_group_ws_bytes(int64_t n_dst) {
  if (n_dst <= 0) return 0;
  return group_layout(nullptr, n_dst).total;
}

extern "C" int spx_conv_group(const int32_t* pair, int64_t pair_ld, int kvol, int64_t n_dst, const int64_t* d_n_dst,
                              int32_t* perm, int32_t* pair_grouped, void* ws, size_t ws_bytes, spx_stream_t stream) {
  if (!pair || !perm || !pair_grouped || kvol <= 0 || n_dst <= 0 || pair_ld < n_dst) return SPX_ERR_INVALID_ARG;
  if (kvol > 30) return SPX_ERR_UNSUPPORTED;
  if (n_dst >= (int64_t(1) << 31) / kRowInts) return SPX_ERR_TOO_LARGE;
  GroupWs L = group_layout(ws, n_dst);
  if (!ws || ws_bytes < L.total) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  const unsigned nblk = (unsigned)((n_dst + kGR - 1) / kGR);
  hipLaunchKernelGGL(k_group_keys, dim3(nblk), dim3(kGR), 0, s, pair, pair_ld, kvol, n_dst, d_n_dst, L.key, L.rowmajor);
  // the window count follows the LIVE row count (device side); the grid covers the windows of the capacity
  hipLaunchKernelGGL(k_group_sort, dim3((unsigned)window_count(n_dst)), dim3(64 * kSortWaves), 0, s, L.key, n_dst, d_n_dst,
                     perm);
  hipLaunchKernelGGL(k_group_permute, dim3(nblk), dim3(kGR), 0, s, L.rowmajor, perm, kvol, n_dst, d_n_dst, n_dst,
                     pair_grouped);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
