// conv_group.hip — row order for the output-stationary conv kernels: rows grouped by their offset mask.
//
// A 16-row MFMA tile issues the multiplications of offset k when ANY of its rows has a neighbour at k; rows that lack it
// multiply zeros.  In the canonical (b, z, y, x) row order only 44-66 % of the MFMAs issued for the submanifold layers of
// VoxelBackBone8x (reference spconv_backbone.py:99-114) are useful (tools/tile_waste_probe.py); with the rows of a tile
// sharing the same set of offsets it is 78-94 %.  spx_conv_group sorts the destination rows of a rule table by their
// 27-bit offset mask (stable: rows with equal masks keep the canonical order) and writes
//   perm[j]          = table row processed at position j
//   pair_grouped[k][j] = pair[k][perm[j]]          (-1 beyond the live rows)
// Rows are grouped INSIDE EIGHT WINDOWS of the table (window = eighth of the live rows; sort key = window << 27 | mask):
// the balanced kernel renumbers its workgroups so that each XCD walks a contiguous eighth of the work list, so with
// windowed groups every XCD's private L2 gathers from its own eighth of the feature matrix (plus halo) instead of from
// all of it — grouping over the whole table made each of the 8 L2s pull most of the matrix (fetch 2 x 106 MB per 82k-row
// launch against 2 x 39 MB windowed, profiles/r01_conv_experiments.md).
// spx_conv_plan / spx_conv_gemm_balanced then run over pair_grouped and scatter position j to dst row perm[j].  Every
// output row is still the same sum over k in ascending order: values do not depend on the order of the rows.
// The sort itself is rocPRIM's device radix sort (stable, deterministic); the kernels around it are below.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "spx_common.h"

namespace {

constexpr int kGR = 256;          // rows per workgroup
constexpr int kRowInts = 32;      // row-major staging copy: one 128-byte line per row
constexpr uint32_t kDeadKey = 1u << 31;   // sorts after every (window, mask) key
constexpr int kWindows = 8;               // one per XCD
constexpr int kMaskBits = 27;             // kvol <= 27 keeps the window above the mask; larger kernels: one window

// pass 1: key[row] = offset mask (dead rows last) and a row-major copy of the table, written through LDS so that both
// the k-major reads and the row-major writes are coalesced.
__global__ __launch_bounds__(kGR) void k_group_keys(const int32_t* __restrict__ pair, int64_t ld, int K, int64_t n,
                                                    const int64_t* d_n, uint32_t* __restrict__ key,
                                                    int32_t* __restrict__ rowmajor) {
  __shared__ int32_t s[kRowInts][kGR + 1];
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t row0 = (int64_t)blockIdx.x * kGR, row = row0 + threadIdx.x;
  const int64_t wsz = (nlive + kWindows - 1) / kWindows;
  uint32_t m = 0;
  for (int k = 0; k < kRowInts; ++k) {
    int32_t id = -1;
    if (k < K && row < nlive) id = pair[(int64_t)k * ld + row];
    if (id >= 0) m |= 1u << k;
    s[k][threadIdx.x] = id;
  }
  if (K <= kMaskBits && row < nlive) m |= (uint32_t)(row / (wsz > 0 ? wsz : 1)) << kMaskBits;
  if (row < n) key[row] = row < nlive ? m : kDeadKey;
  __syncthreads();
  for (int i = 0; i < kRowInts; ++i) {
    const int e = threadIdx.x + kGR * i, rr = e / kRowInts, c = e % kRowInts;
    if (row0 + rr < n) rowmajor[(row0 + rr) * kRowInts + c] = s[c][rr];
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
  uint32_t *key_in, *key_out;
  int32_t* rowmajor;
  void* temp;
  size_t temp_bytes, total;
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
  L.key_in = reinterpret_cast<uint32_t*>(take((size_t)n * 4));
  L.key_out = reinterpret_cast<uint32_t*>(take((size_t)n * 4));
  L.rowmajor = reinterpret_cast<int32_t*>(take((size_t)n * kRowInts * 4));
  size_t tb = 0;
  // size query only: nothing is dereferenced or launched
  (void)rocprim::radix_sort_pairs(nullptr, tb, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                  rocprim::counting_iterator<int32_t>(0), (int32_t*)nullptr, (size_t)n, 0u, 32u);
  L.temp_bytes = tb;
  L.temp = take(tb ? tb : 1);
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
  if (kvol > 30) return SPX_ERR_UNSUPPORTED;   // the mask is the sort key
  if (n_dst >= (int64_t(1) << 31) / kRowInts) return SPX_ERR_TOO_LARGE;
  GroupWs L = group_layout(ws, n_dst);
  if (!ws || ws_bytes < L.total) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  const unsigned nblk = (unsigned)((n_dst + kGR - 1) / kGR);
  hipLaunchKernelGGL(k_group_keys, dim3(nblk), dim3(kGR), 0, s, pair, pair_ld, kvol, n_dst, d_n_dst, L.key_in, L.rowmajor);
  size_t tb = L.temp_bytes;
  if (rocprim::radix_sort_pairs(L.temp, tb, (const uint32_t*)L.key_in, L.key_out, rocprim::counting_iterator<int32_t>(0), perm,
                                (size_t)n_dst, 0u, 32u, s) != hipSuccess)
    return SPX_ERR_LAUNCH;
  hipLaunchKernelGGL(k_group_permute, dim3(nblk), dim3(kGR), 0, s, L.rowmajor, perm, kvol, n_dst, d_n_dst, n_dst,
                     pair_grouped);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
