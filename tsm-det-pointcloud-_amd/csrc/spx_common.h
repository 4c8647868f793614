// spx_common.h — shared device/host helpers for libspx (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/spx.h"

#define SPX_WAVE 64

#define SPX_CHECK_LAUNCH()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return SPX_ERR_LAUNCH; \
  } while (0)

static inline hipStream_t spx_s(spx_stream_t s) { return (hipStream_t)s; }

static inline size_t spx_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// live row count: device pointer if given (clamped to the launch capacity), else the host value
__device__ __forceinline__ int64_t spx_live_n(const int64_t* d_n, int64_t n) {
  if (d_n == nullptr) return n;
  int64_t v = *d_n;
  return v < n ? v : n;
}

__device__ __forceinline__ uint64_t spx_hash64(uint64_t key) { return key * 0x9E3779B97F4A7C15ull; }

struct Int3 {
  int32_t v[3];
};

static inline Int3 spx_i3(const int32_t* p) {
  Int3 r;
  r.v[0] = p[0];
  r.v[1] = p[1];
  r.v[2] = p[2];
  return r;
}

__device__ __forceinline__ int64_t spx_lin_key(int b, int z, int y, int x, const Int3& s) {
  return (((int64_t)b * s.v[0] + z) * s.v[1] + y) * s.v[2] + x;
}

__device__ __forceinline__ int spx_lane() { return threadIdx.x & 63; }

// Device-side fill instead of hipMemsetAsync.  Round 1 saw an endless CAS probe in the voxeliser on the second replay of
// a torch-captured hipGraph while this helper was a byte-pattern hipMemsetAsync, and switched to a fill kernel.  Round 2
// looked for the cause (profiles/r02_graph_memset_probe.log, tools/hip/graph_memset_probe.hip, tools/graph_memset_in_lib.py):
// stand-alone, memset NODES replay correctly on this ROCm for every size / pattern the library uses, so the stall is not
// pinned on them; what made it a HANG rather than a wrong answer were the unbounded probe loops, which are now bounded
// (SPX_ERR_TABLE_FULL).  The fill kernel stays: one code path for eager and captured execution, and -DSPX_FILL_USE_MEMSET
// (dev builds only) restores the memset form for that A/B.  `bytes` must be a multiple of 4 (every buffer here is).
static __global__ void spx_k_fill32(uint32_t* __restrict__ p, uint32_t v, size_t nwords) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nwords; i += stride) p[i] = v;
}

static inline void spx_fill_async(void* ptr, int byte_value, size_t bytes, hipStream_t s) {
  if (bytes == 0) return;
#ifdef SPX_FILL_USE_MEMSET
  (void)hipMemsetAsync(ptr, byte_value, bytes, s);
  return;
#endif
  const uint32_t b = (uint32_t)(byte_value & 0xFF);
  const uint32_t v = b | (b << 8) | (b << 16) | (b << 24);
  const size_t nwords = bytes / 4;
  size_t nb = (nwords + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(spx_k_fill32, dim3((unsigned)nb), dim3(256), 0, s, reinterpret_cast<uint32_t*>(ptr), v, nwords);
}
