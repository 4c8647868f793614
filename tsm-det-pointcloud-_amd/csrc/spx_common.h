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
