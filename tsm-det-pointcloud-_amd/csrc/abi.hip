// abi.hip — library-level entry points of libspx (error strings, ABI version).
#include "spx_common.h"

extern "C" const char* spx_strerror(int code) {
  switch (code) {
    case SPX_OK: return "ok";
    case SPX_ERR_INVALID_ARG: return "invalid argument (null pointer, bad extent, kernel volume > SPX_MAX_KVOL, capacity too small)";
    case SPX_ERR_WORKSPACE: return "workspace missing or smaller than the *_ws_bytes() answer";
    case SPX_ERR_UNSUPPORTED: return "unsupported channel count or mode";
    case SPX_ERR_LAUNCH: return "HIP kernel launch failed";
    case SPX_ERR_TOO_LARGE: return "problem too large (rows >= 2^31 or grid cells >= 2^40)";
    case SPX_ERR_CAPACITY: return "device: more active outputs than the static row capacity of a rule table; rows dropped";
    case SPX_ERR_RING_STALL: return "device: a bounded wait on the weight ring of spx_conv_gemm_ring gave up; output incomplete";
    case SPX_ERR_TABLE_FULL: return "device: hash table full (workspace declared pre-cleared holds stale keys); rows dropped";
    default: return "unknown spx error code";
  }
}

extern "C" int spx_abi_version(void) { return SPX_ABI_VERSION; }

extern "C" int spx_read_status(const int32_t* d_status, spx_stream_t stream) {
  if (!d_status) return SPX_ERR_INVALID_ARG;
  int32_t host = 0;
  hipStream_t s = spx_s(stream);
  if (hipMemcpyAsync(&host, d_status, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess) return SPX_ERR_LAUNCH;
  if (hipStreamSynchronize(s) != hipSuccess) return SPX_ERR_LAUNCH;
  return host > 0 ? SPX_ERR_INVALID_ARG : (int)host;
}
