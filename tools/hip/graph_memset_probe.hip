// graph_memset_probe.hip — dev probe (GPU box): does a byte-pattern hipMemsetAsync NODE of a captured hipGraph re-initialise
// its buffer on every replay?  Round 1 saw an endless CAS probe in the voxeliser on the SECOND replay of a captured graph
// and replaced hipMemsetAsync by a fill kernel (csrc/spx_common.h) without keeping the evidence.  This program captures
//   clear(buf) ; dirty(buf)           (dirty overwrites every word with the replay number)
// once with a memset node and once with a fill-kernel node, replays each graph three times and, after every replay, checks
// through a third node (`snapshot`, placed between clear and dirty) what the clear actually left in the buffer.
//   hipcc --offload-arch=gfx950 -O2 tools/hip/graph_memset_probe.hip -o /tmp/graph_memset_probe && /tmp/graph_memset_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__);     \
      return 2;                                                                \
    }                                                                          \
  } while (0)

__global__ void k_fill(uint32_t* p, uint32_t v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// counts the words that are NOT the cleared pattern
__global__ void k_snapshot(const uint32_t* p, uint32_t expect, size_t n, unsigned long long* bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long c = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != expect;
  if (c) atomicAdd(bad, c);
}
__global__ void k_dirty(uint32_t* p, const uint32_t* tag, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0x01010101u + *tag;
}

static int run(bool use_memset, size_t bytes, int byte_value) {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  uint32_t *buf, *tag;
  unsigned long long* bad;
  const size_t n = bytes / 4;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&tag, 4));
  CK(hipMalloc(&bad, 8));
  CK(hipMemset(buf, 0x33, bytes));
  const uint32_t b = (uint32_t)byte_value & 0xFF, pat = b | (b << 8) | (b << 16) | (b << 24);
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  if (use_memset)
    CK(hipMemsetAsync(buf, byte_value, bytes, s));
  else
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, buf, pat, n);
  hipLaunchKernelGGL(k_snapshot, dim3(1024), dim3(256), 0, s, buf, pat, n, bad);
  hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, s, buf, tag, n);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  int worst = 0;
  for (int rep = 1; rep <= 3; ++rep) {
    uint32_t t = rep;
    CK(hipMemcpyAsync(tag, &t, 4, hipMemcpyHostToDevice, s));
    CK(hipMemsetAsync(bad, 0, 8, s));
    CK(hipGraphLaunch(ge, s));
    unsigned long long h = 0;
    CK(hipMemcpyAsync(&h, bad, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    printf("  %-12s %9zu bytes pattern 0x%02X replay %d: words not cleared when the next node ran = %llu of %zu\n",
           use_memset ? "memset node" : "fill kernel", bytes, byte_value, rep, h, n);
    if (h) worst = 1;
  }
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  CK(hipFree(buf));
  CK(hipFree(tag));
  CK(hipFree(bad));
  CK(hipStreamDestroy(s));
  return worst;
}

int main() {
  int rt = 0;
  CK(hipRuntimeGetVersion(&rt));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("graph_memset_probe: HIP runtime %d, device %s\n", rt, prop.gcnArchName);
  int bad_memset = 0, bad_fill = 0;
  const size_t sizes[] = {4096, 262144 * 8, (size_t)262144 * 5 * 4, (size_t)64 << 20};   // the voxeliser's key / top tables at cfg 2
  const int pats[] = {0xFF, 0x7F, 0x00};
  for (size_t sz : sizes)
    for (int p : pats) {
      bad_memset |= run(true, sz, p);
      bad_fill |= run(false, sz, p);
    }
  printf("RESULT memset nodes %s ; fill-kernel nodes %s\n", bad_memset ? "LEFT STALE WORDS" : "replayed correctly",
         bad_fill ? "LEFT STALE WORDS" : "replayed correctly");
  return 0;
}
