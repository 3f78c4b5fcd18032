// Round 5 probe: latency of a dependent load on MI355X by coherence scope -- plain (L1), sc0 (past the CU's L1: the XCD's L2), sc1 (agent scope:
// past the L2 to the memory side), for a footprint that fits one L2 (1 MB) and one that does not (256 MB).  One wavefront chases pointers.
//   hipcc --offload-arch=gfx950 -O2 tools/lat_probe.hip -o /tmp/lat_probe && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
template <int MODE> __device__ __forceinline__ unsigned ld(const unsigned *p) {
  if (MODE == 0) return *p;
  if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (MODE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int MODE> __global__ void chase(const unsigned *buf, int n, unsigned *out, long long *cyc) {
  unsigned i = threadIdx.x == 0 ? 0u : 0u;
  for (int k = 0; k < (n > 100000 ? n / 8 : 64); ++k) i = ld<MODE>(buf + (size_t)i * 16);      // warm the path (the small footprint: one whole walk)
  const long long t0 = clock64();
  for (int k = 0; k < n; ++k) i = ld<MODE>(buf + (size_t)i * 16);
  const long long t1 = clock64();
  if (threadIdx.x == 0) { *out = i; *cyc = t1 - t0; }
}
__global__ void fill(unsigned *buf, const unsigned *next, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i * 16] = next[i];
}
int main() {
  const char *names[4] = {"plain", "workgroup scope (sc0)", "agent scope (sc1)", "system scope (sc0 sc1)"};
  for (size_t lines : {size_t(1) << 14, size_t(1) << 22}) {      // 64-byte lines: 1 MB, 256 MB
    std::vector<unsigned> perm(lines), next(lines);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 g(7); std::shuffle(perm.begin() + 1, perm.end(), g);
    for (size_t k = 0; k < lines; ++k) next[perm[k]] = perm[(k + 1) % lines];
    unsigned *buf, *dn, *out; long long *cyc;
    hipMalloc(&buf, lines * 64); hipMalloc(&dn, lines * 4); hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    hipMemcpy(dn, next.data(), lines * 4, hipMemcpyHostToDevice);
    const int n = lines <= (size_t(1) << 14) ? 8 * (int)lines : 20000;      // the small footprint is walked eight times: seven of them can hit the L2
    for (int mode = 0; mode < 4; ++mode) {
      hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, buf, dn, lines);      // written by other CUs right before
      if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(1), dim3(64), 0, 0, buf, n, out, cyc);
      if (mode == 1) hipLaunchKernelGGL(chase<1>, dim3(1), dim3(64), 0, 0, buf, n, out, cyc);
      if (mode == 2) hipLaunchKernelGGL(chase<2>, dim3(1), dim3(64), 0, 0, buf, n, out, cyc);
      if (mode == 3) hipLaunchKernelGGL(chase<3>, dim3(1), dim3(64), 0, 0, buf, n, out, cyc);
      long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("%4zu MB  %-24s %7.0f cycles per dependent load\n", lines * 64 >> 20, names[mode], (double)c / n);
    }
    hipFree(buf); hipFree(dn); hipFree(out); hipFree(cyc);
  }
  return 0;
}
