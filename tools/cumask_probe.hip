// round 5: do streams with a CU mask (hipExtStreamCreateWithCUMask) keep their kernels on their CUs on this device, and do two
// streams with complementary masks run side by side without delaying each other's launches?
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <set>
#include <chrono>
__global__ void where(int *out, long long spin) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(16);
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 15) << 16 | (int)__builtin_amdgcn_s_getreg((16 - 1) << 11 | 4);
}
__global__ void tiny(int *p) { if (threadIdx.x == 0) p[0] += 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  printf("CUs %d\n", cus);
  const int words = (cus + 31) / 32;
  for (int nSmall : {8, 16, 32}) {
    std::vector<uint32_t> mA(words, 0), mB(words, 0);
    for (int i = 0; i < cus; ++i) (i < nSmall ? mA : mB)[i / 32] |= 1u << (i % 32);
    hipStream_t sA, sB;
    if (hipExtStreamCreateWithCUMask(&sA, words, mA.data()) != hipSuccess || hipExtStreamCreateWithCUMask(&sB, words, mB.data()) != hipSuccess) { printf("stream with CU mask: failed\n"); return 1; }
    int *oA, *oB, *cnt; hipMalloc(&oA, 4096 * 4); hipMalloc(&oB, 65536 * 4); hipMalloc(&cnt, 4); hipMemset(cnt, 0, 4);
    hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, sA, oA, 2000LL);
    hipLaunchKernelGGL(where, dim3(65536), dim3(64), 0, sB, oB, 2000LL);
    hipDeviceSynchronize();
    std::vector<int> hA(4096), hB(65536); hipMemcpy(hA.data(), oA, 4096 * 4, hipMemcpyDeviceToHost); hipMemcpy(hB.data(), oB, 65536 * 4, hipMemcpyDeviceToHost);
    auto key = [](int v) { return ((v >> 16) & 15) * 1000 + ((v >> 13) & 7) * 100 + ((v >> 12) & 1) * 50 + ((v >> 8) & 15); };   // xcc, se, sh, cu
    std::set<int> ca, cb, both; for (int v : hA) ca.insert(key(v)); for (int v : hB) cb.insert(key(v)); for (int k : ca) if (cb.count(k)) both.insert(k);
    printf("mask of %d CUs: small stream ran on %zu distinct (xcc, se, sh, cu), large on %zu, shared %zu\n", nSmall, ca.size(), cb.size(), both.size());
    // a chain of 2000 tiny dependent launches on the small stream: alone, and beside a long kernel that fills the large stream's CUs
    auto chain = [&](bool beside, hipStream_t sSmall) {
      hipDeviceSynchronize();
      if (beside) hipLaunchKernelGGL(where, dim3(65536), dim3(64), 0, sB, oB, 3000000LL / 16);      // ~ 16 waves per CU x ... long enough
      const double t0 = now();
      for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sSmall, cnt);
      hipStreamSynchronize(sSmall);
      const double t1 = now();
      hipDeviceSynchronize();
      return (t1 - t0) * 1e6 / 2000;
    };
    hipStream_t plain; hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
    printf("  2000 dependent tiny launches, us each: masked stream alone %.1f, beside the large stream's kernel %.1f; unmasked stream beside it %.1f\n", chain(false, sA), chain(true, sA), chain(true, plain));
    hipStreamDestroy(sA); hipStreamDestroy(sB); hipStreamDestroy(plain); hipFree(oA); hipFree(oB); hipFree(cnt);
  }
  return 0;
}
