// Debug: how many kernels per second one host thread gets onto a stream, with a ~1 KB by-value argument (like MzrDev)
// and with a pointer argument; and the same sequence replayed from a hipGraph.   hipcc --offload-arch=gfx950 -O2 tools/launch_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { double a[140]; };
__global__ void k_big(Big b, int s, double *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && s < 0) out[0] = b.a[3]; }
__global__ void k_ptr(const Big *b, int s, double *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && s < 0) out[0] = b->a[3]; }
int main() {
  const int n = 20000;
  Big hb; for (int i = 0; i < 140; ++i) hb.a[i] = i;
  Big *db; double *out; hipMalloc(&db, sizeof(Big)); hipMalloc(&out, 8); hipMemcpy(db, &hb, sizeof(Big), hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  for (int rep = 0; rep < 2; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_big, dim3(400), dim3(256), 0, st, hb, i, out);
    auto t1 = std::chrono::steady_clock::now(); hipStreamSynchronize(st); auto t2 = std::chrono::steady_clock::now();
    printf("by value 1120 B : host %.2f us/launch, until done %.2f us/launch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_ptr, dim3(400), dim3(256), 0, st, db, i, out);
    t1 = std::chrono::steady_clock::now(); hipStreamSynchronize(st); t2 = std::chrono::steady_clock::now();
    printf("pointer         : host %.2f us/launch, until done %.2f us/launch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
  }
  // graph of 2000 launches
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_big, dim3(400), dim3(256), 0, st, hb, i, out);
  hipStreamEndCapture(st, &g);
  auto t0 = std::chrono::steady_clock::now();
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  auto t1 = std::chrono::steady_clock::now();
  printf("graph instantiate (2000 nodes): %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
  for (int rep = 0; rep < 3; ++rep) {
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st); t1 = std::chrono::steady_clock::now();
    printf("graph replay    : %.2f us/kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 10000);
  }
  return 0;
}
