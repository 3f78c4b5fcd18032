// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths of this library
// (MI355X_MICROARCH.md, HBM section: only 16-B/lane streaming reads are calibrated there).
//   k_copy8     one double per lane, contiguous: reads n*8 B, writes n*8 B
//   k_rows8     16 lanes read one 160-byte row each at a scattered row index and write 8 B per row:
//               the particle-row pattern of the KWT sweep; reads rows*160 B (+ 4 B index), writes rows*8 B
//   k_copy16    two doubles per lane (the guide's calibrated case, for reference)
// Buffers are 2 GiB: past the 256 MiB Infinity Cache.  Build and run: tools/profile_bundle.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_copy8(const double *a, double *b, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }
__global__ void k_copy16(const double2 *a, double2 *b, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }
__global__ void k_rows8(const double *a, const int *row, double *out, size_t rows) {
  size_t g = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) / 16; int l = threadIdx.x & 15;
  if (g >= rows) return;
  const double *p = a + (size_t)row[g] * 20;
  double v = p[l] + (l < 4 ? p[16 + l] : 0.0);
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
  if (l == 0) out[g] = v;
}
int main() {
  const size_t n = (size_t)1 << 28;   // 2 GiB of doubles
  double *a, *b; int *row;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
  hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
  const size_t rows = n / 20;
  std::vector<int> h(rows);
  unsigned long long s = 12345;
  for (size_t i = 0; i < rows; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = (int)((s >> 33) % rows); }
  hipMalloc(&row, rows * 4); hipMemcpy(row, h.data(), rows * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_copy8, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_copy16, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, n / 2);
    hipLaunchKernelGGL(k_rows8, dim3((rows * 16 + 255) / 256), dim3(256), 0, 0, a, row, b, rows);
  }
  hipDeviceSynchronize();
  printf("copy8 read %zu write %zu | copy16 read %zu write %zu | rows8 read %zu write %zu\n", n * 8, n * 8, n * 8, n * 8, rows * 164, rows * 8);
  return 0;
}
