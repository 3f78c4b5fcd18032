// Forcing remap in front of basin2reach for a whole window (gfx950): runoff of the hydrologic
// model's own layer (polygon vector or grid) -> runoff of the river-network HRUs.
//
// Replaces  remap_1D_runoff  route/build/src/process_remap.f90:163-261
//           remap_2D_runoff  route/build/src/process_remap.f90:58-157
//           sort_flux        route/build/src/process_remap.f90:268-316
//
// The reference walks the mapping rows with one running cursor into the ragged overlap arrays and
// lets a later row overwrite an earlier one that names the same HRU.  Both are resolved on the host
// once (mzr_set_remap): every destination HRU gets the overlap range of the LAST row that names it,
// with every overlap already translated to a flat source index (-1 = not in the runoff file / grid),
// so the kernel is a gather per (destination HRU, step) whose additions run in the row's own order:
// bit-identical to the reference.  HBM-bound: 8 B written per (HRU, step) + 8 B gathered per overlap
// (weights and indices are re-used across the steps of a tile) + one transpose of the window.
#include "mzr_device.h"

#define RT 8   // steps per lane
// The runoff file is time-major ([step][cell]); a gather of single values from it would use 8 of
// every 64 bytes fetched.  The window is therefore transposed once (coalesced both ways through an
// LDS tile) to [cell][step], ldT = steps rounded up to RT, so that the RT consecutive steps a lane
// needs from one cell are one aligned 64-byte segment.
__global__ void __launch_bounds__(256) k_transpose(int nSteps, int nSrc, int ldT, const double *src, double *srcT) {
  __shared__ double tile[32][33];
  const int c0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int t = t0 + k, c = c0 + tx;
    tile[k][tx] = (t < nSteps && c < nSrc) ? src[(size_t)t * nSrc + c] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, t = t0 + tx;
    if (c < nSrc && t < ldT) srcT[(size_t)c * ldT + t] = tile[tx][k];
  }
}

// grid: x over destination HRUs, y over tiles of RT steps
__global__ void __launch_bounds__(256) k_remap(int H, int nSteps, int ldT, const int *rowStart, const int *rowCnt, const int *srcIdx,
                                               const double *weight, const double *srcT, double *dst) {
  const int hx = blockIdx.x * blockDim.x + threadIdx.x;
  if (hx >= H) return;
  const int t0 = blockIdx.y * RT;
  const int e0 = rowStart[hx], n = rowCnt[hx];     // n < 0: no mapping row names this HRU
  const double xTol = 1.e-6;                        // process_remap.f90:74,181
  double acc[RT], sw[RT];
#pragma unroll
  for (int j = 0; j < RT; ++j) { acc[j] = 0.0; sw[j] = 0.0; }
  for (int e = e0; e < e0 + n; ++e) {
    const int ix = srcIdx[e];
    if (ix < 0) continue;                           // polygon / cell not in the runoff file (:208-211, :108-126)
    const double w = weight[e];
    const double *col = srcT + (size_t)ix * ldT + t0;
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const double v = col[j];                      // steps past the window read the zero padding and are not stored
      if (v > -xTol) { sw[j] = sw[j] + w; acc[j] = acc[j] + w * v; }     // :223-226
    }
  }
#pragma unroll
  for (int j = 0; j < RT; ++j) {
    if (t0 + j >= nSteps) continue;
    double r = acc[j];
    if (sw[j] > xTol) { if (fabs(1.0 - sw[j]) > xTol) r = r / sw[j]; }     // :246-248
    dst[(size_t)(t0 + j) * H + hx] = n < 0 ? 0.0 : r;   // never written by the reference: stays at its initial zero
  }
}

// sort_flux: srcOf[h] = position in the file of the LAST entry that names HRU h, -1 if none
__global__ void __launch_bounds__(256) k_sort_flux(int H, int nSteps, int nSrc, const int *srcOf, int removeNegatives,
                                                   const double *src, double *dst) {
  const int hx = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (hx >= H) return;
  const int ix = srcOf[hx];
  double v = ix < 0 ? -9999.0 : src[(size_t)t * nSrc + ix];   // realMissing where nothing maps (:298)
  if (removeNegatives && v < 0.0) v = 0.0;                      // :312-314
  dst[(size_t)t * H + hx] = v;
}

int mzr_remap_ld(int nSteps) { return (nSteps + RT - 1) / RT * RT; }
// srcT: scratch of nSrc * mzr_remap_ld(nSteps) doubles
void mzr_launch_remap(int H, int nSteps, int nSrc, const int *rowStart, const int *rowCnt, const int *srcIdx,
                      const double *weight, const double *src, double *srcT, double *dst, hipStream_t stream) {
  const int ldT = mzr_remap_ld(nSteps);
  dim3 blockT(256), gridT((nSrc + 31) / 32, (ldT + 31) / 32);
  hipLaunchKernelGGL(k_transpose, gridT, blockT, 0, stream, nSteps, nSrc, ldT, src, srcT);
  dim3 block(256), grid((H + 255) / 256, (nSteps + RT - 1) / RT);
  hipLaunchKernelGGL(k_remap, grid, block, 0, stream, H, nSteps, ldT, rowStart, rowCnt, srcIdx, weight, srcT, dst);
}
void mzr_launch_sort_flux(int H, int nSteps, int nSrc, const int *srcOf, int removeNegatives, const double *src, double *dst,
                          hipStream_t stream) {
  dim3 block(256), grid((H + 255) / 256, nSteps);
  hipLaunchKernelGGL(k_sort_flux, grid, block, 0, stream, H, nSteps, nSrc, srcOf, removeNegatives, src, dst);
}
