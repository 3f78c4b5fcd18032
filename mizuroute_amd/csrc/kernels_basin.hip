// HRU -> reach mapping and hillslope unit-hydrograph delay for a whole time window (gfx950).
//
// Replaces, for W time steps at once:
//   basin2reach                      route/build/src/process_remap.f90:319-422
//   IRF_route_basin/hru_irf/irf_conv route/build/src/basinUH.f90:19-178
//
// The reference keeps a per-reach shift register QFUTURE(1:n) and, every step, adds
// FRAC_FUTURE(j)*BASIN_QI to slot j, emits slot 1 and shifts.  Neither step depends on the river
// network, so the whole window is computed before the routing sweep, fully parallel over
// (reach, step).  The value emitted at step t entered the register at slot t-tau+1 at step tau, so
//   BASIN_QR(1)(t) = (((S0(t) + F(t)*QI(0)) + F(t-1)*QI(1)) + ... ) + F(0)*QI(t)     t <  n
//                  = (((F(n-1)*QI(t-n+1)) + F(n-2)*QI(t-n+2)) + ... ) + F(0)*QI(t)   t >= n
// with the additions in exactly the order the shift register performs them (oldest first), hence
// bit-identical to the reference; the register contents after the window follow from the same
// fold for the virtual output times W..W+n-2.
#include "mzr_device.h"

// A block = 256 reaches x a tile of BT steps of the window (the HRU list of a reach is read once per tile).  The lanes GATHER
// from the tile's runoff rows (HRUs in the caller's order), so every row of a tile is needed whole by every block of the tile:
// round 4 maps the blocks so that ALL blocks of a tile run on ONE XCD (workgroups go to the XCDs round robin by their linear
// index: block L sits on XCD L mod 8) and the tile's rows (BT x 8 H bytes: 3.2 MB at 100 k HRUs) stay in that XCD's 4 MB L2.
// With x = reach block, y = tile as the launch grid, the blocks of a tile were dealt to all eight XCDs and each L2 fetched
// every row again: 62.7 GB fetched for 13 GB of forcing (profiles/r04b_summary.md).
// grid: 1-D, 8 * ceil(tiles / 8) * nReachBlocks blocks.
#define BT 4
__global__ void __launch_bounds__(256) k_basin2reach(MzrDev d, int tBegin, int tEnd, int nRB) {
  if (d.err->code != 0) return;      // an earlier window of the queue failed: its successors leave everything as it is (mzr_sync may route them again)
  const int L = blockIdx.x, xcd = L & 7, m = L >> 3;
  const int tile = (m / nRB) * 8 + xcd, rb = m % nRB;
  const int r = rb * blockDim.x + threadIdx.x;
  const int t0 = tBegin + tile * BT;
  if (t0 >= tEnd) return;
  if (r >= d.N) return;
  if (d.haloSlot && d.haloSlot[r] >= 0) return;     // lateral inflow of a halo reach is imported
  const int e0 = d.hruOff[r], e1 = d.hruOff[r + 1];
  const double area = d.basarea[r];
  double acc[BT];
#pragma unroll
  for (int j = 0; j < BT; ++j) acc[j] = 0.0;
  for (int e = e0; e < e1; ++e) {
    const int hx = d.hruIdx[e];
    const double w = d.hruW[e];
#pragma unroll
    for (int j = 0; j < BT; ++j) {
      if (t0 + j < tEnd) {
        const double v = d.runoff[(size_t)(t0 + j) * d.H + hx];
        if (v < d.negRunoffTol) mzr_raise(d, 20, r, t0 + j, 1);   // process_remap.f90:397-402
        acc[j] = acc[j] + w * v * d.time_conv * d.length_conv;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < BT; ++j) {
    const int t = t0 + j;
    if (t >= tEnd) continue;
    double rr;
    if (e1 > e0) {
      double a = acc[j];
      if (a < d.runoffMin) a = d.runoffMin;
      rr = a * area;
    } else {
      rr = d.runoffMin;
    }
    if (d.doesBasinRoute == 1) d.qi[(size_t)t * d.N + r] = rr;
    else d.qlat[(size_t)(t + 1) * d.N + r] = rr;            // main_route.f90:223-226
  }
}

// Register-tiled fold: one lane produces HT consecutive outputs of one reach, so every BASIN_QI
// value is loaded once per HT outputs instead of once per output (the convolution has ~200 taps at
// dt = 1 h).  The coefficients sit in a copy zero-padded by HT on both sides (fracPad) and are read
// with a wave-uniform address: tap k of output j at input tau is F[first + j - tau], outside [0, n) it is
// 0 and the (exact) addition of 0*q stands in for the reference's "no contribution yet".  `first`
// is the step (k_hillslope_out) or virtual step W+j (k_hillslope_state) of output 0; contributions
// are still added oldest-first per output, so results are unchanged.
#define HT 32
// wave-uniform reads through the scalar cache: a pointer into the constant address space makes the
// compiler issue s_load for a uniform address, so the coefficients arrive in SGPRs and cost neither
// LDS bandwidth (one LDS serves four SIMDs) nor vector registers
typedef const double __attribute__((address_space(4))) *mzr_cptr;
template <bool STATE>
__device__ __forceinline__ void hillslope_tile(const MzrDev &d, mzr_cptr Fpad, int r, int first, int count, bool active,
                                               const double *in, double *out, const double *S0, double *S1, double *headQ = nullptr) {
  const int n = d.ntdhBas, N = d.N, W = d.W;
  if (!active) return;
  const bool lake = d.lakeSlot && d.lakeSlot[r] >= 0;   // lakes: impulse response, basinUH.f90:116-119
  double acc[HT];
#pragma unroll
  for (int j = 0; j < HT; ++j) { const int tv = first + j; acc[j] = (j < count && tv < n) ? S0[(size_t)tv * N + r] : 0.0; }
  const int tauHi = STATE ? W - 1 : first + count - 1;          // newest input any output of the tile sees
  int tauLo = first - n + 1; if (tauLo < 0) tauLo = 0;           // oldest input of output 0
  if (!lake) {
    for (int tau = tauLo; tau <= tauHi; ++tau) {
      const double q = in[(size_t)tau * N + r];
      mzr_cptr F = Fpad + HT + (first - tau);                    // F[j] = tap of output j (wave-uniform address)
#pragma unroll
      for (int j = 0; j < HT; ++j) acc[j] = acc[j] + F[j] * q;
    }
  } else if (!STATE) {
#pragma unroll
    for (int j = 0; j < HT; ++j) if (j < count) acc[j] = acc[j] + 1.0 * in[(size_t)(first + j) * N + r];
  }
#pragma unroll
  for (int j = 0; j < HT; ++j) {
    if (j >= count) continue;
    if (STATE) S1[(size_t)(first - W + j) * N + r] = acc[j];
    else out[(size_t)(first + j + 1) * N + r] = acc[j];
  }
  // a headwater reach's KWT discharge of the step IS this value (kwt_route.f90:181-205): written here, the rows need not be
  // read back and copied (k_kwt_window_init: 13 GB read + 6.5 GB written per 16 384-step window at 100 k reaches, 8.6 ms)
  if (!STATE && headQ && d.kwHeadFlag[r]) {
#pragma unroll
    for (int j = 0; j < HT; ++j) if (j < count) headQ[(size_t)(first + j) * N + r] = acc[j];
  }
}

// grid: x over reaches, y over tiles of HT steps: BASIN_QR(1) of steps [y*HT, y*HT+HT)
__global__ void __launch_bounds__(256) k_hillslope_out(MzrDev d, int tBegin, int tEnd) {
  if (d.err->code != 0) return;
  const mzr_cptr Fpad = (mzr_cptr)(unsigned long long)d.fracPad;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = r < d.N && !(d.haloSlot && d.haloSlot[r] >= 0);
  const int t0 = tBegin + blockIdx.y * HT;
  const int count = tEnd - t0 < HT ? tEnd - t0 : HT;
  hillslope_tile<false>(d, Fpad, r, t0, count, active, d.qi, d.qlat, d.basS0, d.basS1, d.kwHeadQ);
}

// grid: x over reaches, y over tiles of HT register slots: QFUTURE(j+1) after the window
__global__ void __launch_bounds__(256) k_hillslope_state(MzrDev d) {
  if (d.err->code != 0) return;
  const mzr_cptr Fpad = (mzr_cptr)(unsigned long long)d.fracPad;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int j0 = blockIdx.y * HT;
  const int count = d.ntdhBas - j0 < HT ? d.ntdhBas - j0 : HT;
  hillslope_tile<true>(d, Fpad, r, d.W + j0, count, r < d.N, d.qi, d.qlat, d.basS0, d.basS1);
}

// ---- constituent (tracer): the same two kernels on BASIN_solute_inst / solute_future (basinUH.f90:130-137)
__global__ void __launch_bounds__(256) k_hillslope_out_solute(MzrDev d, int tBegin, int tEnd) {
  const mzr_cptr Fpad = (mzr_cptr)(unsigned long long)d.fracPad;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int t0 = tBegin + blockIdx.y * HT;
  const int count = tEnd - t0 < HT ? tEnd - t0 : HT;
  hillslope_tile<false>(d, Fpad, r, t0, count, r < d.N, d.solInst, d.basSol, d.solS0, d.solS1);
}
__global__ void __launch_bounds__(256) k_hillslope_state_solute(MzrDev d) {
  const mzr_cptr Fpad = (mzr_cptr)(unsigned long long)d.fracPad;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int j0 = blockIdx.y * HT;
  const int count = d.ntdhBas - j0 < HT ? d.ntdhBas - j0 : HT;
  hillslope_tile<true>(d, Fpad, r, d.W + j0, count, r < d.N, d.solInst, d.basSol, d.solS0, d.solS1);
}
// basin2reach_mass (process_remap.f90:425-500) and the gates of main_route.f90:207-213 / :228-236: a reach whose runoff
// of the step is zero drops its constituent.  grid: x over reaches, y over steps.
__global__ void __launch_bounds__(256) k_basin2reach_mass(MzrDev d) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (r >= d.N) return;
  const int e0 = d.hruOff[r], e1 = d.hruOff[r + 1];
  double rs = 0.0;
  if (e1 > e0) {
    for (int e = e0; e < e1; ++e) {
      const double v = d.solSrc[(size_t)t * d.H + d.hruIdx[e]];
      if (v < 0.0) mzr_raise(d, 20, r, t, 2);
      rs = rs + d.hruW[e] * v * d.time_conv_solute * d.mass_conv_solute;
    }
    rs = rs * d.basarea[r];
  }
  if (d.doesBasinRoute == 1) d.solInst[(size_t)t * d.N + r] = d.qi[(size_t)t * d.N + r] > 0 ? rs : 0.0;
  else d.basSol[(size_t)(t + 1) * d.N + r] = d.qlat[(size_t)(t + 1) * d.N + r] > 0 ? rs : 0.0;
}
void mzr_launch_basin_solute(const MzrDev &d, hipStream_t stream) {
  dim3 block(256), grid((d.N + 255) / 256, d.W);
  hipLaunchKernelGGL(k_basin2reach_mass, grid, block, 0, stream, d);
  if (d.doesBasinRoute == 1) {
    dim3 gridO((d.N + 255) / 256, (d.W + HT - 1) / HT), gridS((d.N + 255) / 256, (d.ntdhBas + HT - 1) / HT);
    hipLaunchKernelGGL(k_hillslope_out_solute, gridO, block, 0, stream, d, 0, d.W);
    hipLaunchKernelGGL(k_hillslope_state_solute, gridS, block, 0, stream, d);
  }
}

// evaporation / precipitation of the lake reaches through the HRU mapping (main_route.f90:172-200);
// grid: x over lakes, y over steps.  lakeReachInt[l] = internal reach index of lake l.
__global__ void __launch_bounds__(64) k_lake_forcing(MzrDev d, const int *lakeReachInt, const double *evap, const double *precip,
                                                     double *lakeEvap, double *lakePrecip) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (l >= d.nLake) return;
  const int r = lakeReachInt[l];
  const int e0 = d.hruOff[r], e1 = d.hruOff[r + 1];
  double ev, pr;
  if (e1 > e0) {
    double a = 0.0, b = 0.0;
    for (int e = e0; e < e1; ++e) {
      const double ve = evap[(size_t)t * d.H + d.hruIdx[e]], vp = precip[(size_t)t * d.H + d.hruIdx[e]];
      if (ve < d.negRunoffTol || vp < d.negRunoffTol) mzr_raise(d, 20, r, t, 1);
      a = a + d.hruW[e] * ve * d.time_conv * d.length_conv;
      b = b + d.hruW[e] * vp * d.time_conv * d.length_conv;
    }
    if (a < d.runoffMin) a = d.runoffMin;
    if (b < d.runoffMin) b = d.runoffMin;
    ev = a * d.basarea[r]; pr = b * d.basarea[r];
  } else {
    ev = d.runoffMin; pr = d.runoffMin;
  }
  lakeEvap[(size_t)t * d.nLake + l] = ev; lakePrecip[(size_t)t * d.nLake + l] = pr;
}

void mzr_launch_lake_forcing(const MzrDev &d, const int *lakeReachInt, const double *evap, const double *precip,
                             double *lakeEvap, double *lakePrecip, int nSteps, hipStream_t stream) {
  if (d.nLake <= 0) return;
  dim3 block(64), grid((d.nLake + 63) / 64, nSteps);
  hipLaunchKernelGGL(k_lake_forcing, grid, block, 0, stream, d, lakeReachInt, evap, precip, lakeEvap, lakePrecip);
}

// steps [tBegin, tEnd) of the window (tBegin a multiple of the tile sizes): the hillslope fold is causal,
// so the window can be produced in chunks while the routing sweep already consumes the first ones
void mzr_launch_basin_chunk(const MzrDev &d, int tBegin, int tEnd, hipStream_t stream) {
  const int n = tEnd - tBegin;
  if (n <= 0) return;
  dim3 block(256);
  const int nRB = (d.N + 255) / 256, tiles = (n + BT - 1) / BT;
  hipLaunchKernelGGL(k_basin2reach, dim3((unsigned)(8 * ((tiles + 7) / 8) * nRB)), block, 0, stream, d, tBegin, tEnd, nRB);
  if (d.doesBasinRoute == 1) {
    dim3 gridO((d.N + 255) / 256, (n + HT - 1) / HT);
    hipLaunchKernelGGL(k_hillslope_out, gridO, block, 0, stream, d, tBegin, tEnd);
  }
}
// QFUTURE after the window (needs every BASIN_QI of the window)
void mzr_launch_basin_state(const MzrDev &d, hipStream_t stream) {
  if (d.doesBasinRoute != 1) return;
  dim3 block(256), gridS((d.N + 255) / 256, (d.ntdhBas + HT - 1) / HT);
  hipLaunchKernelGGL(k_hillslope_state, gridS, block, 0, stream, d);
}
void mzr_launch_basin(const MzrDev &d, hipStream_t stream) {
  mzr_launch_basin_chunk(d, 0, d.W, stream);
  mzr_launch_basin_state(d, stream);
}
