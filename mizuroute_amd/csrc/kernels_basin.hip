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

// grid: x over reaches, y over steps of the window
__global__ void __launch_bounds__(256) k_basin2reach(MzrDev d) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (r >= d.N) return;
  if (d.haloSlot && d.haloSlot[r] >= 0) return;     // lateral inflow of a halo reach is imported
  const double *ro = d.runoff + (size_t)t * d.H;
  const int e0 = d.hruOff[r], e1 = d.hruOff[r + 1];
  double rr;
  if (e1 > e0) {
    double acc = 0.0;
    for (int e = e0; e < e1; ++e) {
      const double v = ro[d.hruIdx[e]];
      if (v < d.negRunoffTol) mzr_raise(d, 20, r, t, 1);   // process_remap.f90:397-402
      acc = acc + d.hruW[e] * v * d.time_conv * d.length_conv;
    }
    if (acc < d.runoffMin) acc = d.runoffMin;
    rr = acc * d.basarea[r];
  } else {
    rr = d.runoffMin;
  }
  if (d.doesBasinRoute == 1) d.qi[(size_t)t * d.N + r] = rr;
  else d.qlat[(size_t)(t + 1) * d.N + r] = rr;            // main_route.f90:223-226
}

// grid: x over reaches, y over steps: BASIN_QR(1) of step t
__global__ void __launch_bounds__(256) k_hillslope_out(MzrDev d) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (r >= d.N) return;
  if (d.haloSlot && d.haloSlot[r] >= 0) return;
  const int n = d.ntdhBas, N = d.N;
  double acc = (t < n) ? d.basS0[(size_t)t * N + r] : 0.0;
  if (d.lakeSlot && d.lakeSlot[r] >= 0) {   // lakes: impulse response, basinUH.f90:116-119
    d.qlat[(size_t)(t + 1) * N + r] = acc + 1.0 * d.qi[(size_t)t * N + r];
    return;
  }
  const int tau0 = t - n + 1 > 0 ? t - n + 1 : 0;
  for (int tau = tau0; tau <= t; ++tau) acc = acc + d.fracFuture[t - tau] * d.qi[(size_t)tau * N + r];
  d.qlat[(size_t)(t + 1) * N + r] = acc;
}

// grid: x over reaches, y over register slots j: QFUTURE(j+1) after the window
__global__ void __launch_bounds__(256) k_hillslope_state(MzrDev d) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (r >= d.N) return;
  const int n = d.ntdhBas, N = d.N, W = d.W;
  const int tv = W + j;                       // step at which this slot would be emitted
  double acc = (tv < n) ? d.basS0[(size_t)tv * N + r] : 0.0;
  if (d.lakeSlot && d.lakeSlot[r] >= 0) { d.basS1[(size_t)j * N + r] = acc; return; }
  const int tau0 = tv - n + 1 > 0 ? tv - n + 1 : 0;
  for (int tau = tau0; tau < W; ++tau) acc = acc + d.fracFuture[tv - tau] * d.qi[(size_t)tau * N + r];
  d.basS1[(size_t)j * N + r] = acc;
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

void mzr_launch_basin(const MzrDev &d, hipStream_t stream) {
  dim3 block(256), grid((d.N + 255) / 256, d.W);
  hipLaunchKernelGGL(k_basin2reach, grid, block, 0, stream, d);
  if (d.doesBasinRoute == 1) {
    hipLaunchKernelGGL(k_hillslope_out, grid, block, 0, stream, d);
    dim3 gridS((d.N + 255) / 256, d.ntdhBas);
    hipLaunchKernelGGL(k_hillslope_state, gridS, block, 0, stream, d);
  }
}
