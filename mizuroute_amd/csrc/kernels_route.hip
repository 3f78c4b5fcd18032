// Stage kernels for the Eulerian reach solvers (gfx950): one lane per reach, one launch per stage.
//
//   SUM  accum_inst_runoff   route/build/src/accum_runoff.f90:32-93
//   IRF  irf_rch             route/build/src/irf_route.f90:40-264
//   MC   mc_rch              route/build/src/mc_route.f90:46-416
//   DW   dfw_rch             route/build/src/dfw_route.f90:49-370
//   KW   kw_rch              route/build/src/kwe_route.f90:46-363
//   channel hydraulics       route/build/src/hydraulic.f90:46-535
//   implicit ADE + Thomas    route/build/src/advection_diffusion.f90:19-258
//   water balance            route/build/src/water_balance.f90:22-112
//
// Schedule: in launch `s` lane r advances reach r through window step t = s - sigma[r].  A reach
// is exactly one stage behind each of its immediate upstreams, so the upstream discharge of step
// t (written one launch earlier) is final, and reaches at different depths of the network work
// on different time steps of the window in the same launch (time-skewed level sweep).
// These solvers are FP64 transcendental-bound (Newton iterations with pow), not HBM-bound.
#include <algorithm>
#include <mutex>
#include "mzr_device.h"
#include "lake_device.h"
#include "mzr_math.h"

namespace {

constexpr double c23 = 2.0 / 3.0, c53 = 5.0 / 3.0, c103 = 10.0 / 3.0;

struct Chan { double b, zc, S, n, zf, D; double Qbf;       // width, side slope, slope, Manning n, floodplain slope, bank depth;
                                                                 // per reach, once: bankfull discharge (hydraulic.f90:345)
              double sq1zc, sq1zf, isqSn, ib3, Abf, Pbf, Bbf;
              mutable double obC1 = -1.0, obC2 = 0.0; };   // coefficients of the above-bankfull depth iteration, on first use   // ... and the sub-expressions the Newton iterations and Muskingum-Cunge's sub-steps
                                                                 // would evaluate again and again with the same operands: sqrt(1+zc**2), sqrt(1+zf**2),
                                                                 // n/sqrt(S), 1/b**3, bankfull A, P, B
#define MZR_CHAN_TAB 10      // doubles per reach of the channel table (MzrDev::chanTab, slot-major [MZR_CHAN_TAB][N])

__device__ __forceinline__ double d_Btop(double y, const Chan &c) {
  if (y <= c.D) return c.b + 2 * y * c.zc;
  const double bt = c.b + 2 * c.D * c.zc;
  return bt + c.zf * (y - c.D) * 2;
}
__device__ __forceinline__ double d_Pwet(double y, const Chan &c) {
  if (y <= c.D) return c.b + 2 * y * c.sq1zc;
  const double p = c.b + 2 * c.D * c.sq1zc;
  return p + 2 * (y - c.D) * c.sq1zf;
}
__device__ __forceinline__ double d_area(double y, const Chan &c) {
  if (y <= c.D) return y * (c.b + c.zc * y);
  const double a = c.D * (c.b + c.zc * c.D);
  return a + (y - c.D) * (d_Btop(y, c) + d_Btop(c.D, c)) / 2.0;
}
__device__ double d_water_height(double flowArea, const Chan &c) {
  const double A_bank = d_area(c.D, c);
  if (flowArea > A_bank) {
    const double Bb = d_Btop(c.D, c);
    const double disc = Bb * Bb - 4.0 * c.zf * (A_bank - flowArea);
    return c.D + (-Bb + sqrt(disc)) / (2.0 * c.zf);
  }
  if (c.zc == 0) return flowArea / c.b;
  return (-c.b + sqrt(c.b * c.b + 4.0 * flowArea * c.zc)) / (2.0 * c.zc);
}
__device__ __forceinline__ void d_chan_overbank(const Chan &c) {
  const double sqS = sqrt(c.S), zh = c.zf / 2, czh = cbrt(zh);
  c.obC1 = sqS / c.n / pow_2_3(c.Pbf);
  c.obC2 = 2 * (zh * (czh * czh)) * sqS / c.n / cbrt(c.zf * c.zf + 1.0);
}
// Newton-Raphson normal depth, hydraulic.f90:306-433 (integer powers as left-to-right products)
__device__ double d_flow_depth(double Qin, const Chan &c) {
  if (!(Qin > 1.e-50)) return 0.0;
  const double Abf = c.Abf, Bbf = c.Bbf;
  const double Qbf = c.Qbf;
  double fd = 0.0;
  if (Qin < Qbf) {
    // Coef1 = (sqrt(S)/n / Q)**3, first guess (1 / Coef1 / b**3)**0.2, then Newton on h = Coef1 A**5 / P**2 - 1 with
    // dh/dy = Coef1 (5 A**4 Bt P - 2 Coef2 A**5) / P**3 (:344-350).  h and dh/dy enter only as h / dhdy, which with
    // u3 = 1 / Coef1 is (A**5 - u3 P**2) P / (5 A**4 Bt P - 2 Coef2 A**5): one division per iteration instead of five, and
    // the stopping test |(fd - y0) / fd| > 0.005 without its division.  This loop is a chain of dependent FP64 operations
    // with nothing to overlap it, and a Muskingum-Cunge reach walks through it in each of its 2-200 sub-steps per step.
    const double u = Qin * c.isqSn, u3 = u * u * u;
    const double Coef2 = 2 * c.sq1zc;
    double y0 = pow_0p2(u3 * c.ib3);
    int guard = 0;
    bool more = true;
    while (more && guard++ < 200) {
      const double A = d_area(y0, c), Bt = d_Btop(y0, c), P = d_Pwet(y0, c);
      const double A2 = A * A, A4 = A2 * A2, A5 = A4 * A;
      fd = y0 - (A5 - u3 * (P * P)) * P / (5 * A4 * Bt * P - 2 * Coef2 * A5);
      more = fabs(fd - y0) > 0.005 * fabs(fd);
      y0 = fd;
    }
  } else {
    // above bankfull (:383-425).  Every power here has thirds as exponent: x**(2/3) = cbrt(x)**2, x**(5/3) = x cbrt(x)**2,
    // ye**(10/3) / ye**(2/3) = ye**3 cbrt(ye) / cbrt(ye)**2 -- two cube roots per iteration instead of five generic
    // pow() (a few hundred instructions each, and a flooding reach is typically also one with many Muskingum-Cunge
    // sub-steps); the two coefficients are the reach's own and are kept for the next call.  Negative ye: NaN like pow().
    double y0 = c.D + 2.0;
    if (c.obC1 < 0.0) d_chan_overbank(c);
    const double Coef1 = c.obC1, Coef2 = c.obC2;
    int guard = 0;
    bool more = true;
    while (more && guard++ < 200) {
      const double ye = y0 - c.D, X = Abf + Bbf * ye;
      const double cx = cbrt(X), cy = ye < 0.0 ? NAN : cbrt(ye);
      const double X23 = cx * cx, y23 = cy * cy;
      const double h = Coef1 * (X * X23) + Coef2 * (ye * ye * ye * cy) / y23 - Qin;
      const double dhdy = Coef1 * c53 * Bbf * X23 + Coef2 * (c103 - c23) * (ye * y23);
      fd = y0 - h / dhdy;
      more = fabs(fd - y0) > 0.005 * fabs(fd);
      y0 = fd;
    }
  }
  return fd;
}
__device__ double d_friction_slope(double Qin, double y, const Chan &c) {
  const double A = d_area(y, c), P = d_Pwet(y, c);
  const double v = Qin * c.n / A / pow_2_3(A / P);
  return v * v;
}
// hydraulic.f90:438-484: ck = 5/3 Sf**0.3 Q**0.4 / Bt**0.4 / n**0.6 with Sf = (Q n / A / R**(2/3))**2 from Manning's
// equation.  Written out, n and the thirds cancel: ck = 5/3 (Q / A) (P / Bt)**0.4 -- one power and two divisions
// instead of four powers, a cube root and six divisions, on the critical chain of every Muskingum-Cunge sub-step.
__device__ double d_celerity(double Qin, double y, const Chan &c) {
  if (!(y > 0.0)) return 0.0;
  const double A = d_area(y, c), P = d_Pwet(y, c), Bt = d_Btop(y, c);
  return c53 * Qin * pow_0p4(P / Bt) / A;
}
__device__ double d_diffusivity(double Qin, double y, const Chan &c) {
  if (!(y > 0.0)) return 0.0;
  const double Bt = d_Btop(y, c);
  const double Sf = d_friction_slope(Qin, y, c);
  return fabs(Qin) / Sf / Bt / 2.0;
}

// Implicit central-difference ADE with Neumann outflow (advec_scheme=2, downBC=2, wck=wdk=1;
// dfw_route.f90:36-37) solved with the reference's Thomas recurrence.  With these weights the
// matrix rows are constant, so the forward sweep needs no arrays besides D and b1.
template <int NM>
__device__ void d_solve_ade(double L, double dtl, double Fup, double ck, double dk, const double *prev, double *sol) {
  const double wck = 1.0, wdk = 1.0;
  const int Nx = NM - 1;
  const double dx = L / (Nx - 1);
  const double Cd = dk * dtl / (dx * dx);
  const double Ca = ck * dtl / dx;
  const double midv = 2.0 + 4 * wdk * Cd;
  const double upv = wck * Ca - 2.0 * wdk * Cd;        // diagonal(3:NM,1)
  const double lowv = -wck * Ca - 2.0 * wdk * Cd;      // diagonal(1:NM-2,3)
  const double e1 = (1.0 - wck) * Ca + 2.0 * (1.0 - wdk) * Cd;
  const double e2 = 2.0 - 4.0 * (1.0 - wdk) * Cd;
  const double e3 = (1.0 - wck) * Ca - 2.0 * (1.0 - wdk) * Cd;
  double D[NM + 1], b1[NM + 1];
  // row values, 1-based like advection_diffusion.f90
  auto mid = [&](int i) { return (i == 1 || i == NM) ? 1.0 : midv; };
  auto up = [&](int i) { return i >= 3 ? upv : 0.0; };
  auto low = [&](int i) { return i <= NM - 2 ? lowv : (i == NM - 1 ? -1.0 : 0.0); };
  auto rhs = [&](int i) {
    if (i == 1) return Fup;
    if (i == NM) return prev[NM - 1] - prev[NM - 2];
    return e1 * prev[i - 2] + e2 * prev[i - 1] - e3 * prev[i];
  };
  D[1] = mid(1); b1[1] = rhs(1);
#pragma unroll
  for (int i = 2; i <= NM; ++i) {
    const double coef = low(i - 1) / D[i - 1];
    D[i] = mid(i) - coef * up(i);
    b1[i] = rhs(i) - coef * b1[i - 1];
  }
  sol[NM - 1] = b1[NM] / D[NM];
#pragma unroll
  for (int i = NM - 1; i >= 1; --i) sol[i - 1] = (b1[i] - up(i + 1) * sol[i]) / D[i];
}

// shared preamble of irf/mc/dw/kw (irf_route.f90:81-100; water management is not active here)
struct Pre { double q_up, q_up_mod, Qlat; bool isHW; };
template <bool COH> __device__ __forceinline__ Pre d_preamble(const MzrDev &d, int r, const double *Qrow, double qlat) {
  Pre p; p.q_up = 0.0; p.isHW = true;
  const int ng = d.nGood[r];
  if (ng > 0) {
    p.isHW = false;
    const int u0 = d.upStart[r];
    const uint32_t gm = d.goodMask[r];
    for (int i = 0; i < ng; ++i) {                 // do iUps=1,nUps ; cycle if .not.goodBas(iUps)
      if (!((gm >> i) & 1u)) continue;
      p.q_up = p.q_up + ldx<COH>(Qrow + u0 + i);
    }
    p.q_up_mod = p.q_up; p.Qlat = qlat;
  } else if (d.hw_drain_point == 1) {
    p.q_up = p.q_up + qlat; p.q_up_mod = p.q_up; p.Qlat = 0.0;
  } else {
    p.q_up_mod = p.q_up; p.Qlat = qlat;
  }
  return p;
}

// water_balance.f90:22-112 (non-lake)
__device__ __forceinline__ double d_wb(double vol1, double vol0, double Qup, double Qlat, double Qout, double wmAct, double dt) {
  const double dVol = vol1 - vol0;
  const double Qin = Qup * dt, Qlateral = Qlat * dt, precip = 0.0;
  const double Qo = -1.0 * Qout * dt, Qtake = -1.0 * wmAct * dt, evapo = 0.0;
  return dVol - (Qin + Qlateral + precip + Qtake + Qo + evapo);
}

// The channel of a reach: its six parameters and what the solvers derive from them alone.  Round 6: the derived values -- four square
// roots, two cube roots, a 2/3 power and seven divisions, ~300 dependent FP64 instructions that every reach-step of Muskingum-Cunge,
// KW and DW used to start with -- are computed ONCE per parameter set by k_chan_table (this very function, so the same bits) and
// loaded: ten coalesced doubles per reach-step.
__device__ __forceinline__ Chan d_chan_compute(const MzrDev &d, int r) {
  Chan c; c.b = d.width[r]; c.zc = d.side[r]; c.S = d.slope[r]; c.n = d.mann[r]; c.zf = d.fldp[r]; c.D = d.depth[r];
  c.sq1zc = sqrt(1 + c.zc * c.zc); c.sq1zf = sqrt(1 + c.zf * c.zf);
  c.isqSn = c.n / sqrt(c.S); c.ib3 = 1.0 / (c.b * c.b * c.b);
  const double Abf = d_area(c.D, c), Pbf = d_Pwet(c.D, c);
  c.Abf = Abf; c.Pbf = Pbf; c.Bbf = d_Btop(c.D, c);
  c.Qbf = Abf * pow_2_3(Abf / Pbf) * sqrt(c.S) / c.n;      // hydraulic.f90:345
  return c;
}
__device__ __forceinline__ Chan d_chan(const MzrDev &d, int r) {
  if (!d.chanTab) return d_chan_compute(d, r);      // (wave-uniform: a kernel argument)
  Chan c; c.b = d.width[r]; c.zc = d.side[r]; c.S = d.slope[r]; c.n = d.mann[r]; c.zf = d.fldp[r]; c.D = d.depth[r];
  const double *t = d.chanTab + r;
  const size_t N = d.N;
  c.sq1zc = t[0]; c.sq1zf = t[N]; c.isqSn = t[2 * N]; c.ib3 = t[3 * N]; c.Abf = t[4 * N]; c.Pbf = t[5 * N]; c.Bbf = t[6 * N]; c.Qbf = t[7 * N];
  c.obC1 = t[8 * N]; c.obC2 = t[9 * N];
  return c;
}

}  // namespace

__global__ void __launch_bounds__(256) k_chan_table(MzrDev d, double *tab) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= d.N) return;
  MzrDev dd = d; dd.chanTab = nullptr;
  const Chan c = d_chan_compute(dd, r);
  d_chan_overbank(c);
  const size_t N = d.N;
  double *t = tab + r;
  t[0] = c.sq1zc; t[N] = c.sq1zf; t[2 * N] = c.isqSn; t[3 * N] = c.ib3; t[4 * N] = c.Abf; t[5 * N] = c.Pbf; t[6 * N] = c.Bbf; t[7 * N] = c.Qbf;
  t[8 * N] = c.obC1; t[9 * N] = c.obC2;
}
int mzr_chan_table_doubles() { return MZR_CHAN_TAB; }
void mzr_launch_chan_table(const MzrDev &d, double *tab, hipStream_t stream) {
  hipLaunchKernelGGL(k_chan_table, dim3((d.N + 255) / 256), dim3(256), 0, stream, d, tab);
}

// ------------------------------------------------------------------------------------------------
// main_route.f90:125-148 (observations of this step, or one more step since the last ones) and direct_insertion
// (data_assimilation.f90:28-97): returns REACH_Q with the decaying discharge error taken off.
template <bool COH>
__device__ __forceinline__ double d_direct_insertion(const MzrDev &d, int r, int t, double Q) {
  int el = ldx<COH>(d.qelapsed + r);
  double qobs = ldx<COH>(d.qobs + r);
  if (d.obsHave[t]) {
    for (int g = d.gaugeFirst[r]; g >= 0; g = d.gaugeNext[g]) {
      const double v = d.obsVal[(size_t)t * d.nGauge + g];
      if ((v != v) || (v < 0)) continue;
      qobs = v; el = 0;
    }
  } else {
    el = el + 1;
  }
  stx<COH>(d.qobs + r, qobs); stx<COH>(d.qelapsed + r, el);
  double qerr = ldx<COH>(d.qerr + r);
  const int B = d.qBlendPeriod;
  if (qobs > 0.0) qerr = Q - qobs;
  if (el > B) qerr = 0.0;
  double Qc = 0.0;
  if (el <= B) {
    switch (d.QerrTrend) {
      case 1: Qc = qerr; break;
      case 2: Qc = qerr * (1.0 - (double)el / (double)B); break;
      case 3: {
        const double x0 = 0.25, y0 = (double)0.90f;      // default-real literals in the reference (:78)
        const double k = log(1.0 / y0 - 1.0) / (B / 2.0 - B * x0);
        Qc = qerr / (1.0 + exp(-k * (1.0 * el - B / 2.0)));
        break;
      }
      default:
        if (qerr != 0.0) { const double k = log(0.1 / fabs(qerr)) / (1.0 * B); Qc = qerr * exp(k * el); }
        break;
    }
  }
  stx<COH>(d.qerr + r, qerr);
  return fmax(Q - Qc, 0.0);
}

// One reach, one step of the window.  COH: the persistent sweep -- discharge rows and the reach's own state are produced and
// consumed by different wavefronts of the SAME launch, so they go through sc1 accesses (ldx / stx, mzr_device.h);
// COH = false is the launch-per-stage form with plain accesses.
// FULL = false compiles the rarely used branches out -- lakes, water-management fluxes, gauge observations, the constituent's
// REACH_VOL(0) rows -- and with them a third of the registers (Muskingum-Cunge 167 -> see DESIGN.md 4): these kernels are chains
// of dependent FP64 operations, and how many wavefronts a SIMD holds is what hides them.
template <int METHOD, bool COH, bool FULL = true>
__device__ __forceinline__ void stage_reach(const MzrDev &d, int r, int t) {
  const int N = d.N;
  double *Qrow = d.Q + (size_t)t * N;
  if (d.haloSlot) {          // tributary outlet computed in another partition: discharge is imported
    const int hs = d.haloSlot[r];
    if (hs >= 0) { stx<COH>(Qrow + r, d.imQ[(size_t)t * d.nHalo + hs]); return; }
  }
  const double qlat = d.qlat[(size_t)(t + 1) * N + r];
  const double dt = d.dt;

  if (FULL && METHOD != 0 && d.lakeSlot) {   // lake reach: lake_route replaces the reach solver (main_route.f90:375-381)
    const int ls = d.lakeSlot[r];
    if (ls >= 0) {
      double vol = ldx<COH>(d.vol + r), vol0 = vol, ele = ldx<COH>(d.ele + r), wb = 0.0, wmAct = 0.0;
      const double Q = mzr_lake::lake_route(d, r, t, ls, Qrow, qlat, vol, vol0, ele, wb, wmAct, COH);
      if (d.trVol0) d.trVol0[(size_t)t * N + r] = vol0;
      stx<COH>(Qrow + r, Q); stx<COH>(d.vol + r, vol); stx<COH>(d.vol0 + r, vol0); stx<COH>(d.ele + r, ele); stx<COH>(d.wb + r, wb);
      stx<COH>(d.qsum + r, ldx<COH>(d.qsum + r) + Q);
      if (d.wmact) stx<COH>(d.wmact + r, wmAct);
      if (d.hEle) { stx<COH>(d.hEle + r, ldx<COH>(d.hEle + r) + ele); stx<COH>(d.hFlood + r, ldx<COH>(d.hFlood + r) + ldx<COH>(d.floodvol + r)); }
      if (d.hInflow) stx<COH>(d.hInflow + r, ldx<COH>(d.hInflow + r) + d.inflow[r]);     // (lake_route stores the inflow plainly)
      return;
    }
  }

  if (METHOD == 0) {   // SUM: all upstreams, regardless of goodBas (accum_runoff.f90:60-75)
    double q = qlat;
    const int nu = d.nUp[r];
    if (nu > 0) {
      const int u0 = d.upStart[r];
      double qu = 0.0;
      for (int i = 0; i < nu; ++i) qu = qu + ldx<COH>(Qrow + u0 + i);
      q = q + qu;
    }
    stx<COH>(Qrow + r, q);
    stx<COH>(d.qsum + r, ldx<COH>(d.qsum + r) + q);
    return;
  }

  Pre p = d_preamble<COH>(d, r, Qrow, qlat);
  stx<COH>(d.inflow + r, p.q_up);
  double vol = ldx<COH>(d.vol + r);
  const double vol_prev = vol;        // REACH_VOL(0) = REACH_VOL(1)
  double vol0 = vol_prev;
  // water management: abstraction from storage, then upstream inflow, then lateral flow; injection
  // into the lateral flow (irf_route.f90:118-142, identical in mc/dfw/kwe)
  const double wmflux = (FULL && d.is_flux_wm && d.wm) ? d.wm[(size_t)t * N + r] : 0.0;
  double wmAct = wmflux;
  if (FULL && d.is_flux_wm && wmflux != -9999.0) {
    double Qabs = wmflux;
    if (Qabs > 0) {
      if (vol / dt > Qabs) {
        vol = vol - Qabs * dt;
      } else {
        Qabs = Qabs - vol / dt;
        vol = 0.0;
        if (p.q_up > Qabs) {
          p.q_up_mod = p.q_up - Qabs;
        } else {
          Qabs = Qabs - p.q_up;
          p.q_up_mod = 0.0;
          if (p.Qlat > Qabs) {
            p.Qlat = p.Qlat - Qabs;
          } else {
            Qabs = Qabs - p.Qlat;
            p.Qlat = 0.0;
            wmAct = wmflux - Qabs;
          }
        }
      }
    } else {
      p.Qlat = p.Qlat - Qabs;
    }
  }
  const double L = d.length[r];
  double Qout;

  if (METHOD == 1) {   // IRF: conv_upsbas_qr, irf_route.f90:210-264
    const int nt = d.ntdh[r];
    const double qu = p.q_up_mod;
    if (L > d.min_length_route) {
      double q0 = ldx<COH>(d.irfQ + r) + d.uh[r] * qu;
      const double lim = (fmax(0.0, vol) / dt + qu) * (double)0.999f;   // default-real literal, :245
      q0 = fmin(lim, q0);
      vol = vol - (q0 - qu) * dt;
      Qout = q0 + p.Qlat;
      // eoshift(shift=1) of the convolution window, eight taps at a time: the loads of a batch are all in flight before its first
      // store (taken one by one, a tap's store -- which may alias the next tap's load as far as the compiler knows -- keeps the next
      // load waiting: one memory round trip per tap, and this kernel is nothing but memory round trips)
      int j = 1;
      for (; j + 8 <= nt; j += 8) {
        double a[8], u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = ldx<COH>(d.irfQ + (size_t)(j + k) * N + r); u[k] = d.uh[(size_t)(j + k) * N + r]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) stx<COH>(d.irfQ + (size_t)(j + k - 1) * N + r, a[k] + u[k] * qu);
      }
      if (j + 4 <= nt) {
        double a[4], u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = ldx<COH>(d.irfQ + (size_t)(j + k) * N + r); u[k] = d.uh[(size_t)(j + k) * N + r]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) stx<COH>(d.irfQ + (size_t)(j + k - 1) * N + r, a[k] + u[k] * qu);
        j += 4;
      }
      for (; j < nt; ++j) {
        const double v = ldx<COH>(d.irfQ + (size_t)j * N + r) + d.uh[(size_t)j * N + r] * qu;
        stx<COH>(d.irfQ + (size_t)(j - 1) * N + r, v);
      }
      stx<COH>(d.irfQ + (size_t)(nt - 1) * N + r, 0.0);
    } else {
      for (int j = 0; j < nt; ++j) stx<COH>(d.irfQ + (size_t)j * N + r, 0.0);
      stx<COH>(d.irfQ + r, qu);
      Qout = qu + p.Qlat;
      vol0 = 0.0; vol = 0.0;
    }
  } else if (METHOD == 4) {   // Muskingum-Cunge, mc_route.f90:204-416
    const Chan c = d_chan(d, r);
    const double Q00 = ldx<COH>(d.mol + r), Q01 = ldx<COH>(d.mol + (size_t)N + r);
    double Q10, Q11, flood = 0.0, ele = 0.0;
    if (!p.isHW || d.hw_drain_point == 1) {
      if (L > d.min_length_route) {
        const double theta = dt / L;
        Q10 = p.q_up_mod;
        double Qbar = (Q00 + Q10 + Q01) / 3.0;
        if (Qbar > 1.e-50) {
          double depth = d_flow_depth(fabs(Qbar), c);
          double ck = d_celerity(fabs(Qbar), depth, c);
          double Cn = ck * theta;
          int ntSub = 1; double dTsub = dt;
          if (Cn > 1.0) { ntSub = (int)ceil(dt / L * ck); dTsub = dt / ntSub; }
          const double Y = 0.5, SL = c.S * L, thSub = dTsub / L;
          double qin_prev = Q00, qout_prev = Q01, ssum = 0.0;
          // From the second sub-step on the inflow pair is (Q10, Q10), so a sub-step is a fixed map of the previous
          // outflow alone, and that map contracts towards Q10: once it returns its own argument (or alternates between two
          // neighbouring doubles) every remaining sub-step is known, and only their running sum -- the same additions in
          // the same order -- is left to do.  With 20..200 sub-steps per step (smooth channels) most of them are skipped.
          double qprev2 = -1.0;
          int nexec = 0;      // sub-steps really executed (the fixed-point and closed-form exits leave the loop early)
#ifdef MZR_MC_STATS
          int _nexec = 0;
#endif
          for (int ix = 1; ix <= ntSub; ++ix) {
            nexec = ix;
#ifdef MZR_MC_STATS
            _nexec = ix;
#endif
            const double qin = Q10;
            double qo;
            Qbar = (qin + qin_prev + qout_prev) / 3.0;
            if (Qbar > 1.e-50) {
              // (the first sub-step's mean discharge is the step's own -- (Q10 + Q00 + Q01) / 3 = (Q00 + Q10 + Q01) / 3 bit for bit, the
              // first addition commutes -- so its normal depth and celerity are the ones the sub-step count was worked out from: the
              // reference solves twice (mc_route.f90:246-262 and :283-296), same arguments, same results.  Half the work of the
              // reaches with one sub-step, which most are)
              if (ix > 1) {
                depth = d_flow_depth(fabs(Qbar), c);
                ck = d_celerity(fabs(Qbar), depth, c);
              }
              const double topWidth = d_Btop(depth, c);
              const double X = 0.5 * (1.0 - Qbar / (topWidth * SL * ck));
              Cn = ck * thSub;
              // C0 qin + C1 qin_prev + C2 qout_prev with the common denominator of the three coefficients (:305-312)
              const double den = 1 - X + Cn * (1 - Y);
              qo = ((-X + Cn * (1 - Y)) * qin + (X + Cn * Y) * qin_prev + (1 - X - Cn * Y) * qout_prev) / den;
              qo = fmax(0.0, qo);
            } else {
              qo = 0.0;
            }
            ssum = ssum + qo;
            if (ix >= 2 && ix < ntSub) {
              if (qo == qout_prev) {                       // fixed point of the sub-step map
                for (int k = ix + 1; k <= ntSub; ++k) ssum = ssum + qo;
                break;
              }
              if (ix >= 3 && qo == qprev2) {               // two-cycle: ..., qo, qout_prev, qo, ...
                for (int k = ix + 1; k <= ntSub; ++k) ssum = ssum + (((k - ix) & 1) ? qout_prev : qo);
                break;
              }
              // Close to its fixed point q* the map is linear to second order, q(k) = q* + e rho**k.  Once the outflow moves by
              // less than mcTailTol (1e-7) of itself per sub-step and the differences contract, the sub-steps still to come
              // are added in closed form (a geometric series with rho from the last two differences).  Whatever rho is worth,
              // the sum left out is at most |d2| / (1 - rho_true) -- the bound is on the step actually taken, not on the
              // estimate -- and observed deviations from iterating on are ~1e-10 of the discharge (stated tolerance 1e-6).
              // Flat short reaches (rho ~ 0.9, 50-200 sub-steps, a hundred of them in 100 k) are what this is for: they set
              // the duration of every launch.  MZR_MC_TAIL_TOL=0 iterates every sub-step.
              if (ix >= 3 && qo > 0.0) {
                const double d2 = qo - qout_prev, d1 = qout_prev - qprev2;
                if (fabs(d2) <= d.mcTailTol * qo && fabs(d2) < 0.995 * fabs(d1)) {
                  const double rho = d2 / d1, g = rho / (1.0 - rho);
                  const double qs = qo + d2 * g;                                  // q*
                  const int m = ntSub - ix;
                  double rm = exp((double)m * log(fabs(rho)));                    // |rho|**m
                  if (rho < 0.0 && (m & 1)) rm = -rm;
                  ssum = ssum + ((double)m * qs + (qo - qs) * g * (1.0 - rm));
                  break;
                }
              }
            }
            qprev2 = qout_prev;
            qin_prev = qin; qout_prev = qo;
          }
#ifdef MZR_MC_STATS
          if (d.dbgCycles) {   // tools/dbg_mc.py: sub-steps asked for / executed (log2 bins), executed per reach
            atomicAdd(&d.dbgCycles[16 + min(15, 31 - __clz(ntSub))], 1ull); atomicAdd(&d.dbgCycles[min(15, 31 - __clz(max(1, _nexec)))], 1ull);
            d.dbgCycles[32 * 1024 + 64 + r] += (unsigned long long)_nexec;
            atomicMax(&d.dbgCycles[32 * 1024 + 64 + 200000 + (s & 4095)], (unsigned long long)_nexec);
          }
#endif
          if (d.mcSub) d.mcSub[r] = (unsigned short)(nexec < 65535 ? nexec : 65535);
          Q11 = ssum / (double)ntSub;
          if (fabs(Q11) > 0.0) {
            const double pr = fmin((vol / dt + Q10) * (double)0.999f / Q11, 1.0);   // default-real literal, :352
            Q11 = Q11 * pr;
          }
        } else {
          Q11 = 0.0;
        }
        vol = vol + (Q10 - Q11) * dt;
        const double stor = d.storage[r];
        flood = vol > stor ? vol - stor : 0.0;
        ele = d_water_height(vol / L, c);
        Qout = Q11 + p.Qlat;
      } else {
        Q10 = p.q_up_mod; Q11 = p.q_up_mod;
        Qout = p.q_up_mod + p.Qlat;
        vol0 = 0.0; vol = 0.0;
      }
    } else {
      Q10 = 0.0; Q11 = 0.0; Qout = p.Qlat; vol0 = 0.0; vol = 0.0;
    }
    stx<COH>(d.mol + r, Q10); stx<COH>(d.mol + (size_t)N + r, Q11);
    stx<COH>(d.floodvol + r, flood); stx<COH>(d.ele + r, ele);
  } else {   // DW (5) and KW (3): dfw_route.f90:209-370, kwe_route.f90:205-363
    constexpr int NM = 20;
    const Chan c = d_chan(d, r);
    const double Qu = p.q_up_mod;
    double flood = 0.0, ele = 0.0;
    if (!p.isHW || d.hw_drain_point == 1) {
      if (L > d.min_length_route) {
        double prev[NM], sol[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) prev[i] = ldx<COH>(d.mol + (size_t)i * N + r);
        const double Qbar = (Qu + prev[0] + prev[NM - 2]) / 3.0;
        const double depth = d_flow_depth(fabs(Qbar), c);
        const double ck = d_celerity(fabs(Qbar), depth, c);
        const double dk = (METHOD == 5) ? d_diffusivity(fabs(Qbar), depth, c) : 0.0;
        d_solve_ade<NM>(L, dt / 1, Qu, ck, dk, prev, sol);
        if (fabs(sol[NM - 2]) > 0.0) {
          const double volTmp = fmax(0.0, vol);
          const double qoutTmp = sol[NM - 2] * dt;
          const double pr = fmin((volTmp + dt * Qu) * 0.999 / qoutTmp, 1.0);
#pragma unroll
          for (int i = 1; i < NM; ++i) sol[i] = sol[i] * pr;
        }
        vol = vol + (Qu - sol[NM - 2]) * dt;
        const double stor = d.storage[r];
        flood = vol > stor ? vol - stor : 0.0;
        ele = d_water_height(vol / L, c);
        Qout = sol[NM - 2] + p.Qlat;
#pragma unroll
        for (int i = 0; i < NM; ++i) stx<COH>(d.mol + (size_t)i * N + r, sol[i]);
      } else {
        Qout = Qu + p.Qlat;
        for (int i = 0; i < NM - 1; ++i) stx<COH>(d.mol + (size_t)i * N + r, 0.0);
        stx<COH>(d.mol + (size_t)(NM - 1) * N + r, Qout);
        vol0 = 0.0; vol = 0.0;
      }
    } else {
      Qout = p.Qlat; vol0 = 0.0; vol = 0.0;
      for (int i = 0; i < NM - 1; ++i) stx<COH>(d.mol + (size_t)i * N + r, 0.0);
      stx<COH>(d.mol + (size_t)(NM - 1) * N + r, Qout);
    }
    stx<COH>(d.floodvol + r, flood); stx<COH>(d.ele + r, ele);
  }
  if (FULL && d.qmod) Qout = d_direct_insertion<COH>(d, r, t, Qout);      // irf_route.f90:188-198 and alike: after the solver, before anybody reads REACH_Q
  stx<COH>(Qrow + r, Qout);
  stx<COH>(d.vol + r, vol); stx<COH>(d.vol0 + r, vol0);
  if (FULL && d.trVol0) d.trVol0[(size_t)t * N + r] = vol0;      // REACH_VOL(0) of the step, for the constituent pass
  if (!(FULL && d.qmod)) stx<COH>(d.wb + r, d_wb(vol, vol0, p.q_up, p.Qlat, Qout, wmAct, dt));      // the water balance only without data assimilation (:200-202)
  if (FULL && d.wmact) stx<COH>(d.wmact + r, wmAct);
  stx<COH>(d.qsum + r, ldx<COH>(d.qsum + r) + Qout);
  // history sums of the other per-method fluxes (histVars_data.f90:229-246), when asked for
  if (d.hInflow) stx<COH>(d.hInflow + r, ldx<COH>(d.hInflow + r) + p.q_up);
  if (d.hEle) { stx<COH>(d.hEle + r, ldx<COH>(d.hEle + r) + ldx<COH>(d.ele + r)); stx<COH>(d.hFlood + r, ldx<COH>(d.hFlood + r) + ldx<COH>(d.floodvol + r)); }
}

// Reach of a lane position p (block-aligned: launches start at a multiple of 256): through the method's lane permutation
// when there is one (mzr_device.h, lanePerm).  -1 = nothing to do.
// Lanes per workgroup of the stage kernels, measured per method (profiles/r04_experiments.md, reach-steps/s at 100 k / 625 k
// reaches): one wavefront for IRF (5.3 / 6.6 x 10^9 against 4.6 / 6.0 with four: a workgroup keeps its slots until its
// last wavefront is done, and the tap counts differ) and for KW / DW (4.7 against 4.4-4.7 x 10^9 at 100 k); four for
// Muskingum-Cunge, whose lanes are dealt to the wavefronts of a 256-position block by sub-step count (5.2 against 4.3-4.5 x 10^9
// on the shard: the block's wavefronts share what they fetch of the block's span through their CU).
__host__ __device__ constexpr int stage_wg(int method) { return (method == 4 || method == 0) ? 256 : 64; }
__device__ __forceinline__ int stage_lane_pos(int L, int base) { return base + L * (int)blockDim.x + (int)threadIdx.x; }
__device__ __forceinline__ int stage_lane_reach(const MzrDev &d, int p, int rBegin, int rEnd) {
  if (p >= ((rEnd + 255) & ~255)) return -1;
  const int r = d.lanePerm ? d.lanePerm[p] : p;
  return (r >= rBegin && r < rEnd) ? r : -1;
}
// lane position of a thread: the blocks [0, nHB) of a launch serve the heavy positions (mzr_device.h, nHeavyPos), the others the
// positions from `base` on; heavy = the position is one of those (its reach is taken whatever block range the launch covers)
__device__ __forceinline__ int stage_lane_reach_at(const MzrDev &d, int blk, int nHB, int base, int rBegin, int rEnd) {
  if (blk < nHB) {
    const int r = d.lanePerm[d.permN + blk * (int)blockDim.x + (int)threadIdx.x];
    return (r >= rBegin && r < rEnd) ? r : -1;
  }
  return stage_lane_reach(d, stage_lane_pos(blk - nHB, base), rBegin, rEnd);
}

// Muskingum-Cunge without the rare branches needs 155 VGPRs (3 wavefronts per SIMD); MZR_MC_WAVES asks the compiler for more
// wavefronts (fewer registers, spills if it must): measured, see DESIGN.md 6
#ifndef MZR_MC_WAVES
#define MZR_MC_WAVES 0
#endif
#if MZR_MC_WAVES
#define MZR_STAGE_OCC(M, F) __attribute__((amdgpu_waves_per_eu(((M) == 4 && !(F)) ? MZR_MC_WAVES : 1, ((M) == 4 && !(F)) ? MZR_MC_WAVES : 8)))
#else
#define MZR_STAGE_OCC(M, F)
#endif
// Block tb of a reach's steps: with d.stepBlock = KB > 1 the time skew of the schedule is counted in blocks of KB steps -- launch s
// takes every reach of stage j through the steps [KB (s - j), KB (s - j) + KB) -- so the reach's state (the IRF convolution window,
// the 20 nodes of KW / DW) is still in the caches when its next step reads it, and a window is S + W / KB - 1 launches.  The
// upstream rows of the block were written by the launch before (stage j - 1: the same block) or earlier.  Same arithmetic per
// reach and step, same order.
template <int METHOD, bool FULL>
__device__ __forceinline__ void stage_reach_block(const MzrDev &d, int r, int tb) {
  if (tb < 0) return;
  const int KB = d.stepBlock;
  const int t0 = tb * KB;
  if (t0 >= d.W) return;
  const int t1 = min(d.W, t0 + KB);
#pragma unroll 1
  for (int t = t0; t < t1; ++t) stage_reach<METHOD, false, FULL>(d, r, t);
}

// BLK = false is the one-step-per-launch form as it always was: the loop of the blocked form costs registers even when it
// runs once (DW 208 -> 260 VGPRs, Muskingum-Cunge 153 -> 202, IRF 74 -> 112), so the two are separate instantiations.
template <int METHOD, bool FULL, bool BLK>
__global__ void __launch_bounds__(stage_wg(METHOD)) MZR_STAGE_OCC(METHOD, FULL) k_stage(MzrDev d, int s, int rBegin, int rEnd) {
  const int nHB = d.lanePerm ? d.nHeavyPos / (int)blockDim.x : 0;
  const int r = stage_lane_reach_at(d, (int)blockIdx.x, nHB, rBegin & ~255, rBegin, rEnd);
  if (r < 0) return;
  if (BLK) { stage_reach_block<METHOD, FULL>(d, r, s - d.sigma[r]); return; }
  const int t = s - d.sigma[r];
  if (t < 0 || t >= d.W) return;
  stage_reach<METHOD, false, FULL>(d, r, t);
}

// Two windows in one launch (round 4, "overlapping windows").  The skewed schedule of a window of W steps over S stages is
// S + W - 1 launches, and the last S - 1 of them (the window drains: only stages > s - W still have steps left) are as
// empty as the first S - 1 of the next window (it fills: only stages <= s have started).  The two touch complementary
// stages -- launch j of window k+1 works on stages 0..j, launch W + j of window k on stages j+1..S-1 -- and step 0 of a
// reach in window k+1 only needs the reach's own last step of window k (one launch earlier) and its upstream reaches' step
// 0 (same window, one launch earlier).  So the host keeps the drain of a window back until the next window arrives and
// issues both as ONE launch: blocks [0, nBlocksB) take the new window (p.b), the others the old one (p.a), each with the
// window's own rows (discharge, lateral flow, lake forcing: double-buffered on the host side).  A window then costs W
// launches instead of S + W - 1, every one of them over all reaches.  Same arithmetic per reach and step, same order.
struct MzrDevPair { MzrDev a, b; };
template <int METHOD, bool FULL, bool BLK>
__global__ void __launch_bounds__(stage_wg(METHOD)) MZR_STAGE_OCC(METHOD, FULL) k_stage_pair(MzrDevPair p, int sA, int rBeginA, int rEndA, int sB, int rBeginB, int rEndB, int nBlocksB) {
  // blocks: [heavy positions, new window | heavy positions, old window | new window | old window]
  const int nHB = p.b.lanePerm ? p.b.nHeavyPos / (int)blockDim.x : 0;
  const int bx = (int)blockIdx.x;
  const bool old = bx < 2 * nHB ? bx >= nHB : bx - 2 * nHB >= nBlocksB;      // wave-uniform: the domain description is read through scalar loads either way
  const MzrDev &d = old ? p.a : p.b;
  const int rB = old ? rBeginA : rBeginB, rE = old ? rEndA : rEndB;
  const int r = bx < 2 * nHB ? stage_lane_reach_at(d, bx - (old ? nHB : 0), nHB, 0, rB, rE)
                             : stage_lane_reach(d, stage_lane_pos(bx - 2 * nHB - (old ? nBlocksB : 0), rB & ~255), rB, rE);
  if (r < 0) return;
  if (BLK) { stage_reach_block<METHOD, FULL>(d, r, (old ? sA : sB) - d.sigma[r]); return; }
  const int t = (old ? sA : sB) - d.sigma[r];
  if (t < 0 || t >= d.W) return;
  stage_reach<METHOD, false, FULL>(d, r, t);
}

// ------------------------------------------------------------------------------------------------
// Persistent sweep of an Eulerian method over a window: what one launch per stage does (k_stage), without the launches.
// A window of W steps over S stages is S + W - 1 "launches" of the skewed schedule; with a kernel per launch every one
// of them costs a launch latency and waits for its slowest lane, and the first and last S of them are mostly empty.
// Here the items of all launches (up to 64 reaches of one stage each) are numbered in launch order and drawn by the
// wavefronts from eight ticket counters (one per XCD, item i in queue i % 8); a reach's step t starts when its upstream
// reaches have published step t and itself step t - 1 (rtDone), so a slow lane holds up its own downstream chain only.
// Tickets depend only on tickets of earlier launches and every queue is served in order: any number of resident
// wavefronts makes progress.  Discharge rows and per-reach state cross wavefronts through sc1 accesses (stage_reach<.., true>).
namespace {
// pause of a polling wavefront; true = give up (another wavefront raised an error, or none of the progress words this
// wavefront polls has changed for d.stallTicks: code 93 with a record of the wait instead of a hung GPU).
// sig = sum of the words the lane polled, bad* = the first dependency of the lane that is not there yet.
__device__ __forceinline__ bool rt_pause(const MzrDev &d, int &spins, long long &t0, int sig, int &sigLast, bool waiting,
                                         int r, int s, int q, int badReach, int badSeen, int badNeed) {
  __builtin_amdgcn_s_sleep(4);
  if ((++spins & 31) == 0) {
    if (ldx<true>(&d.err->code) != 0) return true;
    const long long now = wall_clock64();     // 100 MHz
    if (!t0 || __ballot(sig != sigLast) != 0ull) t0 = now;
    else if (now - t0 > d.stallTicks) {
      const unsigned long long bad = __ballot(waiting);
      const int first = __ffsll((long long)bad) - 1;
      if ((int)threadIdx.x == first) mzr_raise_stall(d, 21, r, s, badReach, badSeen, badNeed, q, first, __popcll(bad), now - t0, d.rtHead);
      return true;
    }
    sigLast = sig;
  }
  return false;
}
}  // namespace

template <int METHOD>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_sweep_route(MzrDev d, int sBegin, int sEnd) {
  if (sEnd < 0) { mzr_census(d.rtHead + 8 * 16); return; }      // host: mzr_sweep_route_capacity
  if (ldx<true>(&d.err->code) != 0) return;      // a window that failed stays as it is (and is not built upon)
  const int arr = mzr_sweep_join(d.rtHead);      // (a wavefront that starts behind time does not join)
  if (arr < 0) return;
  const int lane = threadIdx.x;
  const int q0 = arr < 64 ? (arr & 7) : (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7);     // the first 64 take queue j % 8; otherwise HW_REG_XCC_ID, a speed hint only
  const int *P = d.rtP, *RAs = d.rtRA;
#pragma unroll 1
  for (int dq = 0; dq < 8; ++dq) {
    const int q = (q0 + dq) & 7;
    const int pEnd = P[sEnd * 8 + q];
    int sCur = sBegin, pLo = P[sBegin * 8 + q], pHi = P[(sBegin + 1) * 8 + q];   // tickets [pLo, pHi) of queue q belong to launch sCur
#pragma unroll 1
    for (;;) {
      int k = 0;
      if (lane == 0) k = atomicAdd(d.rtHead + q * 16, 1);
      k = __builtin_amdgcn_readfirstlane(k);
      if (k >= pEnd) break;
      if (k >= pHi) {
        ++sCur; pLo = pHi; pHi = P[(sCur + 1) * 8 + q];
        if (k >= pHi) {
          int lo = sCur + 1, hi = sEnd - 1;          // largest s with P[s] <= k
          while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (P[mid * 8 + q] <= k) lo = mid; else hi = mid - 1; }
          sCur = lo; pLo = P[sCur * 8 + q]; pHi = P[(sCur + 1) * 8 + q];
        }
      }
      const int s = sCur;
      const int a = RAs[s];
      const int i = a + ((q - a) & 7) + 8 * (k - pLo);
      // one 16-byte record per lane: reach, first upstream reach, number of upstream reaches | lake flag << 8, stage
      const int4 rec = ((const int4 *)d.rtItemR)[(size_t)i * 64 + lane];
      const int r = rec.x, u0 = rec.y, nu = rec.z & 0xff;
      const bool lakes = __ballot((rec.z >> 8) & 1) != 0ull;
      const int t = s - rec.w;                          // 0 <= t < W by construction of the tables
      // its upstream reaches have published step t, itself step t - 1 (another wavefront's work): polled together
      int spins = 0, sigLast = 0; long long tw0 = 0;
      for (;;) {
        bool ok = true;
        int sig = 0, badReach = -1, badSeen = 0, badNeed = 0;
        if (r >= 0) {
          if (t >= 1) { const int w = ldx<true>(d.rtDone + r); sig += w; if (w < t) { ok = false; badReach = r; badSeen = w; badNeed = t; } }
          for (int j = 0; j < nu; ++j) {
            const int w = ldx<true>(d.rtDone + u0 + j);
            sig += w;
            if (w < t + 1) { if (ok) { badReach = u0 + j; badSeen = w; badNeed = t + 1; } ok = false; }
          }
        }
        if (__ballot(!ok) == 0ull) break;
        if (rt_pause(d, spins, tw0, sig, sigLast, !ok, r, s, q, badReach, badSeen, badNeed)) return;
      }
      asm volatile("" ::: "memory");      // what the progress words guard is read after them
      if (lakes) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                  // a lake's plain state (Hanasaki memory ...)
      if (r >= 0) stage_reach<METHOD, true>(d, r, t);
      if (lakes) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (r >= 0) stx<true>(d.rtDone + r, t + 1);
    }
  }
}

__global__ void k_rt_heads(MzrDev d, int sBegin) {
  if (threadIdx.x < 8) d.rtHead[threadIdx.x * 16] = d.rtP[sBegin * 8 + threadIdx.x];
  mzr_sweep_join_reset(d.rtHead);
}

// wavefronts of a method's sweep kernel the device really holds at once (measured: see mzr_sweep_kwt_capacity);
// d.rtHead must point at 8 * 16 + 48 ints (eight ticket heads, census / arrival words, histogram of the start delays)
int mzr_sweep_route_capacity(int method, const MzrDev &d, hipStream_t stream) {
  static int cached[16][6];
  static std::mutex mu;      // handles of several host threads share the cache; the census itself must not run twice at once either
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0, cus = 0, perCu = 0;
  if (method < 0 || method > 5 || hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 16 && cached[dev][method]) return cached[dev][method];
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  hipError_t e = hipErrorInvalidValue;
  switch (method) {
    case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_sweep_route<0>, 64, 0); break;
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_sweep_route<1>, 64, 0); break;
    case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_sweep_route<3>, 64, 0); break;
    case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_sweep_route<4>, 64, 0); break;
    case 5: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_sweep_route<5>, 64, 0); break;
    default: break;
  }
  if (e != hipSuccess) return 0;
  const int api = cus * perCu;
  int peak[2] = {0, 0};
  int *cnt = d.rtHead + 8 * 16;
  if (hipMemsetAsync(cnt, 0, 2 * sizeof(int), stream) != hipSuccess) return 0;
  dim3 block(64), grid(api + api / 4);
  switch (method) {
    case 0: hipLaunchKernelGGL(k_sweep_route<0>, grid, block, 0, stream, d, 0, -1); break;
    case 1: hipLaunchKernelGGL(k_sweep_route<1>, grid, block, 0, stream, d, 0, -1); break;
    case 3: hipLaunchKernelGGL(k_sweep_route<3>, grid, block, 0, stream, d, 0, -1); break;
    case 4: hipLaunchKernelGGL(k_sweep_route<4>, grid, block, 0, stream, d, 0, -1); break;
    case 5: hipLaunchKernelGGL(k_sweep_route<5>, grid, block, 0, stream, d, 0, -1); break;
    default: break;
  }
  const hipError_t eLaunch = hipGetLastError();
  const hipError_t eSync = hipStreamSynchronize(stream);
  if (eLaunch != hipSuccess || eSync != hipSuccess) {
    fprintf(stderr, "mzr: census of the sweep of method %d: launch of %d wavefronts: %s, synchronize: %s\n", method, api + api / 4, hipGetErrorString(eLaunch), hipGetErrorString(eSync));
    return 0;
  }
  if (hipMemcpy(peak, cnt, sizeof peak, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  const int cap = peak[1] > 0 ? std::min(api, peak[1]) : 0;
  if (cap < 1) fprintf(stderr, "mzr: census of the sweep of method %d: %d wavefronts launched (%d per CU by the occupancy query), %d counted, peak %d\n", method, api + api / 4, perCu, peak[0], peak[1]);
  if (dev >= 0 && dev < 16 && 2 * cap >= api) cached[dev][method] = cap;      // (a census far below the occupancy query ran beside other work: measured again next time)
  return cap;
}

void mzr_launch_sweep_route(int method, const MzrDev &d, int nWaves, int sBegin, int sEnd, hipStream_t stream) {
  if (nWaves < 1 || sEnd <= sBegin) return;
  hipLaunchKernelGGL(k_rt_heads, dim3(1), dim3(64), 0, stream, d, sBegin);
  dim3 block(64), grid(nWaves);
  switch (method) {
    case 0: hipLaunchKernelGGL(k_sweep_route<0>, grid, block, 0, stream, d, sBegin, sEnd); break;
    case 1: hipLaunchKernelGGL(k_sweep_route<1>, grid, block, 0, stream, d, sBegin, sEnd); break;
    case 3: hipLaunchKernelGGL(k_sweep_route<3>, grid, block, 0, stream, d, sBegin, sEnd); break;
    case 4: hipLaunchKernelGGL(k_sweep_route<4>, grid, block, 0, stream, d, sBegin, sEnd); break;
    case 5: hipLaunchKernelGGL(k_sweep_route<5>, grid, block, 0, stream, d, sBegin, sEnd); break;
    default: break;
  }
}

// ------------------------------------------------------------------------------------------------
// constituent_rch + comp_mass_flux (tracer.f90:43-207) as a pass of its own over the routed window: it needs, per reach and
// step, what the solvers left in the window's rows -- REACH_Q, BASIN_QR(1), REACH_VOL(0), and the upstream reaches'
// constituent flux of the same step -- so it follows a method's sweep in the same skewed order, one launch per stage.
// REACH_INFLOW is the sum the solver made (good upstream reaches in UREACHI order; plus the lateral flow for a headwater
// poured in at the top; 0 for a KWT headwater and for lakes, whose solver never sets it).
__global__ void __launch_bounds__(256) k_tracer_stage(MzrDev d, int method, int s, int rBegin, int rEnd) {
  const int r = rBegin + blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rEnd) return;
  const int t = s - d.sigma[r];
  if (t < 0 || t >= d.W) return;
  if (d.haloSlot && d.haloSlot[r] >= 0) return;      // a tributary outlet routed in another partition: its flux came with the boundary record
  const int N = d.N;
  const double dt = d.dt;
  const double *Qrow = d.Q + (size_t)t * N, *Frow = d.solFlux + (size_t)t * N;
  const double bsol = d.basSol[(size_t)(t + 1) * N + r], qr1 = d.qlat[(size_t)(t + 1) * N + r];
  const int ng = d.nGood[r], u0 = d.upStart[r];
  const uint32_t gm = d.goodMask[r];
  const bool isHW = ng == 0;
  const bool lake = d.lakeSlot && d.lakeSlot[r] >= 0;
  double Cup = 0.0, Clat = 0.0, qin = 0.0;
  if (!isHW) {
    for (int i = 0; i < ng; ++i) { if (!((gm >> i) & 1u)) continue; Cup = Cup + Frow[u0 + i]; qin = qin + Qrow[u0 + i]; }
    Clat = bsol;
  } else if (d.hw_drain_point == 1) {
    Cup = Cup + bsol; Clat = 0.0;
    if (method != 2) qin = qin + qr1;
  } else {
    Clat = bsol;
  }
  if (lake) qin = 0.0;
  double mass1 = d.solMass[r];
  const double mass0 = mass1;
  double flux;
  if (!isHW || d.hw_drain_point == 1) {
    const double reach_mass = Cup * dt + mass0;
    const double reach_vol = qin * dt + d.trVol0[(size_t)t * N + r];
    double per_vol = 0.0;
    if (reach_vol > 0.0) per_vol = reach_mass / reach_vol;
    double out = (Qrow[r] - qr1) * per_vol;
    const double maxOut = mass1 / dt + Cup;
    if (out > maxOut) { out = maxOut; mass1 = 0; }
    else mass1 = mass1 + (Cup - out) * dt;
    flux = out + Clat;
  } else {
    flux = Clat; mass1 = 0.0;
  }
  d.solFlux[(size_t)t * N + r] = flux;
  d.solMass[r] = mass1;
}
void mzr_launch_tracer_stage(int method, const MzrDev &d, int s, int rBegin, int rEnd, hipStream_t stream) {
  const int n = rEnd - rBegin;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_tracer_stage, dim3((n + 255) / 256), dim3(256), 0, stream, d, method, s, rBegin, rEnd);
}

// old window `a` at launch sA over reaches [rBeginA, rEndA), new window `b` at launch sB over [rBeginB, rEndB)
void mzr_launch_stage_pair(int method, const MzrDev &a, int sA, int rBeginA, int rEndA, const MzrDev &b, int sB, int rBeginB, int rEndB, hipStream_t stream) {
  const int nA = std::max(0, rEndA - rBeginA), nB = std::max(0, rEndB - rBeginB);
  if (nA + nB <= 0) return;
  MzrDevPair p; p.a = a; p.b = b;
  const int wg = stage_wg(method);
  // (with a lane permutation the reaches of a 256-position block may sit anywhere in it: the covered range ends on a block boundary)
  auto blocks = [wg](const MzrDev &v, int rB, int rE) { const int e = v.lanePerm ? ((rE + 255) & ~255) : rE; return rE > rB ? (e - (rB & ~255) + wg - 1) / wg : 0; };
  const int nBlocksB = blocks(b, rBeginB, rEndB);
  const int nHB = b.lanePerm ? b.nHeavyPos / wg : 0;      // (the host keeps the two windows' heavy positions the same: mc_regroup)
  dim3 block(wg), grid(2 * nHB + nBlocksB + blocks(a, rBeginA, rEndA));
  const bool full = (a.lakeSlot || a.is_flux_wm || a.qmod || a.trVol0) || (b.lakeSlot || b.is_flux_wm || b.qmod || b.trVol0);
  const bool blk = a.stepBlock > 1 || b.stepBlock > 1;
#define MZR_PAIR(M) case M: \
    if (full) { if (blk) hipLaunchKernelGGL((k_stage_pair<M, true, true>), grid, block, 0, stream, p, sA, rBeginA, rEndA, sB, rBeginB, rEndB, nBlocksB); \
                else hipLaunchKernelGGL((k_stage_pair<M, true, false>), grid, block, 0, stream, p, sA, rBeginA, rEndA, sB, rBeginB, rEndB, nBlocksB); } \
    else { if (blk) hipLaunchKernelGGL((k_stage_pair<M, false, true>), grid, block, 0, stream, p, sA, rBeginA, rEndA, sB, rBeginB, rEndB, nBlocksB); \
           else hipLaunchKernelGGL((k_stage_pair<M, false, false>), grid, block, 0, stream, p, sA, rBeginA, rEndA, sB, rBeginB, rEndB, nBlocksB); } \
    break;
  switch (method) {
    MZR_PAIR(0) MZR_PAIR(1) MZR_PAIR(3) MZR_PAIR(4) MZR_PAIR(5)
    default: break;
  }
}

void mzr_launch_stage(int method, const MzrDev &d, int s, int rBegin, int rEnd, hipStream_t stream) {
  const int n = rEnd - rBegin;
  if (n <= 0) return;
  const int wg = stage_wg(method);
  const int rCover = d.lanePerm ? ((rEnd + 255) & ~255) : rEnd;      // a lane permutation moves reaches anywhere inside their 256-position block
  dim3 block(wg), grid((d.lanePerm ? d.nHeavyPos / wg : 0) + (rCover - (rBegin & ~255) + wg - 1) / wg);
  const bool full = (d.lakeSlot || d.is_flux_wm || d.qmod || d.trVol0);
  const bool blk = d.stepBlock > 1;
#define MZR_STAGE(M) case M: \
    if (full) { if (blk) hipLaunchKernelGGL((k_stage<M, true, true>), grid, block, 0, stream, d, s, rBegin, rEnd); else hipLaunchKernelGGL((k_stage<M, true, false>), grid, block, 0, stream, d, s, rBegin, rEnd); } \
    else { if (blk) hipLaunchKernelGGL((k_stage<M, false, true>), grid, block, 0, stream, d, s, rBegin, rEnd); else hipLaunchKernelGGL((k_stage<M, false, false>), grid, block, 0, stream, d, s, rBegin, rEnd); } \
    break;
  switch (method) {
    MZR_STAGE(0) MZR_STAGE(1) MZR_STAGE(3) MZR_STAGE(4) MZR_STAGE(5)
    default: break;
  }
}
