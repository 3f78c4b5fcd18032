// Fast FP64 powers with fifths as exponents (gfx950), shared by the KWT and the Eulerian kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace {

// ---- x**0.4 and x**0.6 for x >= 0 -------------------------------------------------------------
// The celerity law wc = (5/3) K**0.6 q**0.4 (kwt_route.f90:1290) and the stage inversion
// A = (q/K)**0.6 of the shock speed (:1331-1332) are the only transcendental work of this kernel,
// and a generic FP64 pow() costs more instructions than everything else a reach does.
//   x**(2/5) = z*v * 2**(2k),  v = (z**3)**(-1/5)      x = z * 2**(5k), z in [1,32)
//   x**(3/5) = z*w * 2**(3k),  w = (z**2)**(-1/5)
// Inverse fifth root: single-precision hardware seed (v_log_f32 / v_exp_f32), two division-free
// Newton steps v <- v*(1.2 - 0.2*a*v**5), the last with fused residual.  The reference raises to
// the DOUBLES nearest 0.4 / 0.6 (exponents (ALFA-1)/ALFA and 1/ALFA evaluated in FP64), which
// differ from 2/5 and 3/5 by -+2.22e-17; the factor (1 + delta*ln x) restores that.  Measured
// against the correctly rounded power: max 2.3 ulp, mean 0.44 ulp (same class as libm's pow).
__device__ __forceinline__ double pow_fifths(double x, bool three) {
  // ordinary arguments are positive and finite: one class test (v_cmp_class_f64: positive normal or subnormal) and one rare
  // branch for everything else -- zero -> 0, negative or NaN -> NaN, +Inf -> +Inf
  if (__builtin_expect(!__builtin_isfpclass(x, 0x180), 0)) return x == 0.0 ? 0.0 : (x > 0.0 ? x : NAN);
  int e;
  const double m = frexp(x, &e);            // x = m * 2**e, m in [0.5,1)
  const int e1 = e - 1;                     // x = (2m) * 2**e1 ; e1 = 5k + j, j in 0..4
  // k = floor(e1 / 5) without a branch on the sign (e1 >= -1074: the dividend is positive; lanes with small and large
  // discharge share wavefronts, so a branch here ran both sides)
  const int k = (int)((unsigned)(e1 + 1100) / 5u) - 220;
  const int j = e1 - 5 * k;
  const double z = ldexp(m, j + 1);
  const float lz = __builtin_amdgcn_logf((float)z);                        // log2(z)
  const double a = three ? z * z : z * z * z;
  double v = (double)__builtin_amdgcn_exp2f((three ? -0.4f : -0.6f) * lz);
  {
    const double v2 = v * v, v4 = v2 * v2;
    v = v * (1.2 - 0.2 * (a * (v4 * v)));
  }
  {
    const double v2 = v * v, v4 = v2 * v2, v5 = v4 * v;
    v = fma(v, 0.2 * fma(-a, v5, 1.0), v);
  }
  const double lnx = ((double)lz + 5.0 * (double)k) * 0.6931471805599453;
  const double dl = three ? -2.2204460492503132e-17 : 2.2204460492503132e-17;
  double y = z * v;
  y = fma(y, dl * lnx, y);
  return ldexp(y, three ? 3 * k : 2 * k);
}
__device__ __forceinline__ double pow_0p4(double x) { return pow_fifths(x, false); }   // x**((ALFA-1)/ALFA)
__device__ __forceinline__ double pow_0p6(double x) { return pow_fifths(x, true); }    // x**(1/ALFA)

// further exponents of the channel hydraulics (hydraulic.f90:438-484, 306-433) from the same root:
//   x**0.3 = sqrt(x**0.6), x**0.2 = sqrt(x**0.4)   (0.6/2 and 0.4/2 are exactly the doubles 0.3 and 0.2)
//   x**(2/3) = cbrt(x)**2                           (2/3 as a double is 3.7e-17 below 2/3: < 1 ulp for x in 1e-30..1e30)
__device__ __forceinline__ double pow_0p3(double x) { return sqrt(pow_fifths(x, true)); }
__device__ __forceinline__ double pow_0p2(double x) { return sqrt(pow_fifths(x, false)); }
__device__ __forceinline__ double pow_2_3(double x) { const double c = cbrt(x); return c * c; }

}  // namespace
