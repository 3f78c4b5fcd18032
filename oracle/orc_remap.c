/* CPU restatement of the forcing remap in front of basin2reach (TEST INFRASTRUCTURE ONLY, see
 * mzr_oracle.h).  Follows route/build/src/process_remap.f90:
 *   remap_1D_runoff  163-261   runoff on HM_HRU polygons (1-D vector) -> river-network HRUs
 *   remap_2D_runoff   58-157   runoff on a grid -> river-network HRUs
 *   sort_flux        268-316   runoff already on river-network HRUs, in file order
 * Pinned against the unmodified reference routines through oracle/_ref/ref_remap
 * (oracle/ref_harness/ref_remap_driver.f90, tests/test_oracle_vs_ref.py).
 * Indices are 1-based like the reference; ORC_IMISS (-9999) = integerMissing (public_var.f90:44). */
#include <math.h>
#include <stddef.h>
#include "mzr_oracle.h"

#define ORC_IMISS (-9999)
#define ORC_RMISS (-9999.0)
static const double xTol = 1.e-6;   /* process_remap.f90:74,181 */

/* one time step; basinRunoff[H] is updated in place exactly like the reference's persistent array
   (HRUs the mapping does not mention keep their value) */
int orc_remap_1d(int nMap, const int *hru_ix, const int *num_qhru, const int *qhru_ix,
                 const long long *qhru_id, const long long *src_id, const double *weight,
                 const double *sim, double *basinRunoff) {
  int ixOverlap = 0;
  for (int iHRU = 0; iHRU < nMap; iHRU++) {
    const int jHRU = hru_ix[iHRU];
    if (jHRU == ORC_IMISS) {                                   /* :189-194 */
      if (num_qhru[iHRU] != ORC_IMISS) ixOverlap += num_qhru[iHRU];
      continue;
    }
    double sumWeights = 0.0;
    basinRunoff[jHRU - 1] = 0.0;
    for (int ixPoly = 0; ixPoly < num_qhru[iHRU]; ixPoly++) {
      if (qhru_ix[ixOverlap] == ORC_IMISS) { ixOverlap++; continue; }   /* :208-211 */
      const int ixRunoff = qhru_ix[ixOverlap];
      if (qhru_id && src_id && qhru_id[ixOverlap] != src_id[ixRunoff - 1]) return 20;   /* :217-220 */
      if (sim[ixRunoff - 1] > -xTol) {                         /* :223-226 */
        sumWeights = sumWeights + weight[ixOverlap];
        basinRunoff[jHRU - 1] = basinRunoff[jHRU - 1] + weight[ixOverlap] * sim[ixRunoff - 1];
      }
      ixOverlap++;
    }
    if (sumWeights > xTol) {                                   /* :246-248 */
      if (fabs(1.0 - sumWeights) > xTol) basinRunoff[jHRU - 1] = basinRunoff[jHRU - 1] / sumWeights;
    }
  }
  return 0;
}

/* sim2d is the Fortran array sim2d(1:n1, 1:n2) in memory order (first index fastest);
   i_index addresses dimension 1, j_index dimension 2 (:104-105,129) */
int orc_remap_2d(int nMap, const int *hru_ix, const int *num_qhru, const int *i_index, const int *j_index,
                 const double *weight, int n1, int n2, const double *sim2d, double *basinRunoff) {
  int ixOverlap = 0;
  for (int iHRU = 0; iHRU < nMap; iHRU++) {
    const int jHRU = hru_ix[iHRU];
    if (jHRU == ORC_IMISS) {
      if (num_qhru[iHRU] != ORC_IMISS) ixOverlap += num_qhru[iHRU];
      continue;
    }
    double sumWeights = 0.0;
    basinRunoff[jHRU - 1] = 0.0;
    for (int ixPoly = 0; ixPoly < num_qhru[iHRU]; ixPoly++) {
      const int jj = j_index[ixOverlap], ii = i_index[ixOverlap];
      if (ii < 1 || ii > n1) { ixOverlap++; continue; }        /* :108-115 */
      if (jj < 1 || jj > n2) { ixOverlap++; continue; }        /* :118-126 */
      const double v = sim2d[(size_t)(jj - 1) * n1 + (ii - 1)];
      if (v > -xTol) {
        sumWeights = sumWeights + weight[ixOverlap];
        basinRunoff[jHRU - 1] = basinRunoff[jHRU - 1] + weight[ixOverlap] * v;
      }
      ixOverlap++;
    }
    if (sumWeights > xTol) {
      if (fabs(1.0 - sumWeights) > xTol) basinRunoff[jHRU - 1] = basinRunoff[jHRU - 1] / sumWeights;
    }
  }
  return 0;
}

/* sorted_flux[nOut] = realMissing; sorted_flux(IX_in(i)) = flux_in(i); negatives (the missing
   value included) -> 0 when remove_negatives (:297-314) */
int orc_sort_flux(int nIn, const int *ix_in, const double *flux_in, int remove_negatives, int nOut, double *sorted_flux) {
  for (int j = 0; j < nOut; j++) sorted_flux[j] = ORC_RMISS;
  for (int i = 0; i < nIn; i++) {
    const int j = ix_in[i];
    if (j == ORC_IMISS) continue;
    sorted_flux[j - 1] = flux_in[i];
  }
  if (remove_negatives) for (int j = 0; j < nOut; j++) if (sorted_flux[j] < 0.0) sorted_flux[j] = 0.0;
  return 0;
}
