#!/bin/bash
# Build the reference-solver harness oracle/_ref/ref_route from the UNMODIFIED reference
# Fortran sources where they lie under /root/reference, plus the shim modules and driver in
# oracle/ref_harness/.  Outputs go ONLY to oracle/_ref/ (git-ignored; travels with gpurun).
# Needs /root/reference and flang (ROCm's AMD flang); silently skipped otherwise.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
S=${MZR_REFERENCE_SRC:-/root/reference/route/build/src}
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
OUT="${MZR_REF_OUT:-$HERE/_ref}"      # (MZR_REF_OUT / MZR_REF_FFLAGS: a second build beside it, e.g. -O0 for the drift check of oracle/check_ref_O0.py)
if [ ! -d "$S" ] || [ ! -x "$FC" ]; then
  echo "build_ref: reference sources or flang not available; skipping (prebuilt $OUT is used if present)"
  exit 0
fi
mkdir -p "$OUT/obj"
cd "$OUT/obj"
FFLAGS="${MZR_REF_FFLAGS:--O2 -ffp-contract=off -fopenmp}"
H="$HERE/ref_harness"
SRCS="$S/nrtype.f90 $S/public_var.f90 $S/nr_utils.f90 $S/datetime_data.f90 $S/dataTypes.f90
 $S/base_route.f90 $S/var_lookup.f90 $H/shim_globalData.f90 $H/shim_runtime.f90
 $S/hydraulic.f90 $S/advection_diffusion.f90 $S/gamma_func.f90 $S/process_param.f90
 $S/water_balance.f90 $S/data_assimilation.f90 $S/accum_runoff.f90 $S/basinUH.f90
 $S/lake_route.f90 $S/irf_route.f90 $S/kwt_route.f90 $S/kwe_route.f90 $S/mc_route.f90
 $S/dfw_route.f90 $S/tracer.f90 $S/process_remap.f90 $S/main_route.f90 $H/ref_driver.f90"
OBJS=""
for f in $SRCS; do
  o="$(basename "${f%.f90}").o"
  $FC $FFLAGS -c "$f" -o "$o"
  OBJS="$OBJS $o"
done
$FC $FFLAGS $OBJS -o "$OUT/ref_route"
# second harness: the reference's forcing remap (remap_runoff, sort_flux) on its own
$FC $FFLAGS -c "$H/ref_remap_driver.f90" -o ref_remap_driver.o
$FC $FFLAGS ${OBJS/ ref_driver.o/} ref_remap_driver.o -o "$OUT/ref_remap"
# third harness: the reference's start-up routines for the river network (augment_ntopo with the network_topo.f90 routines
# it calls, mpi_domain_decomposition = classify_river_basin + assign_node), unmodified, on a plain-text case
TOPO_OBJS=""
for f in $S/pfafstetter.f90 $S/network_topo.f90 $S/process_ntopo.f90 $S/domain_decomposition.f90 $H/ref_topo_driver.f90; do
  o="$(basename "${f%.f90}").o"
  $FC $FFLAGS -c "$f" -o "$o"
  TOPO_OBJS="$TOPO_OBJS $o"
done
$FC $FFLAGS ${OBJS/ ref_driver.o/} $TOPO_OBJS -o "$OUT/ref_topo"
$FC --version | head -1 > "$OUT/BUILD_INFO.txt"
echo "flags: $FFLAGS" >> "$OUT/BUILD_INFO.txt"
echo "built $OUT/ref_route $OUT/ref_remap $OUT/ref_topo"
