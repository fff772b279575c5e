#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE ONLY.
# Compiles the UNMODIFIED reference Fortran, from the sources where they lie under
# /root/reference, into oracle/_ref/libgrtrans_ref.so together with oracle/ref_driver.f90
# (our bind(C) glue).  Nothing is copied into the repo; objects and .mod files live in
# oracle/_ref/ (git-ignored, but shipped to the GPU box with the snapshot).
#
# The whole call chain is compiled: fits.f90, camera.f90, grtrans_driver.f90, pgrtrans.f90 and
# grtrans.f90 (grtrans_run) included.  fits.f90 references 19 cfitsio entry points (ft*_) that
# this image lacks; NO stubs are written: the shared library is linked with those symbols left
# undefined and is loaded with RTLD_LAZY (oracle/refbind.py, oracle/refmain.py).  cflag=0 (the
# binary writer of camera.f90:321-340) never calls them.  Only grtrans_program.f90 (the PROGRAM
# unit) is left out.  oracle/refmain.py drives the reference's own grtrans_run; ref_driver.f90
# stays as the per-stage probe and is asserted bit-equal to it (tests/test_ref_main.py).
set -e
REF=${GRT_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF" ]; then
  if [ -f "$OUT/libgrtrans_ref.so" ]; then echo "build_ref: $REF absent, using prebuilt $OUT/libgrtrans_ref.so"; exit 0; fi
  echo "build_ref: $REF absent and no prebuilt library"; exit 1
fi
FC=${FC:-amdflang}
FL="-ffixed-line-length-132 -fopenmp -O2 -fPIC -w"
mkdir -p "$OUT/obj"
cd "$OUT/obj"
CORE="phys_constants.f90 math.f90 bessel.f90 interpolate_aux.f interpolate.f90 class_four_vector.f90 kerr.f90 \
      geokerr_wrapper.f polsynchemis.f90 odepack_aux.f odepack.f90 radtrans_integrate.f90 read_inputs.f90 \
      rad_trans.f90 chandra_tab24.f90 calc_maxjutt.f90 calc_maxcomp.f90 calcgmin.f90 emis.f90 geodesics.f90"
MODELS="sphacc thindisk phatdisk numdisk hotspot hotspot_schnittman ffjet harm koral koral3d harm3d harmpi \
        iharm thickdisk mb09 sariaf powerlaw toy"
compile() {  # recompile only when the source is newer than the object
  src="$1"; obj="$(basename "${src%.*}").o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ]; then echo "  FC $(basename "$src")"; $FC $FL -c "$src" -o "$obj"; fi
}
for f in $CORE; do compile "$REF/$f"; done
for m in $MODELS; do compile "$REF/fluid_model_$m.f90"; done
compile "$REF/fluid.f90"
for f in fits camera grtrans_driver pgrtrans grtrans; do compile "$REF/$f.f90"; done
compile "$HERE/ref_driver.f90"
$FC -fopenmp -shared -o "$OUT/libgrtrans_ref.so" *.o
echo "build_ref: wrote $OUT/libgrtrans_ref.so"
