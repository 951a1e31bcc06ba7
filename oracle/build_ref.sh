#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY — builds the *reference's own* CT-HYB solver (Fortran, f2py module
# `CTQMC.mctqmc`) from the sources where they lie under /root/reference into oracle/_ref/.
# Nothing of the reference is copied into the repository: oracle/_ref/ is git-ignored and holds
# only compiler outputs (objects, generated f2py wrappers, the final .so).
#
# Recipe follows SURVEY.md §8c: amdflang for the .F90 files, g++ for the mt19937 wrapper, numpy.f2py
# for the wrapper sources, scipy's bundled OpenBLAS for BLAS/LAPACK (its symbols carry a `scipy_`
# prefix, hence the objcopy renames). NFFT3/FFTW3 are absent, so the build is WITHOUT -DUSE_NFFT and
# the reference's naive (exact-sum) transforms are used.
#
# Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load what this builds.
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/src/ctqmc_fortran" ]; then
  echo "build_ref.sh: $REF not present (GPU box?) - using prebuilt oracle/_ref if any" >&2
  exit 0
fi
S="$REF/src/ctqmc_fortran"; R="$REF/src/mtrng"
FC=${FC:-/opt/rocm/bin/amdflang}
PY=${PY:-python3}
mkdir -p "$OUT/build"; cd "$OUT/build"

PYINC=$($PY -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$($PY -c "import numpy; print(numpy.get_include())")
F2INC=$($PY -c "import numpy.f2py; print(numpy.f2py.get_include())")
SCIPYLIBS=$($PY -c "import scipy, os; print(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), 'scipy.libs'))")
OPENBLAS=$(ls "$SCIPYLIBS" | grep -E '^libscipy_openblas.*\.so$' | head -1)
EXT=$($PY -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")

g++ -O2 -std=c++11 -fPIC -c "$R/random.cc" -o random.o
for f in "$R/Random" "$S/Parameters" "$S/Ausgabe" "$S/CompoundIndex" "$S/LegendrePoly" "$S/MatrixUpdate" \
         "$S/Accumulators" "$S/AngularMomentum" "$S/SparseMatrix" "$S/States" "$S/Lanczos" "$S/Operator" \
         "$S/Signals" "$S/Progress" "$S/Nfft_base" "$S/Nfft_worm" "$S/Nfft_z" "$S/Trace" "$S/CTQMC"; do
  b=$(basename "$f")
  if [ ! -f "$b.o" ] || [ "$f.F90" -nt "$b.o" ]; then
    "$FC" -cpp -DLAPACK77_Interface -O2 -fPIC -c "$f.F90" -o "$b.o"
  fi
done

# f2py wrapper generation (reads the reference source in place, writes only here)
$PY -m numpy.f2py "$S/CTQMC.F90" -m CTQMC --lower --f2cmap "$S/.f2py_f2cmap" >f2py.log 2>&1
gcc -O2 -fPIC -Wno-incompatible-pointer-types -I"$PYINC" -I"$NPINC" -I"$F2INC" -c "$F2INC/fortranobject.c" -o fortranobject.o
gcc -O2 -fPIC -Wno-incompatible-pointer-types -I"$PYINC" -I"$NPINC" -I"$F2INC" -c CTQMCmodule.c -o CTQMCmodule.o
"$FC" -cpp -O2 -fPIC -c CTQMC-f2pywrappers2.f90 -o CTQMC-f2pywrappers2.o

# scipy's OpenBLAS exports scipy_<name>_ : rename the undefined BLAS/LAPACK references
REN=()
for x in dgetrf dgetri dgemv dger dgemm ddot dsyev dstev zgetrf zgetri zlacpy zgeev dcopy dscal daxpy dnrm2 dsyevd zgemm zgemv; do
  REN+=(--redefine-sym "${x}_=scipy_${x}_")
done
for o in *.o; do objcopy "${REN[@]}" "$o"; done

"$FC" -shared -o "$OUT/CTQMC$EXT" ./*.o -L"$SCIPYLIBS" -l:"$OPENBLAS" -Wl,-rpath,"$SCIPYLIBS" \
      -Wl,-rpath,/opt/rocm/lib/llvm/lib -lstdc++
echo "built $OUT/CTQMC$EXT"
