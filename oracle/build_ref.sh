#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.
# Compiles the reference PHY (srsRAN_Project 23.5) *in place* from /root/reference into
# oracle/_ref/ (git-ignored).  No reference source is copied into this repository; only
# object code lands in oracle/_ref/.  The reference's own CMake build cannot configure in
# this image (MbedTLS/GTest/FFTW/yaml-cpp absent), so the sources that make up the PHY
# are compiled directly, with the same per-file ISA flags the reference's CMake uses
# (lib/phy/upper/channel_coding/ldpc/CMakeLists.txt:36-41).  Files that need absent
# third-party headers (FFTW, yaml-cpp) or CMake-generated headers are simply left out;
# nothing is stubbed.  The harness oracle/ref_capi.cpp instantiates the reference classes
# through their public factory functions / headers.
set -euo pipefail
R=${REFERENCE_ROOT:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
OBJ=$OUT/obj
if [ ! -d "$R/lib/phy" ]; then
  echo "build_ref: $R not present - keeping prebuilt oracle/_ref (if any)"; exit 0
fi
mkdir -p "$OBJ"
CXX=${CXX:-g++}
BASE="-std=c++14 -O3 -fPIC -fno-strict-aliasing -w -mavx -mavx2 -mfma -msse4.1 -mpclmul \
 -DHAVE_SSE -DHAVE_AVX -DHAVE_AVX2 -DHAVE_FMA \
 -I$R/include -I$R/external/fmt/include -I$R/external -I$R"
SRCS=$(find $R/lib/phy $R/lib/srsvec $R/lib/srslog $R/lib/support $R/lib/ran $R/lib/ofh/compression -name '*.cpp' \
  | grep -v -E 'neon|fftw|generic_functions_factories\.cpp|config_yaml\.cpp|/version/version\.cpp|build_info\.cpp' | sort)
SRCS="$SRCS $R/lib/scheduler/support/tbs_calculator.cpp $R/external/fmt/src/format.cc $R/external/fmt/src/os.cc"
compile_one() {
  src=$1
  o=$OBJ/$(echo "${src#$R/}" | tr '/' '_' | sed 's/\.[a-z]*$/.o/')
  extra=""
  case "$src" in
    *ofh/compression/*avx512*) extra="-mavx512f -mavx512bw -mavx512vl -mavx512dq -mavx512cd" ;; # lib/ofh/compression/CMakeLists.txt:37-40
    *avx512*) extra="-mavx512f -mavx512bw" ;;
  esac
  if [ ! -f "$o" ] || [ "$src" -nt "$o" ]; then
    $CXX $BASE $extra -c "$src" -o "$o"
  fi
}
export -f compile_one
export R OBJ CXX BASE
echo $SRCS | tr ' ' '\n' | xargs -P ${JOBS:-8} -I{} bash -c 'compile_one {}'
rm -f $OUT/libsrsran_ref.a
ar rcs $OUT/libsrsran_ref.a $OBJ/*.o
# Harness: C API over the reference classes (my code, reference headers).
$CXX $BASE -shared -o $OUT/libref_capi.so $HERE/ref_capi.cpp -Wl,--whole-archive -Wl,--no-whole-archive $OUT/libsrsran_ref.a -lpthread
# One-off table generator (3GPP polar tables -> srsran_project_23.5_amd/csrc/tables/nr_polar_tables.h); built here, run by hand.
$CXX $BASE -O1 $HERE/gen_polar_tables.cpp $OUT/libsrsran_ref.a -lpthread -o $OUT/gen_polar_tables
# Drop-in test: reference objects vs the "hip" adapters, same stimuli (runs on the GPU box only).
REPO=$(cd "$HERE/.." && pwd)
if [ -f "$REPO/srsran_project_23.5_amd/libmiphy.so" ]; then
  $CXX $BASE -O1 -g -rdynamic -I$REPO/include -I$REPO/srsran_project_23.5_amd/adapters -I/opt/rocm/include $HERE/dropin_test.cpp $OUT/libsrsran_ref.a \
    -L$REPO/srsran_project_23.5_amd -lmiphy -L/opt/rocm/lib -lamdhip64 -lpthread \
    -Wl,-rpath,'$ORIGIN/../../srsran_project_23.5_amd' -Wl,-rpath,/opt/rocm/lib -o $OUT/dropin_test
fi
echo "build_ref: OK -> $OUT/libref_capi.so"
