#!/bin/bash
# A/B builds of the library with extra -D switches, next to the product build (never loaded unless GPG_LIB names it):
#   tools/build_variant.sh nopersist -DGPG_NO_PERSIST      ->  gpgradpy_amd/libgpgrad_hip_nopersist.so
# Built in a scratch copy of csrc/ so that the product's objects are untouched.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
B=/tmp/gpg_variant_$TAG
rm -rf $B && mkdir -p $B/gpgradpy_amd $B/include
cp -r $R/gpgradpy_amd/csrc $B/gpgradpy_amd/csrc
cp $R/include/*.h $B/include/
rm -f $B/gpgradpy_amd/csrc/*.o
make -C $B/gpgradpy_amd/csrc -j6 EXTRA="$*" OUT=$R/gpgradpy_amd/libgpgrad_hip_$TAG.so > $B/build.log 2>&1 || { tail -20 $B/build.log; exit 1; }
ls -la $R/gpgradpy_amd/libgpgrad_hip_$TAG.so
