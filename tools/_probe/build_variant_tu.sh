#!/bin/bash
# tools/_probe/build_variant_tu.sh <out.so> <translation unit without .hip> [extra hipcc
# flags...]: the library with ONE translation unit compiled with extra flags (A/B builds);
# the other objects are compiled once and reused.
set -e
OUT=$1; TU=$2; shift; shift
C=nsol_amd/csrc
T=/tmp/nsol_variant_objs
mkdir -p $T
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I include"
ALL="nsol_blur3_f32 nsol_blur3_lz_f32 nsol_blur3_f64 nsol_blur3_lz_f64 nsol_conv nsol_ops nsol_pd nsol_pd2 nsol_pdk nsol_pdp nsol_lsmr nsol_lbfgsb nsol_sort"
for f in $ALL; do
  if [ $f != $TU ] && { [ ! -f $T/$f.o ] || [ $C/$f.hip -nt $T/$f.o ] || [ $C/nsol_blur3_dma.hpp -nt $T/$f.o -a ${f:0:10} = nsol_blur3 ]; }; then hipcc $FL -c $C/$f.hip -o $T/$f.o & fi
done
V=$T/${TU}_$(basename $OUT .so).o
hipcc $FL "$@" -c $C/$TU.hip -o $V &
wait
OBJS=""
for f in $ALL; do if [ $f = $TU ]; then OBJS="$OBJS $V"; else OBJS="$OBJS $T/$f.o"; fi; done
hipcc --offload-arch=gfx950 -fPIC -shared $OBJS -o $OUT
echo built $OUT
