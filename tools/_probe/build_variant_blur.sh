#!/bin/bash
# tools/_probe/build_variant_blur.sh <out.so> [extra hipcc flags...]: the library with the two
# float32 blur translation units (nsol_blur3_f32, nsol_blur3_lz_f32) compiled with extra
# flags; the other objects as build_variant_tu.sh keeps them in /tmp/nsol_variant_objs.
set -e
OUT=$1; shift
C=${NSOL_CSRC:-nsol_amd/csrc}
T=/tmp/nsol_variant_objs
mkdir -p $T
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I include"
ALL="nsol_blur3_f32 nsol_blur3_lz_f32 nsol_blur3_f64 nsol_blur3_lz_f64 nsol_conv nsol_ops nsol_pd nsol_pd2 nsol_pdk nsol_pdp nsol_lsmr nsol_lbfgsb nsol_sort"
for f in $ALL; do
  case $f in nsol_blur3_f32|nsol_blur3_lz_f32) continue;; esac
  if [ ! -f $T/$f.o ] || [ $C/$f.hip -nt $T/$f.o ]; then hipcc $FL -c $C/$f.hip -o $T/$f.o & fi
done
B=$(basename $OUT .so)
hipcc $FL "$@" -c $C/nsol_blur3_f32.hip -o $T/b3_$B.o &
hipcc $FL "$@" -c $C/nsol_blur3_lz_f32.hip -o $T/b3lz_$B.o &
wait
OBJS=""
for f in $ALL; do
  case $f in nsol_blur3_f32) OBJS="$OBJS $T/b3_$B.o";; nsol_blur3_lz_f32) OBJS="$OBJS $T/b3lz_$B.o";; *) OBJS="$OBJS $T/$f.o";; esac
done
hipcc --offload-arch=gfx950 -fPIC -shared $OBJS -o $OUT
echo built $OUT
