#!/bin/bash
# tools/_probe/build_variant_lb.sh <out.so> [extra hipcc flags...]: the library with extra
# compile flags on nsol_lbfgsb.hip (ablation builds of the Gram kernel); the other objects
# are reused.
set -e
OUT=$1; shift
C=nsol_amd/csrc
T=/tmp/nsol_variant_objs
mkdir -p $T
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I include"
for f in nsol_blur3_f32 nsol_blur3_lz_f32 nsol_blur3_f64 nsol_blur3_lz_f64 nsol_conv nsol_ops nsol_pd nsol_pd2 nsol_pdk nsol_pdp nsol_lsmr nsol_sort; do
  if [ ! -f $T/$f.o ] || [ $C/$f.hip -nt $T/$f.o ] || [ $C/nsol_blur3_dma.hpp -nt $T/$f.o ]; then hipcc $FL -c $C/$f.hip -o $T/$f.o & fi
done
hipcc $FL "$@" -c $C/nsol_lbfgsb.hip -o $T/nsol_lbfgsb_variant.o &
wait
hipcc --offload-arch=gfx950 -fPIC -shared $T/nsol_blur3_f32.o $T/nsol_blur3_lz_f32.o $T/nsol_blur3_f64.o $T/nsol_blur3_lz_f64.o $T/nsol_conv.o $T/nsol_ops.o $T/nsol_pd.o $T/nsol_pd2.o $T/nsol_pdk.o $T/nsol_pdp.o $T/nsol_lsmr.o $T/nsol_sort.o $T/nsol_lbfgsb_variant.o -o $OUT
echo built $OUT
