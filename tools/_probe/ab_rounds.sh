#!/bin/bash
# tools/_probe/ab_rounds.sh <rounds> "<lib1.so> ..." <command...>: like ab_many.sh with a
# chosen number of interleaved rounds
set -e
R=$1; shift
LIBS=$1; shift
LIB=nsol_amd/csrc/libnsol_hip.so
cp $LIB /tmp/lib_base.so
for round in $(seq 1 $R); do
  echo -n "base  "; cp /tmp/lib_base.so $LIB; "$@"
  for l in $LIBS; do echo -n "$(basename $l) "; cp $l $LIB; "$@"; done
done
cp /tmp/lib_base.so $LIB
