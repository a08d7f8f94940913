#!/bin/bash
# tools/_probe/ab_many.sh "<lib1.so> <lib2.so> ..." <command...>: the command under each
# build of the library, three interleaved rounds on ONE box
set -e
LIBS=$1; shift
LIB=nsol_amd/csrc/libnsol_hip.so
cp $LIB /tmp/lib_base.so
for round in 1 2 3; do
  echo -n "base  "; cp /tmp/lib_base.so $LIB; "$@"
  for l in $LIBS; do echo -n "$(basename $l) "; cp $l $LIB; "$@"; done
done
cp /tmp/lib_base.so $LIB
