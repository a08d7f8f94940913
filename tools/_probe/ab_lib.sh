#!/bin/bash
# A/B of two builds of the library on ONE box: swaps csrc/libnsol_hip.so between runs
#   tools/_probe/ab_lib.sh <other.so> <command...>
set -e
OTHER=$1; shift
LIB=nsol_amd/csrc/libnsol_hip.so
cp $LIB /tmp/lib_base.so
for round in 1 2 3; do
  cp /tmp/lib_base.so $LIB; echo -n "base  "; "$@"
  cp $OTHER $LIB;           echo -n "other "; "$@"
done
cp /tmp/lib_base.so $LIB
