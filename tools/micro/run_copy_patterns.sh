#!/bin/bash
# copy_pattern over a few decompositions (n lxb rows halo_x halo_y zchunk nw [split])
cd tools/micro
hipcc --offload-arch=gfx950 -O3 -o copy_pattern copy_pattern.hip 2>/dev/null
for cfg in "512 66 11 4 2 103 12 1" "512 66 11 4 2 103 12 0" "512 128 6 0 0 128 12 0" "512 128 8 0 0 128 16 0" \
           "512 64 12 0 0 128 12 0" "512 128 4 0 0 64 8 0" "512 128 6 0 0 512 12 0" "512 128 6 0 0 32 12 0" \
           "512 128 2 0 0 128 4 0" "512 64 4 0 0 128 4 0"; do
  ./copy_pattern $cfg
done
