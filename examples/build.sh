#!/bin/bash
# Builds the C++ host examples against the in-tree library: examples/frame_cpp.bin (one frame through csrc/MultiRayCaster.hpp, the
# reference-side mirror) and examples/dist_rank_cost.bin (what a rank of the multi-GPU loop costs per frame without a fabric).
# Called by __graft_entry__.build(); the binaries are git-ignored and travel to the GPU box like the library.
set -euo pipefail
cd "$(dirname "$0")/.."
for e in frame_cpp dist_rank_cost; do
  g++ -O2 -std=c++17 examples/$e.cpp -Lmultivolumes_amd -lmultivolumes_amd -Wl,-rpath,"$PWD/multivolumes_amd" -o examples/$e.bin
done
echo "built examples/frame_cpp.bin examples/dist_rank_cost.bin"
