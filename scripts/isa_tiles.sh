#!/bin/bash
# Compiles csrc/mv_march_tiles.hip alone (CPU, cross-compile) with the given extra flags, keeps the ISA in /tmp/tiles_isa/ and prints the
# registers of every kernel in it: scripts/isa_tiles.sh [-DFLAG ...]
set -e
mkdir -p /tmp/tiles_isa && cd /tmp/tiles_isa
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I/root/repo/multivolumes_amd/csrc -c /root/repo/multivolumes_amd/csrc/mv_march_tiles.hip -o tiles.o -save-temps=obj "$@" 2>&1 | grep -E "error|warning: v" | head
f=/tmp/tiles_isa/mv_march_tiles-hip-amdgcn-amd-amdhsa-gfx950.s
grep -A12 "\.name:.*march" $f | grep -E "\.name|vgpr_count|sgpr_count|spill|private_segment_fixed" | paste - - - - - - | awk '{printf "%-100s scratch %s sgpr %s sgpr_spill %s vgpr %s vgpr_spill %s\n", substr($2,1,100), $4, $6, $8, $10, $12}'
awk '/^_ZN2mv16marchTilesKernelILb0ELb1ELb0ELb0EEEvNS_11DeviceStateEj:/,/s_endpgm/' $f > /tmp/tiles_isa/main_kernel.s
