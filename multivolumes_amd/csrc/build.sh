#!/bin/bash
# Builds libmultivolumes_amd.so (gfx950 only) in-tree.  Usage: build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
OUT=../libmultivolumes_amd.so
$HIPCC $FLAGS -c mv_kernels.hip -o mv_kernels.o "$@" &
$HIPCC $FLAGS -c mv_composite.hip -o mv_composite.o "$@" &
$HIPCC $FLAGS -c mv_mesh.hip -o mv_mesh.o "$@" &
$HIPCC $FLAGS -c mv_api.cpp -o mv_api.o "$@" &
g++ -O2 -std=c++17 -fPIC -Wall -c mv_host.cpp -o mv_host.o &
wait
$HIPCC -shared -fPIC --offload-arch=gfx950 -o $OUT mv_kernels.o mv_composite.o mv_mesh.o mv_api.o mv_host.o -lpthread
echo "built $(realpath $OUT)"
