#!/bin/bash
# Builds libmultivolumes_amd.so (gfx950 only) in-tree.  Usage: build.sh [extra hipcc flags, e.g. -DMV_DIAGNOSTICS]
# Objects go to a directory of their own per flag set (obj/<hash of the flags>), are rebuilt from scratch on every call, and
# every compile job's exit status is checked: a translation unit that fails can never be replaced by a stale object.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
OUT=${MV_OUT:-../libmultivolumes_amd.so}
TAG=$(printf '%s' "$FLAGS $*" | md5sum | cut -c1-12)
OBJ=obj/$TAG
rm -rf "$OBJ"
mkdir -p "$OBJ"
pids=()
$HIPCC $FLAGS -c mv_kernels.hip -o $OBJ/mv_kernels.o "$@" & pids+=($!)
$HIPCC $FLAGS -c mv_march.hip -o $OBJ/mv_march.o "$@" & pids+=($!)
# (the tile march's measured-and-rejected variants -- LDS staging, speculation, s_setprio -- live in experiments/ and replace the product
# kernel only when one of their switches is on the command line: scripts/check_switches.sh, the A/B scripts)
TILES=mv_march_tiles.hip
case " $* " in *MV_TILE_STAGE*|*MV_TILE_SPECULATE_BELOW*|*MV_TILE_PRIORITY*) TILES=experiments/mv_march_tiles_experiments.hip;; esac
case " $* " in *MV_EMPTY_REGIONS*|*MV_EMULATE_LIGHT_PAIRS*|*MV_TILE_LDS_PAD*|*MV_DIAG_ITER*|*MV_TILE_DIET*) TILES=experiments/mv_march_tiles_r05.hip;; esac	# (round 5's: profiles/r05_instruction_cuts.txt)
TILES=${MV_TILES_SRC:-$TILES}	# (diagnostics: a patched copy of the tile march, e.g. scripts/r06_valu_pad.sh)
$HIPCC $FLAGS -I. -c $TILES -o $OBJ/mv_march_tiles.o "$@" & pids+=($!)
$HIPCC $FLAGS -c mv_peer.hip -o $OBJ/mv_peer.o "$@" & pids+=($!)
NARROW=""
case " $* " in *MV_TILE_NARROW_BELOW*) NARROW=$OBJ/mv_march_narrow.o; $HIPCC $FLAGS -I. -c experiments/mv_march_narrow.hip -o $NARROW "$@" & pids+=($!);; esac
$HIPCC $FLAGS -c mv_composite.hip -o $OBJ/mv_composite.o "$@" & pids+=($!)
$HIPCC $FLAGS -c mv_mesh.hip -o $OBJ/mv_mesh.o "$@" & pids+=($!)
$HIPCC $FLAGS -c mv_api.cpp -o $OBJ/mv_api.o "$@" & pids+=($!)
$HIPCC $FLAGS -c mv_dist.cpp -o $OBJ/mv_dist.o "$@" & pids+=($!)
$HIPCC $FLAGS -c mv_dist_peer.cpp -o $OBJ/mv_dist_peer.o "$@" & pids+=($!)
$HIPCC $FLAGS -c mv_mesh_api.cpp -o $OBJ/mv_mesh_api.o "$@" & pids+=($!)
g++ -O3 -ffp-contract=off -std=c++17 -fPIC -Wall -c mv_host.cpp -o $OBJ/mv_host.o & pids+=($!)
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
if [ $rc -ne 0 ]; then echo "build.sh: a compile job failed" >&2; rm -rf "$OBJ"; exit 1; fi
$HIPCC -shared -fPIC --offload-arch=gfx950 -o "$OUT.tmp" $OBJ/mv_kernels.o $OBJ/mv_march.o $OBJ/mv_march_tiles.o $NARROW $OBJ/mv_peer.o $OBJ/mv_composite.o $OBJ/mv_mesh.o $OBJ/mv_api.o $OBJ/mv_dist.o $OBJ/mv_dist_peer.o $OBJ/mv_mesh_api.o $OBJ/mv_host.o -lpthread
mv -f "$OUT.tmp" "$OUT"
# content hash of the sources this library was built from (multivolumes_amd/_lib.py::_source_hash computes the same):
# the loader rebuilds when the tree no longer matches it
srchash() {
	for d in . ../../include; do
		for f in $(cd "$d" && LC_ALL=C ls | grep -E '\.(hip|cpp|hpp|h|sh)$'); do printf '%s' "$f"; cat "$d/$f"; done
	done | sha256sum | cut -d' ' -f1
}
srchash > "$OUT.srchash"
rm -rf "$OBJ"
rmdir obj 2>/dev/null || true
echo "built $(realpath $OUT)"
