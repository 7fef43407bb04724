#!/bin/bash
# CPU sanitizer pass (GPU AddressSanitizer is not available on this pool): the oracle and the host helpers of the product library
# (mv_host.cpp: DDS / OBJ loaders, frame constants, camera math) built with AddressSanitizer + UndefinedBehaviorSanitizer by g++,
# everything else of the library unchanged, and the CPU test suite run against them.
# Output: gpurun_out-free; prints the pytest tail and any sanitizer report.  Usage: bash scripts/sanitize_cpu.sh [pytest args]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/oracle/_san"
mkdir -p "$OUT"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
# the oracle
g++ -O1 -std=c++17 -fPIC -ffp-contract=off -mfma -fopenmp $SAN -shared -o "$OUT/libmv_oracle.so" "$ROOT/oracle/mv_oracle.cpp" "$ROOT/oracle/mv_oracle_mesh.cpp"
# the product library: device objects by hipcc as always, host objects by g++ with the sanitizers
cd "$ROOT/multivolumes_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
pids=()
for f in mv_kernels mv_march mv_march_tiles mv_peer mv_composite mv_mesh; do
	$HIPCC $FLAGS -I. -c $f.hip -o "$OUT/$f.o" & pids+=($!)
done
for f in mv_api mv_dist mv_dist_peer mv_mesh_api; do	# (they include device headers: not g++ translation units; without a GPU they only ever reject arguments)
	$HIPCC $FLAGS -c $f.cpp -o "$OUT/$f.o" & pids+=($!)
done
g++ -O1 -std=c++17 -fPIC -Wall $SAN -c mv_host.cpp -o "$OUT/mv_host.o" & pids+=($!)
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
[ $rc -eq 0 ] || { echo "sanitize_cpu.sh: a compile job failed" >&2; exit 1; }
g++ -shared -fPIC $SAN -o "$OUT/libmultivolumes_amd.so" "$OUT"/mv_kernels.o "$OUT"/mv_march.o "$OUT"/mv_march_tiles.o "$OUT"/mv_peer.o "$OUT"/mv_composite.o "$OUT"/mv_mesh.o \
	"$OUT/mv_api.o" "$OUT/mv_dist.o" "$OUT/mv_dist_peer.o" "$OUT/mv_mesh_api.o" "$OUT/mv_host.o" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread -ldl
cd "$ROOT"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:alloc_dealloc_mismatch=0:halt_on_error=1:abort_on_error=0"
export UBSAN_OPTIONS="print_stacktrace=1"
export MV_LIB_PATH="$OUT/libmultivolumes_amd.so" MV_ORACLE_LIB="$OUT/libmv_oracle.so"
python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"

# Second pass: the runtime of the compiled-reference library (oracle/dxil: resource intrinsics, wave rendezvous, the work-graph scheduler,
# the independent sampler).  It needs _Float16, which this image's gcc lacks on x86: built by ROCm's clang with ITS sanitizer runtime,
# linked with the translated shaders' objects as `make -C oracle ref` left them, and run on its own (two ASan runtimes cannot share a
# process): the oracle and the product library unsanitized here.
if ls "$ROOT"/oracle/_ref/*_fp32.o > /dev/null 2>&1; then
	CLANG=${CLANG:-/opt/rocm/lib/llvm/bin/clang}
	"$CLANG" -O1 -fPIC -mf16c -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -shared-libsan -shared \
		-o "$OUT/libmv_dxref.so" "$ROOT/oracle/dxil/dxil_rt.c" "$ROOT/oracle/dxil/d3d_sampler.c" "$ROOT"/oracle/_ref/*_fp32.o "$ROOT"/oracle/_ref/*_fp16.o -lm -lpthread
	unset MV_LIB_PATH MV_ORACLE_LIB
	export LD_PRELOAD="$("$CLANG" -print-file-name=libclang_rt.asan-x86_64.so)"
	export MV_DXREF_LIB="$OUT/libmv_dxref.so"
	python -m pytest tests/test_oracle_vs_reference_dxil.py -q -p no:cacheprovider "$@"
fi
