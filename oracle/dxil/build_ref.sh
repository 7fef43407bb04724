#!/bin/bash
# Builds oracle/_ref/libmv_dxref.so: the reference's shipped compiled shaders (/root/reference/Bin/*.cso) translated to x86-64 through
# LLVM (cso2ll.py) + the runtime that answers their dx.op intrinsics (dxil_rt.c).  Run by `make -C oracle ref` where /root/reference
# exists (the build container); the GPU box only uses the prebuilt library.  Outputs only under oracle/_ref/ (git-ignored).
set -e
cd "$(dirname "$0")/.."
REF=${MV_REFERENCE:-/root/reference}
CLANG=${CLANG:-/opt/rocm/lib/llvm/bin/clang}
[ -d "$REF/Bin" ] || { echo "build_ref.sh: $REF/Bin not found: nothing to build"; exit 0; }
mkdir -p _ref
objs=""
for s in CSRayMarchV CSRayMarchL CSVolumeCull CSInitGridData CSR32FToRGBA16F VSCube PSDepthPeel PSCube PSResolveOIT VSDepth VSBasePass; do
  for mode in fp32 fp16; do
    python3 dxil/cso2ll.py "$REF/Bin/$s.cso" "dxil_${s}_${mode}" $mode "_ref/${s}_${mode}.ll" "_ref/${s}.dxil.ll"
    "$CLANG" -O1 -fPIC -mf16c -c "_ref/${s}_${mode}.ll" -o "_ref/${s}_${mode}.o" -Wno-override-module
    objs="$objs _ref/${s}_${mode}.o"
  done
done
# the work-graph library: two node functions in one module (cso2ll.py translate_lib)
for mode in fp32 fp16; do
  python3 dxil/cso2ll.py "$REF/Bin/LibRayMarch.cso" "dxil_LibRayMarch" $mode "_ref/LibRayMarch_${mode}.ll" "_ref/LibRayMarch.dxil.ll" > /dev/null
  "$CLANG" -O1 -fPIC -mf16c -c "_ref/LibRayMarch_${mode}.ll" -o "_ref/LibRayMarch_${mode}.o" -Wno-override-module
  objs="$objs _ref/LibRayMarch_${mode}.o"
done
# (no -lmv_oracle: the runtime answers every resource intrinsic itself, dxil/d3d_sampler.c)
"$CLANG" -O2 -fPIC -ffp-contract=off -shared -o _ref/libmv_dxref.so dxil/dxil_rt.c dxil/d3d_sampler.c $objs -lm -lpthread
echo "built oracle/_ref/libmv_dxref.so"
