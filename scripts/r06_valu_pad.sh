#!/bin/bash
# round 6: how the C3 frame answers to INSTRUCTION COUNT alone -- the product tile march with n extra `v_add_f32` in every sample iteration that
# fetches (a patched copy of csrc/mv_march_tiles.hip, built per variant through MV_TILES_SRC; results are unchanged: the adds feed nothing).
# Beside it profiles/r06_colocate.txt says how the frame answers to BYTES alone.  usage: scripts/r06_valu_pad.sh [pads = "0 16 32 64"]
cd /root/repo
python3 - <<'PY'
src = open("multivolumes_amd/csrc/mv_march_tiles.hip").read()
anchor = "\t\t\t\tColourFootprint cf = fetchColour(vol, fa);"
assert src.count(anchor) == 1
pad = """#if MV_VALU_PAD
\t\t\t\t{
\t\t\t\t\tfloat pad = t;
#pragma unroll
\t\t\t\t\tfor (int k = 0; k < MV_VALU_PAD; ++k) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(pad));
\t\t\t\t\tasm volatile("" :: "v"(pad));
\t\t\t\t}
#endif
"""
open("multivolumes_amd/csrc/experiments/_tiles_valu_pad.hip", "w").write(src.replace(anchor, pad + anchor))
PY
export MV_TILES_SRC=experiments/_tiles_valu_pad.hip
v=()
for n in ${1:-0 16 32 64}; do v+=("pad$n:-DMV_VALU_PAD=$n"); done
bash scripts/r05_ab.sh "${v[@]}" 2>&1 | tee gpurun_out/r06_valu_pad.txt
rm -f multivolumes_amd/csrc/experiments/_tiles_valu_pad.hip
