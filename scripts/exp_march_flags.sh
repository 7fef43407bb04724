#!/bin/bash
# A/B of compiler options on the tile march alone (mv_march_tiles.hip; everything else as build.sh compiles it): builds
# multivolumes_amd/libmv_flag_<n>.so per option set (run here), then `bash scripts/exp_march_flags.sh run` through gpurun times them.
set -u
cd "$(dirname "$0")/.."
SETS=(
  ""
  "-mllvm -amdgpu-schedule-relaxed-occupancy=true"
  "-mllvm -amdgpu-schedule-relaxed-occupancy=true -mllvm -enable-post-misched=false"
  "-mllvm -amdgpu-schedule-relaxed-occupancy=true -mllvm -amdgpu-schedule-metric-bias=0"
  "-mllvm -amdgpu-schedule-relaxed-occupancy=true -mllvm -amdgpu-schedule-metric-bias=50"
  "-mllvm -amdgpu-disable-unclustered-high-rp-reschedule=true"
  "-mllvm -amdgpu-schedule-relaxed-occupancy=true -mllvm -amdgpu-disable-unclustered-high-rp-reschedule=true"
  "-mllvm -amdgpu-schedule-relaxed-occupancy=true -fno-unroll-loops"
  "-mllvm -amdgpu-schedule-relaxed-occupancy=true -mllvm -amdgpu-sched-strategy=max-memory-clause"
)
if [ "${1:-build}" = run ]; then
  for i in "${!SETS[@]}"; do
    lib=multivolumes_amd/libmv_flag_$i.so
    [ -f $lib ] || continue
    for rep in 1 2; do
      MV_LIB_PATH=$lib python bench.py --no-cpu-baseline --steps 600 --warmup 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$i [${SETS[$i]}]', round(d['ms_per_step'],4), 'launch', round(d['roofline']['launch_ms'],4))"
    done
  done
  exit 0
fi
HIPCC=/opt/rocm/bin/hipcc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
T=/tmp/mv_flag_objs; rm -rf $T; mkdir -p $T
cd multivolumes_amd/csrc
for f in mv_kernels mv_march mv_composite mv_mesh; do $HIPCC $FLAGS -c $f.hip -o $T/$f.o & done
$HIPCC $FLAGS -c mv_api.cpp -o $T/mv_api.o &
g++ -O2 -std=c++17 -fPIC -Wall -c mv_host.cpp -o $T/mv_host.o &
wait
for i in "${!SETS[@]}"; do
  if $HIPCC $FLAGS ${SETS[$i]} -c mv_march_tiles.hip -o $T/mt_$i.o 2> $T/err_$i.txt; then
    $HIPCC -shared -fPIC --offload-arch=gfx950 -o ../libmv_flag_$i.so $T/mv_kernels.o $T/mv_march.o $T/mt_$i.o $T/mv_composite.o $T/mv_mesh.o $T/mv_api.o $T/mv_host.o -lpthread && echo "built $i: ${SETS[$i]}"
  else
    echo "FAILED $i: ${SETS[$i]}: $(tail -2 $T/err_$i.txt | tr '\n' ' ')"
  fi
done
