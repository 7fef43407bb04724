cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
VARIANTS="product: balance:-DMV_TILE_BALANCE=1" CONFIGS="C3 C5" bash scripts/r05_timeline.sh 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|c /= "
PMC=0 bash scripts/r05_ab.sh "product:" "balance:-DMV_TILE_BALANCE=1"
PMC=0 BENCH_ARGS="--config C5" bash scripts/r05_ab.sh "product:" "balance:-DMV_TILE_BALANCE=1"
