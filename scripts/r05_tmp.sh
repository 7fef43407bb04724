cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
PMC=0 bash scripts/r05_ab.sh "product:" "chunk8:-DMV_TILE_XCD_CHUNK=8" "blocked:-DMV_TILE_BLOCKED=1" "blocked8:-DMV_TILE_BLOCKED=1 -DMV_TILE_XCD_CHUNK=8" "blocked16:-DMV_TILE_BLOCKED=1 -DMV_TILE_XCD_CHUNK=16" "blocked32:-DMV_TILE_BLOCKED=1 -DMV_TILE_XCD_CHUNK=32"
for v in product chunk8; do :; done
