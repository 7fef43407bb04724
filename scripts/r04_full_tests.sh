# the whole GPU suite (run from the repo root, one gpurun call)
mkdir -p gpurun_out/r04t
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -m gpu -q --durations=15 -x > gpurun_out/r04t/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -30 gpurun_out/r04t/gpu_tests.log
