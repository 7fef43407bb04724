# round-4 evidence, one gpurun call: the whole GPU suite, the switch sweep, the profiles (run from the repo root)
mkdir -p gpurun_out/r04_final
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r04_final/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r04_final/gpu_tests.log
bash scripts/check_switches.sh > gpurun_out/r04_final/r04_check_switches.txt 2>&1; cat gpurun_out/r04_final/r04_check_switches.txt
bash scripts/make_profiles.sh r04 > gpurun_out/r04_final/make_profiles.log 2>&1; tail -30 gpurun_out/r04_final/make_profiles.log
