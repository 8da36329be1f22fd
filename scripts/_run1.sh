set -e
mkdir -p gpurun_out/r03s
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "selftest or orbit or step_sweep or ragged or full_size" > gpurun_out/r03s/pytest1.log 2>&1 || { tail -30 gpurun_out/r03s/pytest1.log; exit 1; }
tail -3 gpurun_out/r03s/pytest1.log
ROWS="base teapot 4k" timeout -k 10 400 bash scripts/lib_ab_rows.sh r03s/ab libvolym_hip_base.so libvolym_hip.so
timeout -k 10 300 bash scripts/pmc_ablate.sh > gpurun_out/r03s/pmc_ablate.txt 2>&1; cat gpurun_out/r03s/pmc_ablate.txt
