set -e
mkdir -p gpurun_out/r03b
python -m pytest tests/test_gpu_configs.py tests/test_mgpu.py tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r03b/pytest_subset.log 2>&1 || (tail -30 gpurun_out/r03b/pytest_subset.log; exit 1)
tail -3 gpurun_out/r03b/pytest_subset.log
export VOLYM_HIP_LIB=$PWD/volym_amd/libvolym_hip_dev.so
for opt in "" "117=2"; do
  echo "== VOLYM_DEV_OPTS=$opt 1080p"; VOLYM_DEV_OPTS=$opt python scripts/moving_view.py --frames 400 --degrees 0.0 0.25 1.0 --in-flight 3 2>&1 | grep "in flight <= 3"
  echo "== VOLYM_DEV_OPTS=$opt 4K"; VOLYM_DEV_OPTS=$opt python scripts/moving_view.py --frames 300 --degrees 0.0 1.0 --in-flight 3 --width 3840 --height 2160 2>&1 | grep "in flight <= 3"
done
