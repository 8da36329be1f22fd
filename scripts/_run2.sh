set -e
ROWS="base teapot 4k" timeout -k 10 600 bash scripts/lib_ab_rows.sh r03s/ab libvolym_hip_base.so libvolym_hip.so
