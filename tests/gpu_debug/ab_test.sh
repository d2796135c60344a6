# correctness of build B first (golden + edge + sign through the GPU tests), then A/B timing
PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_b.so python -m pytest tests -m gpu -q -x -k "golden or fixed_vector or edge or fuzz or config2" 2>&1 | tail -2
bash tests/gpu_debug/ab.sh
