#!/bin/bash
mkdir -p gpurun_out/r05b
F='^===|runtime:|HSA Copy|Blit|copy engine|copyBuffer|staging|ShaderName'
AMD_LOG_LEVEL=4 python tests/gpu_debug/d2h_in_library.py 2>&1 | grep -E "$F" | sed -n '/MARK sign begin/,/MARK sign end/p' | cut -c1-260 > gpurun_out/r05b/log_torch_runtime.txt
PLUME_NO_TORCH_PRELOAD=1 AMD_LOG_LEVEL=4 python tests/gpu_debug/d2h_in_library.py 2>&1 | grep -E "$F" | sed -n '/MARK sign begin/,/MARK sign end/p' | cut -c1-260 > gpurun_out/r05b/log_rocm_runtime.txt
python tests/gpu_debug/d2h_in_library.py 1048576 2>&1 | grep -E "runtime|MARK" > gpurun_out/r05b/time_torch.txt
PLUME_NO_TORCH_PRELOAD=1 python tests/gpu_debug/d2h_in_library.py 1048576 2>&1 | grep -E "runtime|MARK" > gpurun_out/r05b/time_rocm.txt
wc -l gpurun_out/r05b/*; cat gpurun_out/r05b/time_*.txt
