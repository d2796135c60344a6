#!/bin/bash
# Back in the container after `gpurun -- 'bash profiles/collect.sh r04'`: copy the summaries to be judged into profiles/ (gpurun_out/ is scratch), regenerate the ISA mix
# of the shipped build and summarise the counter passes.      bash profiles/finish.sh r04
set -e
shopt -s expand_aliases 2>/dev/null || true
R=${1:-r06}
cd "$(dirname "$0")/.."
cp gpurun_out/prof_$R/${R}_kernel_stats.csv gpurun_out/prof_$R/${R}_kernel_trace.csv profiles/
cp gpurun_out/prof_${R}s/${R}s_kernel_stats.csv profiles/ 2>/dev/null || true
cp gpurun_out/prof_${R}x/${R}x_kernel_stats.csv profiles/ 2>/dev/null || true
cp gpurun_out/prof_${R}i/${R}i_kernel_stats.csv profiles/ 2>/dev/null || true                      # the default run: two batches in flight
cp gpurun_out/prof_${R}c4/${R}c4_kernel_stats.csv profiles/ 2>/dev/null || true                    # one GPU's share of config 4 under the 8-way split: 2^19 V2 verifies
[ -s gpurun_out/bench_${R}_c4share.log ] && tail -1 gpurun_out/bench_${R}_c4share.log > profiles/${R}c4_bench_line.json
cp gpurun_out/host_trace_$R.txt profiles/${R}_e2e_host_trace.txt 2>/dev/null || true
[ -s gpurun_out/bench_$R.log ] && tail -1 gpurun_out/bench_$R.log > profiles/${R}_bench_line.json
[ -s gpurun_out/bench_${R}_sign.log ] && tail -1 gpurun_out/bench_${R}_sign.log > profiles/${R}_bench_line_sign.json
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S zk-nullifier-sig_amd/csrc/plume_kernels.hip -o /tmp/plume_k.s 2>/dev/null
{
  echo "# ISA-level instruction mix of the hot loop bodies, $R build (python profiles/isa_mix.py <hipcc -S listing> <kernel> <min block size>); a block ends at a label or a branch"
  echo "# est_ns/wave prices multiply-adds at 1.83 ns, other VOP3 at 1.77, plain VOP1/VOP2 at 1.05, s_nop at 0 (tests/gpu_debug/instr_rates_r03.txt)"
  python3 profiles/isa_mix.py /tmp/plume_k.s k_verify_msm_s 600
  python3 profiles/isa_mix.py /tmp/plume_k.s k_verify_msm 600
  python3 profiles/isa_mix.py /tmp/plume_k.s k_verify_scalars 300
  python3 profiles/isa_mix.py /tmp/plume_k.s k_sign_hmul 600
  python3 profiles/isa_mix.py /tmp/plume_k.s k_verify_ingest 600
  python3 profiles/isa_mix.py /tmp/plume_k.s k_tab_pass_b 300
} > profiles/${R}_isa_mix.txt
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c zk-nullifier-sig_amd/csrc/plume_kernels.hip -o /dev/null 2> /tmp/plume_remarks.txt || true
python3 profiles/kernel_resources.py /tmp/plume_remarks.txt > profiles/${R}_kernel_resources.txt || true
python3 profiles/summarize_pmc.py $R
