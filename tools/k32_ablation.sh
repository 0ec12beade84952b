#!/bin/bash
# Ablations of the K = 32 three-stage two-workgroup GEMM (needs tools/ab_variants/gemm2c_k32.patch applied to paintmind_amd/csrc/gemm2b.hip:
#   git apply tools/ab_variants/gemm2c_k32.patch).  gpurun -- 'bash tools/k32_ablation.sh'; results: profiles/r05_b_*.
run() { PM_EXTRA_FLAGS="$1" bash paintmind_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed [$1]"; return; }; echo "== [$1] K32=$2"; PMHIP_G2B_K32=$2 PMHIP_G256_RES_KMIN=4096 python tools/producer_bench.py 2>&1 | grep "M=65536" ; }
E="-DPM_ABL_NO_RESLOAD -DPM_ABL_NO_STORE"
run "$E" 0
run "$E" 1
run "$E -DPM_K32_ABL=1" 1
run "$E -DPM_K32_ABL=2" 1
run "$E -DPM_K32_ABL=4" 1
run "$E -DPM_K32_ABL=8" 1
run "$E -DPM_K32_ABL=7" 1
bash paintmind_amd/csrc/build.sh > /dev/null 2>&1
