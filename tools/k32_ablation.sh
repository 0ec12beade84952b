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
