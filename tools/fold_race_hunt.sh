#!/bin/bash
# bash tools/fold_race_hunt.sh  (on the GPU box): one variant after the other, each bounded
mkdir -p gpurun_out
for v in ${VARIANTS:-logits_fold logits_plain swiglu_fold heads_fold}; do
  timeout ${TMO:-330} python3 tools/fold_race_hunt.py $v ${ROUNDS:-20000} 3 8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/race_hunt.log | tail -40
done
