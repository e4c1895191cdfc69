#!/bin/bash
# GPU box: the patch-form bf16 kernel's ablation builds (scripts/dev_wino_variant.sh p<bits> -DBF16P_ABL=<bits>) per layer
mkdir -p gpurun_out/r04_bf
for v in "" p1 p2 p4 p8 p12; do
  if [ -z "$v" ]; then echo "== product"; LIB=""; else echo "== ablation $v"; LIB="NTK_LIB_PATH=build_abl/libntmtrack_$v.so"; fi
  env $LIB timeout -k 10 200 python scripts/r04/bf16_trunk.py 640 2>&1 | grep -v amdgpu.ids | grep "patch" | sed 's/tile [0-9.]* ms ([0-9]* TF), //'
done
