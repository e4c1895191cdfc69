#!/bin/bash
# GPU box: the split form's ablation builds (SRC=conv_bf16p scripts/dev_wino_variant.sh p<bits> -DBF16P_ABL=<bits>) per layer:
# p1 no DMA after the prologue, p2 no barrier / DMA wait, p4 fragments read once per stage, p8 no MFMAs
for v in "" $*; do
  if [ -z "$v" ]; then echo "== product"; LIB=""; else echo "== ablation $v"; LIB="NTK_LIB_PATH=build_abl/libntmtrack_$v.so"; fi
  env $LIB timeout -k 10 200 python scripts/r04/split3_layers.py 640 only3 2>&1 | grep -v amdgpu.ids | sed 's/ winograd43.*split3 / err /; s/ winograd43.*//'
done
