#!/bin/bash
# Dev: build ablated variants of the Winograd kernel (conv_wino.hip -DWINO_ABL=<mask>, see the source) into
# build_abl/libntmtrack_abl<mask>.so, re-using the other objects of the product build.  Run here (hipcc cross
# compiles), then time on the GPU box:  NTK_LIB_PATH=build_abl/libntmtrack_abl1.so python scripts/dev_wino.py 640
set -e
cd "$(dirname "$0")/../ntm-tracker_amd/csrc"
mkdir -p ../../build_abl
OTHERS=$(ls build/*.o | grep -v conv_wino.o)
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DWINO_ABL=$m -c conv_wino.hip -o ../../build_abl/conv_wino_abl$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_abl/libntmtrack_abl$m.so $OTHERS ../../build_abl/conv_wino_abl$m.o
  rm ../../build_abl/conv_wino_abl$m.o
done
