#!/bin/bash
# Dev: build a variant of the Winograd kernel into build_abl/libntmtrack_<tag>.so (other objects from the product
# build).  usage: scripts/dev_wino_variant.sh <tag> <extra hipcc flags...>
# Time on the GPU box:  NTK_LIB_PATH=build_abl/libntmtrack_<tag>.so python scripts/dev_wino.py 640
set -e
tag=$1; shift
cd "$(dirname "$0")/../ntm-tracker_amd/csrc"
mkdir -p ../../build_abl
OTHERS=$(ls build/*.o | grep -v conv_wino.o)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c conv_wino.hip -o ../../build_abl/conv_wino_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_abl/libntmtrack_$tag.so $OTHERS ../../build_abl/conv_wino_$tag.o
rm ../../build_abl/conv_wino_$tag.o
