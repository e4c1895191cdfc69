#!/bin/bash
# Dev: build a variant of a Winograd kernel into build_abl/libntmtrack_<tag>.so (other objects from the product build).
# usage: [SRC=conv_wino43] scripts/dev_wino_variant.sh <tag> <extra hipcc flags...>     (SRC defaults to conv_wino)
# Time on the GPU box:  NTK_LIB_PATH=build_abl/libntmtrack_<tag>.so python scripts/dev_wino43.py 640
set -e
tag=$1; shift
SRC=${SRC:-conv_wino}
cd "$(dirname "$0")/../ntm-tracker_amd/csrc"
mkdir -p ../../build_abl
OTHERS=$(ls build/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $SRC.hip -o ../../build_abl/variant_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_abl/libntmtrack_$tag.so $OTHERS ../../build_abl/variant_$tag.o
rm ../../build_abl/variant_$tag.o
