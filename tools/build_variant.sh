#!/bin/bash
# builds tools/_build/libmelogan_$1.so = product library with source file $2 recompiled with the remaining flags
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
C=melo-gan_amd/csrc
TAG=$1; SRC=$2; shift 2
EXTRA=""; [ "$SRC" = conv_mfma ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $EXTRA "$@" -c $C/$SRC.hip -o tools/_build/${SRC}_$TAG.o
OBJS=""
for f in runtime conv_mfma conv16_mfma conv_thin conv_bf16 linear_skinny wgrad_mfma small_kernels; do
  if [ "$f" = "$SRC" ]; then OBJS="$OBJS tools/_build/${SRC}_$TAG.o"; else OBJS="$OBJS $C/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o tools/_build/libmelogan_$TAG.so
echo built $TAG
