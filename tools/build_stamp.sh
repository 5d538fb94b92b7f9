#!/bin/bash
# builds tools/_build/libmelogan_stamp$1.so = product library with -DMG_STAMPS $2.. on the conv kernel
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
C=melo-gan_amd/csrc
TAG=$1; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DMG_STAMPS -fno-slp-vectorize "$@" -c $C/conv_mfma.hip -o tools/_build/conv_stamp$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/runtime.o tools/_build/conv_stamp$TAG.o $C/conv16_mfma.o $C/conv_thin.o $C/conv_bf16.o $C/linear_skinny.o $C/wgrad_mfma.o $C/small_kernels.o -o tools/_build/libmelogan_stamp$TAG.so
echo built $TAG
