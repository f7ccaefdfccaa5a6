#!/bin/bash
# Builds variants of libmiopal.so with extra -D flags for A/B timing inside ONE gpurun call
# (boxes differ by several per cent, so variants are only compared on the same box):
#   tools/ab_build.sh NAME "-DMIOPAL_PAIR_AHEAD=1 -DMIOPAL_PAIR_XCOL=0"   -> variants/libmiopal_NAME.so
# Use with MIOPAL_LIBRARY=variants/libmiopal_NAME.so (pyopal_amd/_capi.py).
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
build=$root/variants/build_$name
mkdir -p "$build"
cp "$root"/pyopal_amd/csrc/*.hip "$root"/pyopal_amd/csrc/*.h "$root"/pyopal_amd/csrc/Makefile "$build"/
mkdir -p "$root/variants/include_link"
make -s -j8 -C "$build" OUT="$root/variants/libmiopal_$name.so" \
  CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I$root/pyopal_amd/csrc $flags" \
  "$root/variants/libmiopal_$name.so"
