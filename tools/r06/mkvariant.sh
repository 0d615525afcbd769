#!/bin/bash
# usage: tools/archive/r05/mkvariant.sh <name> <source.hip> <flags...>  -> cer-mvs_amd/csrc/variants/libcermvs_<name>.so
# ONE source of the product library recompiled with extra -D flags, linked with the product's other objects (A/B runs: CER_MVS_LIB=...)
set -e
name=$1; src=$2; shift; shift
cd "$(dirname "$0")/../../cer-mvs_amd/csrc"
mkdir -p variants
base=$(basename $src .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wall -Wno-unused-function "$@" -c $src -o variants/${base}_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "^$base.o$") variants/${base}_$name.o -o variants/libcermvs_$name.so
echo "variants/libcermvs_$name.so"
