#!/bin/bash
# product library + the opt-in variant library + (optionally) an A/B variant of conv_s16.hip: build_all.sh [name flags...]
cd "$(dirname "$0")/../../cer-mvs_amd/csrc" || exit 1
make -j3 2>&1 | grep -i "error\|warning: loop" | head
make variants/libcermvs_optin.so 2>&1 | grep -i "error" | head
if [ -n "$1" ]; then
  name=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wall -Wno-unused-function "$@" -c conv_s16.hip -o variants/conv_s16_$name.o 2>&1 | grep -i error
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "^conv_s16.o$") variants/conv_s16_$name.o -o variants/libcermvs_$name.so
fi
echo build_all done
