#!/bin/bash
# usage: variant_f0.sh <name> [extra hipcc flags...]  -> tools/ubench/build/libsmilehip_<name>.so
# A private copy of libsmilehip whose F0 kernels are compiled with experiment flags (-DSMILEHIP_PHASE_TIMING).
set -e
cd "$(dirname "$0")"
NAME=$1; shift
SRC=../../opensmile_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$SRC -I../../include"
mkdir -p build
/opt/rocm/bin/hipcc $F "$@" -c $SRC/lld_f0.hip -o build/lld_f0_$NAME.o
OBJS=$(ls $SRC/*.o | grep -v lld_f0.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsmilehip_$NAME.so build/lld_f0_$NAME.o $OBJS
