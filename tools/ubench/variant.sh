#!/bin/bash
# usage: variant.sh <name> [extra hipcc flags...]  -> tools/ubench/build/libsmilehip_<name>.so
# A private copy of libsmilehip whose fast kernel is compiled with experiment flags
# (-DSMILEHIP_PHASE_TIMING, -DSMILEHIP_DEBUG_*). Run with SMILEHIP_LIB=<path> python bench.py.
set -e
cd "$(dirname "$0")"
NAME=$1; shift
SRC=../../opensmile_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$SRC -I../../include"
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -Xclang -target-feature -Xclang -load-store-opt "$@" -c $SRC/lld_mfcc512.hip -o build/lld_mfcc512_$NAME.o
OBJS=$(ls $SRC/*.o | grep -v lld_mfcc512.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsmilehip_$NAME.so build/lld_mfcc512_$NAME.o $OBJS
