#!/bin/bash
# usage: variant_cmp.sh <name> [extra hipcc flags...]  -> tools/ubench/build/libsmilehip_<name>.so
# A private copy of libsmilehip whose ComParE A+B kernels are compiled with experiment flags (-DSMILEHIP_PHASE_TIMING).
set -e
cd "$(dirname "$0")"
NAME=$1; shift
SRC=../../opensmile_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$SRC -I../../include"
mkdir -p build
/opt/rocm/bin/hipcc $F "$@" -c $SRC/lld_compare.hip -o build/lld_compare_$NAME.o
# every other object of the product library as the Makefile built it (make -C opensmile_amd/csrc first)
OBJS=$(cd $SRC && ls *.o | grep -v '^lld_compare.o$' | sed "s#^#$SRC/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsmilehip_$NAME.so build/lld_compare_$NAME.o $OBJS
