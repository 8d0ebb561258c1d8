#!/bin/bash
# usage: variant_any.sh <file stem, e.g. gemaps> <name> [extra hipcc flags...]  -> tools/ubench/build/libsmilehip_<name>.so
# A private copy of libsmilehip with opensmile_amd/csrc/lld_<stem>.hip compiled with experiment flags (-DSMILEHIP_PHASE_TIMING).
set -e
cd "$(dirname "$0")"
STEM=$1; NAME=$2; shift; shift
SRC=../../opensmile_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$SRC -I../../include"
mkdir -p build
/opt/rocm/bin/hipcc $F "$@" -c $SRC/lld_$STEM.hip -o build/lld_${STEM}_$NAME.o
OBJS=$(ls $SRC/*.o | grep -v "lld_$STEM.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsmilehip_$NAME.so build/lld_${STEM}_$NAME.o $OBJS
