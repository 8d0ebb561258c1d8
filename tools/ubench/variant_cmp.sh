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
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsmilehip_$NAME.so build/lld_compare_$NAME.o \
  $SRC/lld_f0.o $SRC/lld_mfcc512.o $SRC/lld_kernels.o $SRC/lld_stage_kernels.o $SRC/lld_stage2_kernels.o $SRC/lld_is09.o $SRC/lld_functionals.o $SRC/smilehip_core.o $SRC/smilehip_plan.o $SRC/smilehip_batch.o $SRC/smilehip_stage.o $SRC/tables.o
