#!/bin/bash
# usage: variant_any.sh <file stem, e.g. gemaps> <name> [extra hipcc flags...]  -> tools/ubench/build/libsmilehip_<name>.so
# A private copy of libsmilehip with opensmile_amd/csrc/lld_<stem>.hip compiled with the Makefile's own flags for that file plus
# the experiment's (-DSMILEHIP_PHASE_TIMING, ...).
set -e
cd "$(dirname "$0")"
STEM=$1; NAME=$2; shift; shift
SRC=../../opensmile_amd/csrc
mkdir -p build
CMD=$(make -C $SRC -n -B lld_$STEM.o 2>/dev/null | grep -m1 "hipcc.* -c ")
CMD=${CMD/ -o lld_$STEM.o/ -o $PWD/build/lld_${STEM}_$NAME.o}
(cd $SRC && $CMD "$@")
OBJS=$(ls $SRC/*.o | grep -v "lld_$STEM.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsmilehip_$NAME.so build/lld_${STEM}_$NAME.o $OBJS
