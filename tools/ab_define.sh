#!/bin/bash
# A/B builds of ONE translation unit with a macro set: bash tools/ab_define.sh <file.hip> <MACRO> <value> [<value> ...]  (repo root, after make)
#   -> tools/ab/libcapf_<MACRO><value>.so = the product objects with that file rebuilt under -D<MACRO>=<value>; select with CAPF_LIB=...
# (csrc/bneck_bf16.hip BN_EXP knock-outs: 1 no conv2 loop, 2 no conv3 / shortcut MFMAs, 4 one of four epilogue blocks, 8 no halo loads, 16 no y
#  stores, 32 direct 8-byte stores -- timing only; csrc/lifter_fused.hip CTX_MIN_BLOCKS: resident blocks per CU the compiler must allow)
SRC=$1; MAC=$2; shift; shift
cd contextaware-poseformer_amd/csrc
mkdir -p ../../tools/ab
B=$(basename $SRC .hip)
FP=""; case $B in lifter|lifter_fused|lifter_chain|preprocess) FP="-ffp-contract=off";; esac
for e in "$@"; do
  hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -I../../include -I. $FP -D$MAC=$e -x hip -c $B.hip -o /tmp/${B}_$e.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=capf.map -o ../../tools/ab/libcapf_$MAC$e.so $(ls build/*.o | grep -v "/$B.o") /tmp/${B}_$e.o || exit 1
done
