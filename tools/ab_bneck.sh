#!/bin/bash
# Knock-out builds of the fused bottleneck (csrc/bneck_bf16.hip, BN_EXP bits: 1 no conv2 loop, 2 no conv3 / shortcut MFMAs, 4 one of four
# epilogue blocks, 8 no halo loads, 16 no y stores) as tools/ab/libcapf_bn<e>.so; run from the repo root AFTER make (links build/*.o).
# Timing only: the results of these builds are wrong by construction.   CAPF_LIB=tools/ab/libcapf_bn1.so python tools/launch_table.py ...
cd contextaware-poseformer_amd/csrc
mkdir -p ../../tools/ab
for e in "$@"; do
  hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -I../../include -I. -DBN_EXP=$e -x hip -c bneck_bf16.hip -o /tmp/bneck_$e.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=capf.map -o ../../tools/ab/libcapf_bn$e.so $(ls build/*.o | grep -v bneck_bf16) /tmp/bneck_$e.o || exit 1
done
