#!/bin/bash
# compile-time ablations of the fp32 conv kernel (GPU box): tools/ablate.sh "<shape ids>"
# needs the diagnosis build:  make -C contextaware-poseformer_amd/csrc DIAG=1  (-> tools/ab/libcapf_diag.so, run with CAPF_LIB=tools/ab/libcapf_diag.so)
# CAPF_ABLATE: 0 product, 1 no DMA in the K loop, 3 no LDS fragment reads, 4 no vmcnt/barrier, 5 no epilogue,
#              6 = 1+3+4 (MFMA + address VALU only), 2 no MFMA
for i in ${1:-1 2 5}; do
  for ab in 0 1 3 4 5 6 2; do
    echo -n "abl=$ab "; CAPF_ABLATE=$ab python tools/bench_conv.py --only $i 2>&1 | grep "^\["
  done
done
