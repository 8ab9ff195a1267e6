#!/bin/bash
# ablation of the 3x3 conv kernel on the headline shapes: TRAINNER_CONV_DBG bits 1 = no MFMA, 2 = no global loads, 4 = no epilogue stores
for d in 0 1 2 4 6 7; do
  echo "== DBG=$d"
  TRAINNER_CONV_DBG=$d timeout -k 10 200 python tools/bench_conv.py conv 2>&1 | grep -v amdgpu.ids
done
