#!/bin/bash
# ablation of the Linear (ksize 1) kernel: TRAINNER_CONV_DBG bits 1 = no MFMA, 2 = no global loads, 4 = no epilogue stores
for d in 0 1 2 4 3 6 7; do
  echo "== DBG=$d"
  TRAINNER_CONV_DBG=$d timeout -k 10 120 python tools/bench_linear.py conv 2>&1 | grep -E "540 cdout|180 f32out\+res  "
done
