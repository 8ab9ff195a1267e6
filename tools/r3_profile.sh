#!/bin/bash
# Round-3 profile collection on the GPU box (run through gpurun from the repo root): kernel-trace stats and PMC passes of the bench workloads.
# Counters in their own runs with --kernel-trace only; the program itself right after `--` (MI355X pool rules).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_prof
mkdir -p $O
B="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-roofline --no-loader"
which=${1:-all}
if [ "$which" = all ] || [ "$which" = headline ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_headline -- python3 $B > $O/stats_headline.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/mfma_headline -- python3 $B > $O/mfma_headline.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_headline -- python3 $B > $O/fetch_headline.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_headline -- python3 $B > $O/write_headline.log 2>&1
fi
if [ "$which" = all ] || [ "$which" = others ]; then
  for cfg in "swinir:--arch swinir_m" "span_f16:--arch span --compute f16" "mosr_f16:--arch mosr --compute f16" "gan:--gan --batch 8 --lq 64"; do
    tag=${cfg%%:*}; fl=${cfg#*:}
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- python3 $B $fl > $O/stats_$tag.log 2>&1
  done
fi
echo done
