#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of the secondary bench workloads (run through gpurun from the repo root); tools/make_traffic_json.py turns them into profiles/traffic.json entries.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_prof
mkdir -p $O
B="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-roofline --no-loader"
for cfg in "swinir:--arch swinir_m" "span_f16:--arch span --compute f16" "mosr_f16:--arch mosr --compute f16"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$tag -- python3 $B $fl > $O/fetch_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$tag -- python3 $B $fl > $O/write_$tag.log 2>&1
done
echo done
