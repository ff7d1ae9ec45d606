#!/bin/bash
# Diagnostics (GPU box): the shader clock and power the SMI reports while a workload runs.
#   bash scripts/clock_watch.sh <out file> <command...>
out=$1; shift
: > "$out"
"$@" > "$out.run" 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do
  { date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk"; } >> "$out"
  sleep 0.5
done
wait $pid
echo "exit $?" >> "$out"
grep -v amdgpu.ids "$out.run" | tail -3
