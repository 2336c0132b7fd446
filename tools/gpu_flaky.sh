#!/bin/bash
# repeats a pytest selection under several environments and prints pass/abort per run: tools/gpu_flaky.sh <n> <pytest args...> -- ENV=.. ENV=..
n=$1; shift
sel=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do sel+=("$1"); shift; done
shift
for kv in "none=1" "$@"; do
  for i in $(seq 1 $n); do
    env $kv timeout -k 10 300 python -m pytest "${sel[@]}" -x -q > gpurun_out/flaky_last.log 2>&1
    rc=$?
    echo "$kv run $i rc=$rc $(grep -o "line [0-9]* in test_[a-z_]*" gpurun_out/flaky_last.log | head -1) $(tail -1 gpurun_out/flaky_last.log | cut -c1-80)"
  done
done
