#!/bin/bash
# Builds and runs tools/micro/capture_fork_probe.hip, one process per case (a crash in one case must not
# hide the others).  Usage on the GPU box: bash tools/micro/run_capture_probe.sh > gpurun_out/capture_probe.log
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O2 -o capture_fork_probe capture_fork_probe.hip || exit 1
for c in ${CASES:-0 1 2 3 4 5 6 7 8 9 10 11 12}; do
  timeout -k 5 60 ./capture_fork_probe $c
  echo "case $c: exit status $?"
done
# bisection of case 12 (the literal libtdx sequence): parts left out by bit mask, see the source
for m in ${SKIPS:-}; do
  timeout -k 5 60 ./capture_fork_probe 12 $m
  echo "case 12 skip $m: exit status $?"
done
exit 0
