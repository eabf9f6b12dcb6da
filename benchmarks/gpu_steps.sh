#!/bin/bash
# Runs GPU steps one after the other, each under its own time limit; an ordinary failure (exit 1: a failed assertion)
# lets the next step run, anything else (fault, abort, time limit) ends the script there.
#   bash benchmarks/gpu_steps.sh "<seconds> <logfile> <command ...>" ...
export TMPDIR=/tmp
mkdir -p gpurun_out
for step in "$@"; do
  set -- $step
  secs=$1; log=$2; shift 2
  echo "== [$secs s] $* > gpurun_out/$log"
  timeout -k 10 "$secs" "$@" > "gpurun_out/$log" 2>&1
  rc=$?
  echo "   exit $rc"; tail -3 "gpurun_out/$log"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after exit $rc"; exit $rc; fi
done
exit 0
