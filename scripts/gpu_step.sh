#!/bin/bash
# Run GPU steps one after the other; stop at the first one that is killed by its time limit (rc 124 / 137) or by
# a signal -- an assertion failure (rc 1) does not stop the sequence.  Usage: gpu_step.sh "<cmd>" "<cmd>" ...
# Each step: timeout -k 10 ${STEP_LIMIT:-420} bash -c "<cmd>"
for cmd in "$@"; do
  echo "== $cmd"
  timeout -k 10 ${STEP_LIMIT:-420} bash -c "$cmd"
  rc=$?
  echo "== rc=$rc"
  if [ $rc -ge 124 ]; then
    echo "== step killed (rc $rc): stopping"
    exit $rc
  fi
done
exit 0
