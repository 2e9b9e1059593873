#!/bin/bash
# tools/gstep.sh <seconds> <command...>: one GPU step of a gpurun call under its own timeout.  A test failure (exit < 124) does not
# stop an `&&` chain; a timeout or a kill does, so that no further GPU step starts behind a hung one.
t=$1; shift
timeout -k 10 "$t" "$@"; rc=$?
if [ $rc -ge 124 ]; then echo "gstep: '$*' ended with $rc (timeout / signal): stopping the chain" >&2; exit $rc; fi
exit 0
