#!/bin/bash
# gpurun, retried only while it answers "no box or slot free right now" (exit code 3: nothing ran, nothing was charged).
# Any other outcome -- success, a failing command, a timeout, a refusal -- ends the script with that code.
#   tools/gpurun_wait.sh <log file> <timeout s> '<command>'
LOG=$1; T=$2; CMD=$3
for try in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$CMD" > $LOG 2>&1; rc=$?
  [ $rc -ne 3 ] && break
  sleep 100
done
echo "done rc=$rc tries=$try" >> $LOG
exit $rc
