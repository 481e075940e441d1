#!/bin/bash
# usage: gpuq.sh <logfile> <timeout> <command...>   - retries only while the pool has no free slot (exit code 3)
log=$1; shift; to=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "rc=$rc" >> $log; exit $rc; fi
  sleep 75
done
echo "rc=3 (gave up)" >> $log
