#!/bin/bash
# Run GPU steps in order on the GPU box, each under its own timeout, logging to gpurun_out/<name>.log.
# A step that fails with an ordinary error (test failure, assertion) lets the next step run; a step that is KILLED
# (timeout, signal) stops the whole sequence: no further GPU work is started after a hang.
# usage: tools/gpu_steps.sh "name|timeout_s|command" ...
mkdir -p gpurun_out
rc_all=0
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "== step $name (timeout ${tmo}s): $cmd"
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "== step $name rc=$rc"; tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -ge 128 ]; then echo "== step $name was killed: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
