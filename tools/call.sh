#!/bin/bash
# One entry for the recorded GPU-box calls of every round: `gpurun -- 'bash tools/call.sh r6j'` runs tools/calls/r6j.sh (each of them writes its
# output under gpurun_out/<name>/; what is kept for the judge is copied to profiles/<name>_*/).  `bash tools/call.sh` lists them.
D=$(dirname "$0")/calls
if [ -z "$1" ]; then ls "$D" | sed 's/\.sh$//' | tr '\n' ' '; echo; exit 0; fi
exec bash "$D/$1.sh" "${@:2}"
