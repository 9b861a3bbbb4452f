#!/bin/bash
# tools/gpu.sh "<what for>" <timeout s> '<command list>': one gpurun call, its command list appended to this
# round's disposable log (tools/runs/README.md) so that numbers under profiles/ can be traced to it.
set -u
WHAT="$1"; TMO="$2"; CMD="$3"
LOG="$(dirname "$0")/runs/r06_gpurun_commands.log.sh"
[ -f "$LOG" ] || { echo '#!/bin/bash'; echo '# DISPOSABLE LOG, not source: the command lists of every gpurun call of round 6, in order (tools/runs/README.md)'; } > "$LOG"
{ echo; echo "# ---------------------------------------------------------------- $(date -u +%FT%TZ)  $WHAT"; echo "$CMD"; } >> "$LOG"
exec /usr/local/graft/bin/gpurun --timeout "$TMO" -- "$CMD"
