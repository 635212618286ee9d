#!/bin/bash
# wrapper for gpurun calls: always rebuild the library on the build host first (a stale in-tree .so travels silently otherwise)
#   tools/gpu_run.sh <timeout> <script-or-command...>
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
