#!/bin/bash
# helper: rebuild the library here (cross-compile), then run a command on the MI355X box
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" >/dev/null
exec /usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-900} -- "$@"
