#!/bin/bash
# On the GPU box (via gpurun): rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of ONE Newton-Krylov solve at
# GCY 20^6 (tools/newton_solve.py); outputs under gpurun_out/prof_<tag>/, summary by tools/kernel_summary.py.
# usage: tools/newton_profile.sh <tag> [krylov_f32] [n]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/tools/newton_solve.py "$@" > $OUT/kt.log 2>&1 || { echo "kernel-trace run failed"; tail -5 $OUT/kt.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python3 $ROOT/tools/newton_solve.py "$@" > $OUT/pmc2.log 2>&1 || { echo "pmc2 failed"; tail -5 $OUT/pmc2.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -- python3 $ROOT/tools/newton_solve.py "$@" > $OUT/pmc3.log 2>&1 || { echo "pmc3 failed"; tail -5 $OUT/pmc3.log; exit 1; }
cd $ROOT && python3 tools/kernel_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
