#!/bin/bash
# on the GPU box: one command with several builds of the library side by side (SDFS_LIB_NAME), alternately, since boxes
# differ by more than most changes under test.   usage: tools/ab_libs.sh "<lib suffixes, e.g. '' _v1 _v2>" <reps> <command ...>
cd "$(dirname "$0")/.."
libs=($1); reps=$2; shift 2
[ ${#libs[@]} -eq 0 ] && libs=("")
for rep in $(seq 1 $reps); do
  for v in "${libs[@]}"; do
    [ "$v" = "-" ] && v=""
    echo "== libsdfs_hip$v.so rep $rep"
    SDFS_LIB_NAME=libsdfs_hip$v.so "$@" 2>/dev/null
  done
done
