#!/bin/bash
# on the GPU box: the headline step with several builds of the package, alternately (ABAB...), since boxes differ by more
# than the changes under test.  usage: tools/ab_step.sh <n> <pkgdir> [<pkgdir> ...]   (default: round-3 build, working tree)
cd "$(dirname "$0")/.."
n=${1:-20}; shift
pk=("$@"); [ ${#pk[@]} -eq 0 ] && pk=(tools/probes/r3pkg .)
for rep in 1 2; do
  for p in "${pk[@]}"; do python tools/ab_step.py "$p" "$n" 2>/dev/null; done
done
