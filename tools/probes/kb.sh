#!/bin/bash
# build tools/probes/kernel_bench and list the register use of its kernels (hipcc cross-compiles without a GPU).
# The binary also carries the kernels with the general power routine of rounds 1-3 (namespace sdfs_old, compiled from a
# copy of csrc/ with -DSDFS_POWY=0) so that both are timed interleaved in one process.
here="$(cd "$(dirname "$0")" && pwd)"
mkdir -p /tmp/kb && cd /tmp/kb || exit 1
rm -rf /tmp/kb/kb_oldpow && cp -r "$here/../../sdfs_via_autodiff_amd/csrc" /tmp/kb/kb_oldpow
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-macro-redefined -DKB_OLD_POW ${KB_OLD_DEFS:--DKB_OLD_POWY=0 -DKB_OLD_NT=3} -I/tmp/kb -save-temps=cwd -o "$here/kernel_bench" "$here/kernel_bench.hip" 2>&1 | grep -v "argument unused" | head -30
grep -E "^\s+\.(vgpr_count|private_segment_fixed_size|name|vgpr_spill_count):" /tmp/kb/kernel_bench-hip-amdgcn-amd-amdhsa-gfx950.s | grep -v "\.name: *[a-z_]*$" | paste - - - - | sed 's/  */ /g; s/private_segment_fixed_size/scratch/; s/vgpr_spill_count/spill/' | c++filt | cut -c1-200 | grep -v "debug_pow\|small_sa\|max_rel\|copy_stream"
