#!/bin/bash
# build tools/probes/kernel_bench and list the register use of its kernels (hipcc cross-compiles without a GPU)
mkdir -p /tmp/kb && cd /tmp/kb || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -save-temps=cwd -o /root/repo/tools/probes/kernel_bench /root/repo/tools/probes/kernel_bench.hip 2>&1 | grep -v "argument unused" | head -30
grep -E "^\s+\.(vgpr_count|private_segment_fixed_size|name|vgpr_spill_count):" /tmp/kb/kernel_bench-hip-amdgcn-amd-amdhsa-gfx950.s | grep -v "\.name: *[a-z_]*$" | paste - - - - | sed 's/  */ /g; s/private_segment_fixed_size/scratch/; s/vgpr_spill_count/spill/' | c++filt | cut -c1-200 | grep -v "debug_pow\|small_sa\|max_rel"
