#!/bin/bash
# On the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of tools/pad_kernel_times.py on ONE shape of the padded
# pair plan; outputs under gpurun_out/prof_<tag>/, summary by tools/kernel_summary.py.
# usage: tools/pad_profile.sh <tag> <shape, comma separated>
TAG=$1; SHAPE=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SDFS_PAD_TIMES_PADDED_ONLY=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/tools/pad_kernel_times.py $SHAPE > $OUT/kt.log 2>&1 || { echo "kernel-trace run failed"; tail -5 $OUT/kt.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc1 -- python3 $ROOT/tools/pad_kernel_times.py $SHAPE > $OUT/pmc1.log 2>&1 || { echo "pmc1 failed"; tail -5 $OUT/pmc1.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python3 $ROOT/tools/pad_kernel_times.py $SHAPE > $OUT/pmc2.log 2>&1 || { echo "pmc2 failed"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -- python3 $ROOT/tools/pad_kernel_times.py $SHAPE > $OUT/pmc3.log 2>&1 || { echo "pmc3 failed"; exit 1; }
rocprofv3 --pmc SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- python3 $ROOT/tools/pad_kernel_times.py $SHAPE > $OUT/pmc4.log 2>&1 || { echo "pmc4 failed"; tail -5 $OUT/pmc4.log; }
cd $ROOT && python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
