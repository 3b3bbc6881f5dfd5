#!/bin/bash
# quick SQ counter passes (two, counters only) on any python command; summary of the stepth
# kernels under gpurun_out/$1/summary.txt.  usage: run_sq_quick.sh OUT python script.py args..
OUT=gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/prof_sq -o sq -- "$@" > $OUT/sq.log 2>&1 &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d $OUT/prof_sq2 -o sq -- "$@" > $OUT/sq2.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -o kt -- "$@" > $OUT/kt.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o f -- "$@" > $OUT/f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o w -- "$@" > $OUT/w.log 2>&1 &&
python tools/sq_summary.py $OUT > /dev/null
grep -A18 "ring_tile_kernel<false\|ring_list_kernel<false\|sad_search" $OUT/summary.txt || true
