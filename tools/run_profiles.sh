#!/bin/bash
# Profile passes behind profiles/r2_* (run on the GPU box: gpurun -- 'bash tools/run_profiles.sh').
# Counters go in their own runs, with --kernel-trace/--stats only, as MI355X_MICROARCH.md
# prescribes.  The profiled command is the headline of bench.py (one HIP stream, no extra legs).
set -x
OUT=gpurun_out/prof_r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -o kt -- $CMD > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o f -- $CMD > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o w -- $CMD > $OUT/w.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/prof_sq -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d $OUT/prof_sq2 -o sq -- $CMD > $OUT/sq2.log 2>&1
tail -2 $OUT/kt.log
