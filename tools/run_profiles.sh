#!/bin/bash
# Profile passes behind profiles/r3_* (run on the GPU box: gpurun -- 'bash tools/run_profiles.sh').
# Counters go in their own runs, with --kernel-trace/--stats only, as MI355X_MICROARCH.md
# prescribes.  1. the headline of bench.py (one HIP stream, no extra legs): kernel trace,
# FETCH_SIZE, WRITE_SIZE, two SQ passes.  2. three more legs, SQ passes + kernel trace
# (tools/run_sq_quick.sh): the natural-image leg with the heavy-first order, the adaptive flow
# (ring_list_kernel), the windowed-SAD mode on config 3.
set -x
OUT=gpurun_out/prof_r3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -o kt -- $CMD > $OUT/kt.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o f -- $CMD > $OUT/f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o w -- $CMD > $OUT/w.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/prof_sq -o sq -- $CMD > $OUT/sq.log 2>&1 &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d $OUT/prof_sq2 -o sq -- $CMD > $OUT/sq2.log 2>&1 &&
bash tools/run_sq_quick.sh prof_r3/asset python tools/leg_time.py asset 128 hf > /dev/null &&
bash tools/run_sq_quick.sh prof_r3/adaptive python tools/leg_adaptive.py bench > /dev/null &&
bash tools/run_sq_quick.sh prof_r3/adaptive_asset python tools/leg_adaptive.py asset > /dev/null &&
bash tools/run_sq_quick.sh prof_r3/sad python tools/leg_sad.py > /dev/null
tail -2 $OUT/kt.log
