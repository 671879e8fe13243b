#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of the bench command + separate PMC passes on one wave of pairs.
#   bash tools/profile_round.sh <tag>        -> gpurun_out/prof_<tag>/...
# Environment (defaults = BASELINE config 3 shape, one wave of 512 pairs):
#   PR_W PR_H PR_LEVELS PR_ITERS PR_BATCH   frame size, pyramid levels, iterations, pairs in the wave
#   PR_PASSES = all | traffic                all counter groups, or only FETCH_SIZE + WRITE_SIZE
#   PR_TRACE  = 1 | 0                        also trace the bench command (bench.py --config $PR_CONFIG)
# Counters go in passes of their own (no trace domains next to --pmc), the program directly after `--`.
set -u
TAG=${1:-r03}
W=${PR_W:-1920}; H=${PR_H:-1080}; LEVELS=${PR_LEVELS:-5}; ITERS=${PR_ITERS:-3}; BATCH=${PR_BATCH:-512}
PASSES=${PR_PASSES:-all}; TRACE=${PR_TRACE:-1}; CONFIG=${PR_CONFIG:-3}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
K="python3 $ROOT/tools/kbench.py --w $W --h $H --levels $LEVELS --iterations $ITERS --batch $BATCH --reps 1"
echo "$K" > $OUT/kbench_command.txt
if [ "$TRACE" = 1 ]; then
  (cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -o runc -- python3 $ROOT/bench.py --config $CONFIG --steps 3 --warmup 1 --cpu-sample 0 --no-two-stream --no-family-check > $ROOT/$OUT/trace_bench.json 2> $ROOT/$OUT/trace.log)
  echo "trace done rc=$?"
fi
pass() {  # name, counters...
    local name=$1; shift
    (cd /tmp && timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $ROOT/$OUT/pmc_$name -o runc -- $K > $ROOT/$OUT/pmc_$name.log 2>&1)
    echo "pmc $name rc=$?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
if [ "$PASSES" = all ]; then
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR
pass sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
pass ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum
fi
ls $OUT
