#!/bin/bash
# Collects the hardware-counter passes for the render kernel on the GPU box.
# Each pass is its own rocprofv3 run (<= 8 SQ counters, TCC passes separate),
# kernel-trace/stats are never combined with --pmc.  Usage (from the repo root
# on the GPU box):
#   bash tools/pmc_passes.sh <outdir> [bench args...]                    profiles bench.py
#   PROG=tools/profile_frame.py bash tools/pmc_passes.sh <outdir> glass5 1920 1080 64 31
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
case "$OUT" in /*) ;; *) OUT="$R/$OUT";; esac
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
if [ -n "${PROG:-}" ]; then
  CMD=(python3 "$R/$PROG" "$@")
else
  CMD=(python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@")
fi
run_pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- "${CMD[@]}" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; exit 1; }
}
run_pass occupancy SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass mix1 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32
run_pass mix2 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_INST_CYCLES_SALU SQ_INSTS_VMEM
run_pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
run_pass write WRITE_SIZE GRBM_COUNT
echo "pmc passes done"
