#!/bin/bash
# One counter pass over the vector memory path (address unit, L1, data return) of a frame rendered by
# tools/profile_frame.py.  Usage (repo root on the GPU box):
#   bash tools/pmc_memory_path.sh <outdir> synthetic 1920 1080 16 5
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
case "$OUT" in /*) ;; *) OUT="$R/$OUT";; esac
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mem" -- python3 "$R/tools/profile_frame.py" "$@" > "$OUT/mem.log" 2>&1 || { echo "pass failed"; tail -5 "$OUT/mem.log"; exit 1; }
f=$(find "$OUT/mem" -name '*counter_collection.csv' | head -n 1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "render_" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-40s per dispatch %.4g (%d dispatches)" % (k, sum(v) / len(v), len(v)))
PY
rm -rf "$OUT/mem"
