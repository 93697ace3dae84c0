#!/bin/bash
# The three ways the fast kernels can feed primitive records to the tests, timed on today's code with one counter pass
# each: records staged chunk by chunk in LDS (PT_MATH_FAST_LDS), records in scene order through scalar loads
# (PT_MATH_FAST_SCALAR), kind-sorted records through scalar loads with camera batches (PT_MATH_FAST, hierarchy off).
# Headline frame (1080p @ 64) and the 10 000-primitive scene brute force (1080p @ 4).  Same images bit for bit (tested).
# Usage (repo root on the GPU box):  bash tools/lds_vs_scalar.sh gpurun_out/r03/lds_vs_scalar
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1
case "$OUT" in /*) ;; *) OUT="$R/$OUT";; esac
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
: > "$OUT/lds_vs_scalar.txt"
for cfg in "default 1920 1080 64 5" "synthetic 1920 1080 4 5"; do
  for mode in lds scalar "nobvh"; do
    tag=$(echo "$cfg $mode" | tr ' ' '_')
    timeout -k 10 400 python3 "$R/tools/profile_frame.py" $cfg $mode reps=3 >> "$OUT/lds_vs_scalar.txt" || { echo "FAILED timing $tag"; exit 1; }
    timeout -s KILL 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU \
        --output-format csv -d "$OUT/pmc_$tag" -- python3 "$R/tools/profile_frame.py" $cfg $mode reps=1 > "$OUT/pmc_$tag.log" 2>&1 || { echo "FAILED pmc $tag"; exit 1; }
    python3 - "$OUT/pmc_$tag" "$tag" >> "$OUT/lds_vs_scalar.txt" <<'PY'
import csv, glob, sys
tot = {}
n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_" not in row["Kernel_Name"]: continue
        tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        n = max(n, int(row["Dispatch_Id"]))
launches = 2.0   # profile_frame renders once to warm up and `reps` times after
print("    counters per launch (%s): " % sys.argv[2] + ", ".join("%s %.4g" % (k, v / launches) for k, v in sorted(tot.items())))
PY
    rm -rf "$OUT/pmc_$tag"
  done
done
cat "$OUT/lds_vs_scalar.txt"
