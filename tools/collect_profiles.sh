#!/bin/bash
# Collects, on the GPU box, everything a profiles/rNN/ directory holds: the benchmark line, rocprofv3 kernel statistics
# and the five counter passes for the headline frame, the glass-grid depth-31 frame, the 10 000-primitive frame and
# the headline frame with general record kinds only.  Every GPU step has its own time limit and the script stops at
# the first failure.  Usage (repo root on the GPU box):  bash tools/collect_profiles.sh gpurun_out/r03
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1
case "$OUT" in /*) ;; *) OUT="$R/$OUT";; esac
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
step() { echo "== $*"; "$@" || { echo "FAILED: $*"; exit 1; }; }

echo "== bench"
timeout -k 10 300 python3 "$R/bench.py" > "$OUT/bench.json" || { echo "FAILED: bench"; exit 1; }
timeout -k 10 300 python3 "$R/bench.py" --general-kinds --no-cpu-baseline > "$OUT/bench_general_kinds.json" || { echo "FAILED: bench general"; exit 1; }

stats() {   # name, program words...
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$name" -- "$@" > "$OUT/stats_$name.log" 2>&1 || { echo "FAILED: stats $name"; exit 1; }
  f=$(find "$OUT/stats_$name" -name '*kernel_stats.csv' | head -n 1)
  cp "$f" "$OUT/rocprof_kernel_stats_$name.csv" && rm -rf "$OUT/stats_$name"
}
stats config2 python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline
stats config4_glass5_depth31 python3 "$R/tools/profile_frame.py" glass5 1920 1080 64 31 reps=8
stats config5_10k_prims python3 "$R/tools/profile_frame.py" synthetic 1920 1080 16 5 reps=8
stats config2_general_kinds python3 "$R/tools/profile_frame.py" default 1920 1080 64 5 general reps=8
stats config2_parity_order python3 "$R/tools/profile_frame.py" default 1920 1080 64 5 parity reps=8

cd "$R"
step bash tools/pmc_passes.sh "$OUT/pmc_config2"
export PROG=tools/profile_frame.py
step bash tools/pmc_passes.sh "$OUT/pmc_config4_glass5_depth31" glass5 1920 1080 64 31
step bash tools/pmc_passes.sh "$OUT/pmc_config5_10k_prims" synthetic 1920 1080 16 5
step bash tools/pmc_passes.sh "$OUT/pmc_config2_general_kinds" default 1920 1080 64 5 general
step bash tools/pmc_passes.sh "$OUT/pmc_config2_parity_order" default 1920 1080 64 5 parity
unset PROG

ms_of() { python3 - "$1" <<'EOF'
import csv, sys
best = None
for row in csv.DictReader(open(sys.argv[1])):
    if "render_" in row["Name"] and (best is None or float(row["TotalDurationNs"]) > float(best["TotalDurationNs"])):
        best = row
# (the first launch of a process runs cold, up to three times as long: the slowest launch is left out of the average)
calls = int(best["Calls"])
print((float(best["TotalDurationNs"]) - float(best["MaxNs"])) / max(1, calls - 1) / 1e6 if calls > 1 else float(best["AverageNs"]) / 1e6)
EOF
}
for c in config2 config4_glass5_depth31 config5_10k_prims config2_general_kinds config2_parity_order; do
  ms=$(ms_of "$OUT/rocprof_kernel_stats_$c.csv")
  python3 tools/pmc_summary.py "$OUT/pmc_$c" render_ "$ms" > "$OUT/pmc_$c.json" || exit 1
  for p in occupancy mix1 mix2 fetch write; do rm -rf "$OUT/pmc_$c/$p"; done
done
echo "profiles collected"
