#!/bin/bash
# usage: pmc_quick.sh <tag> <profile_frame args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmcq_$tag
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM --output-format csv -d /tmp/pmcq_$tag -- python3 "$R/tools/profile_frame.py" "$@" reps=1 > /tmp/pmcq_$tag.log 2>&1 || { echo "FAILED pmc $tag"; tail -5 /tmp/pmcq_$tag.log; exit 1; }
python3 - /tmp/pmcq_$tag "$tag" <<'PY'
import csv, glob, sys
tot = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_" not in row["Kernel_Name"]: continue
        tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
n = 2.0
v = {k: x / n for k, x in tot.items()}
print("%-28s VALU %.3f G  SALU %.3f G  LDS %.3f G  VMEM %.3f G  lane util %.3f  occupancy %.3f" % (sys.argv[2], v["SQ_INSTS_VALU"]/1e9, v["SQ_INSTS_SALU"]/1e9, v["SQ_INSTS_LDS"]/1e9, v["SQ_INSTS_VMEM"]/1e9,
      v["SQ_THREAD_CYCLES_VALU"]/(v["SQ_ACTIVE_INST_VALU"]*64), 4*v["SQ_WAVE_CYCLES"]/(v["SQ_BUSY_CYCLES"]/32*1024*8)))
PY
