"""Summarises rocprofv3 --pmc CSV output (one directory per pass, as written by
tools/pmc_passes.sh) for one kernel: sums each counter over the dispatches of
that kernel and prints per-dispatch averages as JSON, with the derived figures
of derive() when a kernel time is given.

    python tools/pmc_summary.py gpurun_out/pmc render_fast_batched [kernel_ms] > profiles/r02/pmc_x.json
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["pt_kernels_fast.hip", "pt_batched.h", "pt_wavepool.h", "pt_internal.h"]


def kernel_source_hash():
    """What the fast render kernels are compiled from: the sources and the compiler flags of the Makefile.  Counter
    passes carry this hash; a reader that finds another one knows the passes belong to older kernels."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "pathtracer_amd", "csrc")
    for name in KERNEL_SOURCES:
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    with open(os.path.join(csrc, "Makefile")) as f:
        for line in f:
            if re.match(r"^(COMMON|NOCOPY|FASTFLAGS|ARCH)\s*\??=", line):
                h.update(line.strip().encode())
    return h.hexdigest()[:16]

N_SIMD = 1024                 # 256 compute units x 4
WAVES_PER_SIMD = 8
N_SHADER_ENGINES = 32         # SQ_BUSY_CYCLES is summed over them
FP32_VALU_PEAK_TFLOPS = 157.3
# wave-instructions per ns per SIMD at 8 waves per SIMD, measured by tools/valu_microbench.hip on this part
RATE_FAST, RATE_SLOW, RATE_TRANS = 1.0, 0.57, 0.29


def derive(pmc, kernel_ms=None):
    """Figures that follow from the counter passes of one kernel launch (per-dispatch averages):

    lane_util        active lanes per issued vector instruction / 64
    occupancy        resident wave-cycles / (kernel cycles x 1024 SIMDs x 8 waves); the SQ counts wave-cycles in units of 4
    executed flop    (2 FMA + MUL + ADD + TRANS) wave-instructions x 64 lanes: issue slots, and x lane_util: on active lanes
    valu_issue_rate  vector wave-instructions per ns per SIMD (needs kernel_ms), beside the rate the microbenchmark gives
                     for this mix when the instructions no counter classifies (compares, selects, min/max, moves, lane
                     ops) all issue at the slow class's rate, and when they all issue at the fast class's rate
    """
    occ, m1, m2 = (pmc[k]["per_dispatch"] for k in ("occupancy", "mix1", "mix2"))
    lane_util = occ["SQ_THREAD_CYCLES_VALU"] / (occ["SQ_ACTIVE_INST_VALU"] * 64.0)
    cycles = occ["SQ_BUSY_CYCLES"] / N_SHADER_ENGINES
    fma, mul, add, trans = m1["SQ_INSTS_VALU_FMA_F32"], m2["SQ_INSTS_VALU_MUL_F32"], m2["SQ_INSTS_VALU_ADD_F32"], m1["SQ_INSTS_VALU_TRANS_F32"]
    i32, cvt, valu = m2["SQ_INSTS_VALU_INT32"], m2["SQ_INSTS_VALU_CVT"], occ["SQ_INSTS_VALU"]
    other = max(0.0, valu - fma - mul - add - trans - i32 - cvt)
    slots = (2.0 * fma + mul + add + trans) * 64.0
    d = {"lane_util": lane_util,
         "occupancy": 4.0 * occ["SQ_WAVE_CYCLES"] / (cycles * N_SIMD * WAVES_PER_SIMD),
         "valu_wave_instructions": valu,
         "valu_mix": {"fma_f32": fma / valu, "mul_f32": mul / valu, "add_f32": add / valu, "trans_f32": trans / valu,
                      "int32": i32 / valu, "cvt": cvt / valu, "unclassified": other / valu},
         "executed_gflop_issue_slots": slots / 1e9, "executed_gflop_active_lanes": slots * lane_util / 1e9,
         "salu_per_valu": m1["SQ_INSTS_SALU"] / valu, "smem_instructions": m1["SQ_INSTS_SMEM"],
         "lds_instructions": m1["SQ_INSTS_LDS"], "lds_bank_conflict_frac": m1["SQ_LDS_BANK_CONFLICT"] / max(1.0, m1["SQ_LDS_IDX_ACTIVE"]),
         "hbm_bytes": (pmc["write"]["per_dispatch"]["WRITE_SIZE"] + 2.0 * pmc["fetch"]["per_dispatch"]["FETCH_SIZE"]) * 1024.0}
    if kernel_ms:
        ns = kernel_ms * 1e6
        fast_known = fma + mul + add + 0.75 * i32        # the hash is 3 slow multiplies in about 13 integer instructions
        slow_known = 0.25 * i32 + cvt
        d.update({"kernel_ms": kernel_ms,
                  "executed_tflops": slots / ns / 1e3, "executed_frac": slots / ns / 1e3 / FP32_VALU_PEAK_TFLOPS,
                  "executed_tflops_active_lanes": slots * lane_util / ns / 1e3,
                  "valu_issue_rate": valu / N_SIMD / ns,
                  "valu_issue_ceiling": {
                      "unclassified_slow": valu / (fast_known / RATE_FAST + (slow_known + other) / RATE_SLOW + trans / RATE_TRANS),
                      "unclassified_fast": valu / ((fast_known + other) / RATE_FAST + slow_known / RATE_SLOW + trans / RATE_TRANS)}})
    return d


def summarise(root, kernel):
    out = {}
    names = set()
    for path in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
        pass_name = os.path.relpath(path, root).split(os.sep)[0]
        sums, disp = {}, set()
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel not in row["Kernel_Name"]:
                    continue
                disp.add(row["Dispatch_Id"])
                names.add(row["Kernel_Name"])
                sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        n = max(1, len(disp))
        out[pass_name] = {"dispatches": len(disp), "per_dispatch": {k: v / n for k, v in sorted(sums.items())}}
    out["profiled_kernel"] = "; ".join(sorted(names))
    out["source_hash"] = kernel_source_hash()
    return out


def main():
    root, kernel = sys.argv[1], sys.argv[2]
    out = summarise(root, kernel)
    if all(k in out and out[k]["dispatches"] for k in ("occupancy", "mix1", "mix2", "fetch", "write")):
        out["derived"] = derive(out, float(sys.argv[3]) if len(sys.argv) > 3 else None)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
