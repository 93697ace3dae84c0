"""Static instruction counts of one kernel between the phase markers of a -DPT_PHASE_MARK assembly listing
(development aid; no GPU needed).
    python tools/isa_phases.py [kernel-name-substring]
Compiles pathtracer_amd/csrc/pt_kernels_fast.hip to assembly with the library's flags, cuts out the kernel and prints,
for each stretch of the listing between two markers (in layout order, which follows the source closely), how many
vector, scalar, LDS and memory instructions it holds, and the kernel's register and spill counts."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pathtracer_amd", "csrc", "pt_kernels_fast.hip")
want = sys.argv[1] if len(sys.argv) > 1 else "render_fast_batched_kernelILb1ELb1ELi976"
extra = sys.argv[2:]
out = os.path.join(tempfile.gettempdir(), "pt_fast_marked.s")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-Wno-unused-function", "-ffp-contract=fast",
       "-fno-slp-vectorize", "-mllvm", "-simplifycfg-hoist-common=false", "-mllvm", "-disable-licm-promotion",
       "--cuda-device-only", "-S", "-DPT_PHASE_MARK", SRC, "-o", out] + extra
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if want in l and l.endswith(":") or (want in l and ": ;" in l))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
kernel = lines[start:end]
def kind(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return None
stretch, name, rows = {}, "entry", []
def close():
    global stretch
    if stretch: rows.append((name, stretch))
    stretch = {}
for no, l in enumerate(kernel):
    t = l.strip()
    m = re.match(r";\s*PHASE (\d+)", t)
    if m:
        close(); name = "after PHASE %s (line %d)" % (m.group(1), no); continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") or re.match(r"^\.?L?BB\d+_\d+:", t): continue
    k = kind(t.split()[0])
    if k: stretch[k] = stretch.get(k, 0) + 1
    if "v_readlane" in t or "v_writelane" in t: stretch["lane"] = stretch.get("lane", 0) + 1
    if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)", t): stretch["trans"] = stretch.get("trans", 0) + 1
    if re.match(r"v_(cmp|cndmask|min|max|med3)", t): stretch["cmpsel"] = stretch.get("cmpsel", 0) + 1
    if re.match(r"v_mov", t): stretch["mov"] = stretch.get("mov", 0) + 1
close()
cols = ["valu", "trans", "cmpsel", "mov", "lane", "salu", "smem", "lds", "vmem", "wait", "branch"]
print("%-32s" % "stretch" + "".join("%8s" % c for c in cols))
tot = {}
for name, st in rows:
    print("%-32s" % name + "".join("%8d" % st.get(c, 0) for c in cols))
    for c in cols: tot[c] = tot.get(c, 0) + st.get(c, 0)
print("%-32s" % "total" + "".join("%8d" % tot.get(c, 0) for c in cols))
for l in lines[end:end + 400]:
    if any(k in l for k in (".sgpr_count", ".vgpr_count", "spill_count", ".name:")):
        print(l.strip())
    if ".name:" in l and want in l: pass
