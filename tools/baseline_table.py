"""Prints the result table of the five benchmark configurations (markdown) from profiles/r03/config_table.json and the counter passes beside it.
    python3 tools/baseline_table.py"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r03")
rows = json.load(open(os.path.join(P, "config_table.json")))
pmc = {"2": "pmc_config2.json", "4": "pmc_config4_glass5_depth31.json"}
ex = {k: json.load(open(os.path.join(P, f)))["derived"]["executed_frac"] * 100 for k, f in pmc.items()}
print("| config | frame | mode | device | cores / GPUs | kernel ms | Msamples/s | flops/sample | % FP32 peak algorithmic | % FP32 peak executed | RMSE vs oracle (of mean) | max-abs | pixels off by > 0.05 | mean radiance within | rows compared |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    if r["mode"] == "cpu oracle":
        print("| %s | %s | CPU oracle | host | %s | %.0f | %.2f | %.1f | — | — | is the reference image | — | — | — | %d |" % (
            r["config"], r["frame"], r["device"].split(", ")[1], r["seconds"] * 1e3, r["msamples_per_s"], r["flops_per_sample"], r["rows"]))
        continue
    e = "%.1f" % ex[r["config"]] if (r["mode"] == "fast" and r["config"] in ex) else "—"
    if r["config"] == "5":
        alg = "— (the hierarchy skips the tests the count charges)"
    else:
        alg = "%.1f" % r["pct_fp32_peak_algorithmic"]
    print("| %s | %s | %s | MI355X | 1 | %.3f | %.0f | %.1f | %s | %s | %.2e | %.3g | %.2e | %.2e | %d%s |" % (
        r["config"], r["frame"], r["mode"], r["kernel_ms"], r["msamples_per_s"], r["flops_per_sample"], alg, e, r["rmse_rel"], r["max_abs"],
        r["bad_pixel_frac"], r["mean_rel"], r["rows_compared"], " (only these rows rendered)" if r["rendered"] != "whole frame" else ""))
