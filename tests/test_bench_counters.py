"""bench.py's roofline extras come from stored counter passes: the stored file must parse, name the kernel the
benchmark launches, and give figures in their possible ranges."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _stored():
    with open(os.path.join(ROOT, bench.STORED_COUNTERS)) as f:
        return json.load(f)


def test_stored_counters_belong_to_the_benchmark_kernel():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    c = bench.stored_counters("render_fast_batched_kernel", 3.25)
    if not os.path.exists(os.path.join(ROOT, bench.STORED_COUNTERS)) or _stored().get("source_hash") != pmc_summary.kernel_source_hash():
        # the kernels changed after the passes were collected: the block must drop out of the line, not go stale in it
        assert c is None
        return
    assert c is not None, "no counter passes stored under " + bench.STORED_COUNTERS
    assert bench.STORED_COUNTERS in c["source"]
    assert 0.0 < c["lane_util"] <= 1.0 and 0.0 < c["occupancy"] <= 1.0
    assert 0.0 < c["executed_frac"] < 1.0
    assert c["executed_tflops_active_lanes"] <= c["executed_tflops"]
    lo, hi = c["valu_issue_ceiling"]["unclassified_slow"], c["valu_issue_ceiling"]["unclassified_fast"]
    assert 0.0 < lo < hi <= 1.2
    # the frame is written once and next to nothing is read: 1920 x 1080 x 3 floats, within 5 %
    assert abs(c["traffic"] / (1920 * 1080 * 12) - 1.0) < 0.05
    assert all(math.isfinite(v) for v in c.values() if isinstance(v, float))


def test_counters_of_other_sources_are_not_reported(monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    monkeypatch.setattr(pmc_summary, "kernel_source_hash", lambda: "0" * 16)
    assert bench.stored_counters("render_fast_batched_kernel", 3.25) is None


def test_counters_of_another_kernel_are_not_reported():
    assert bench.stored_counters("render_parity_kernel", 3.25) is None


def test_flops_table_has_the_benchmark_frame():
    with open(os.path.join(ROOT, "profiles", "flops_per_sample.json")) as f:
        assert bench.FLOPS_KEY in json.load(f)
