"""bench.py's `roofline` object, computed from canned statistics (no GPU): the binding bound is one of the candidates
and its fraction is <= 1 by construction; the contractual HBM figure sits aside; the counter entries of
profiles/traffic.json are matched by plan (tiles, fusion depth) and scaled to the workgroups of a dispatch; a plan
without an entry falls back to the compulsory state traffic."""
import importlib.util
import json
import os

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class _Args:
    pass


def _stats(**kw):
    st = dict(bytes_per_px_iter=56.0, launches=1251, tile_iters=8, tiles=490, region_i=32, region_j=32, pdhg_variant=1, ncu=256,
              launch_chains=2)
    st.update(kw)
    return st


def test_headline_roofline_is_the_binding_bound_from_committed_counters():
    b = _bench()
    r = b.roofline_of(_Args(), _stats(), 128, 128, 10, 5000, 4.32, 8.14, False)
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["workloads"]["10x128x128 scalar"]
    assert r["bound"] == "valu_f64_issue" and 0.0 < r["frac"] <= 1.0
    # by hand: SQ_INSTS_VALU x (245 / 490) x 4 cycles / (256 CUs x 4 SIMDs) / 2.4 GHz / dispatch time
    floor_us = tj["valu_wave_instructions_per_launch"] * 0.5 * 4 / 1024 / 2400.0
    assert abs(r["frac"] - floor_us / 4.32) < 1e-12
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-12 and r["unit"].startswith("G VALU")
    assert set(r["candidates"]) == {"valu_f64_issue", "hbm"} and r["frac"] == max(r["candidates"].values())
    assert abs(r["frac_useful"] - r["frac"] / r["redundancy"]) < 1e-12 and abs(r["redundancy"] - 3.0625) < 1e-9
    assert r["contractual_hbm_frac"] > 1.0 and r["kernels_in_flight"] == 2        # the figure that may exceed the peak sits aside
    assert abs(r["traffic"] - 0.5 * tj["hbm_bytes_per_launch"]) < 1.0 and "traffic.json" in r["traffic_source"]


def test_plans_without_counters_fall_back_to_a_lower_bound():
    b = _bench()
    # a plan nobody profiled: 7 images at T = 9
    r = b.roofline_of(_Args(), _stats(tiles=343, tile_iters=9, launches=556, launch_chains=1), 128, 128, 7, 5000, 9.0, None, False)
    assert r["bound"] == "hbm" and r["candidates"] == {} and 0.0 < r["frac"] <= 1.0 and r["traffic"] is None
    assert "no counter entry" in r["bound_source"] and r["frac_useful"] is None
    # the float mode never borrows the Float64 counters
    r32 = b.roofline_of(_Args(), _stats(bytes_per_px_iter=28.0), 128, 128, 10, 5000, 4.0, None, True)
    assert r32["candidates"] == {} and r32["traffic"] is None


def test_rows_kernel_redundancy_counts_the_halo_rows_that_stop_early():
    b = _bench()
    st = _stats(bytes_per_px_iter=64.0, launches=501, tiles=3528, region_i=64, region_j=64, pdhg_variant=19, launch_chains=2)
    r = b.roofline_of(_Args(), st, 1024, 1024, 8, 2000, 91.0, None, False)
    assert r["kernel"] == "pdhg_rows_kernel" and abs(r["redundancy"] - 1.5176) < 1e-3
    assert r["bound"] == "valu_f64_issue" and 0.6 < r["frac"] < 0.7 and 0.5 < r["candidates"]["hbm"] < 0.56
