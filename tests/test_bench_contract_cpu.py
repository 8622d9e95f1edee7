"""What of bench.py's contract can be held without a GPU: the default flags the driver relies on (N = 1, K / W that finish
within minutes), the fields of the committed line, and that the PMC-derived numbers it quotes come from tables stamped
with the kernel sources it ran on (bench.source_hash)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_default_flags_are_the_contracts(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (1, 20, 3)
    assert a.gaussians == 1_200_000 and a.mode == "auto"        # auto = tile bands of ONE stream at N > 1 (a SLAM stream has one frame per step)
    assert a.sequence_frames >= 300 and not a.no_sequence and not a.no_config5 and not a.no_dropin and a.only is None
    assert not a.no_cpu_baseline and not a.no_schedule and not a.no_surface and not a.surface_map
    assert a.prewarm >= 3000 and a.repeats == 5
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "50", "--warmup", "5"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 50, 5)


def test_committed_pmc_numbers_and_bench_line_belong_to_the_same_kernel_sources():
    """The committed bench line quotes PMC traffic / VALU counts only from tables stamped with the hash of the kernel
    sources it ran on (bench.source_hash; a tree edited since then makes bench.py report null instead)."""
    import bench
    assert len(bench.source_hash()) == 16
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith("r") and f.endswith("_bench.json"))
    r = json.load(open(os.path.join(ROOT, "profiles", files[-1])))["roofline"]
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    v = json.load(open(os.path.join(ROOT, "profiles", "valu_latest.json")))
    assert t["_source_sha16"] == v["_source_sha16"]
    dom = "blend_bwd_entry" if "blend_bwd_entry" in t else "blend_bwd_mfma"      # the walk's kernel was renamed in round 5
    if r["traffic"] is not None:
        assert r["source_sha16"] == t["_source_sha16"]
        assert r["traffic"] == t[dom]["hbm_bytes_per_launch"]
    if r.get("valu") is not None:
        assert r["valu"]["wave_insts_per_launch"] == int(v[dom]["SQ_INSTS_VALU"])


def test_the_committed_bench_line_has_the_contracts_fields():
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith("r") and f.endswith("_bench.json"))
    d = json.load(open(os.path.join(ROOT, "profiles", files[-1])))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    # `value` counts hot-path UNITS (1 ICP track || 1 map iteration), and is named so since round 5; the SLAM frame rate by
    # the reference's definition is its own field, from the sequence leg (BASELINE configs[2])
    assert d["metric"] == "hot_path_units_per_sec" and d["unit"] == "units/s"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None
    seq = d["slam_sequence"]
    assert d["slam_frames_per_sec"] == seq["fps"] > 0 and seq["frames"] >= 300 and seq["start"] == "empty map"
    assert seq["image"] == [680, 1200] and seq["ate_rmse_m"] < 0.10 and seq["stable"] > 0 and seq["stats"]["global_opts"] > 0
    assert d["icp_track_ms_tum_480x640_noisy"]["image"] == [480, 640] and d["icp_track_ms_tum_480x640_noisy"]["ms_median"] > 0
    assert d["dropin_iteration_ms"] > d["dropin"]["one_call_step_ms"] > 0
    c5 = d["config5"]
    assert c5["gaussians"] == 5_000_000 and c5["mode"] == "sharded" and c5["iterations"] == 10
    assert set(c5["split_ms"]) == {"render_fwd_and_loss", "render_bwd", "collective", "adam"}
    assert d["n_gpus"] == 1 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # value is the whole-job rate of the timed block (since round 6: the MEDIAN of the five blocks; the first is beside it)
    assert abs(d["value"] - 1e3 * d["n_gpus"] / d["ms_per_step"]) / d["value"] < 0.01
    if files[-1] >= "r06":
        assert "first_block_ms_per_step" in d and d["repeats"]["blocks"] == 5
        assert abs(d["ms_per_step"] - d["repeats"]["median_ms_per_step"]) < 2e-3
        # BASELINE.json's three quantities where the driver's parser keeps VALUES: inside `config` / `roofline`
        cfg = d["config"]
        for k in ("slam_frames_per_sec", "raster_fwd_bwd_ms", "raster_fwd_bwd_ms_surface", "map_iteration_ms", "map_iteration_ms_surface",
                  "icp_track_ms", "icp_track_ms_tum_480x640_noisy", "hbm_frac_blend_fwd_plus_bwd", "config0_cpu_frames_per_sec",
                  "config5_ms_per_iteration"):
            assert isinstance(cfg[k], (int, float)) and cfg[k] > 0, k
        assert cfg["slam_frames_per_sec"] == d["slam_frames_per_sec"] and cfg["raster_fwd_bwd_ms"] == d["raster_fwd_bwd_ms"]
        assert 0 < r["frac_blend_fwd_plus_bwd"] < 1 and 0 < r["frac_blend_fwd_plus_bwd_surface"] < 1
        # configs[0]: 50 frames at GT poses through the CPU render path (in the cpu_baseline leg)
        c0 = c["config0"]
        assert c0["frames"] == 50 and c0["poses"] == "ground truth" and c0["cpu_frames_per_sec"] > 0 and c0["psnr_mean_db"] > 24
        assert c0["max_abs_colour_diff_hip_vs_cpu"] < 2e-3
        # configs[4]: the median of ten iterations, measured after iterations WITH the phase marks (the driver's r5 line held a cold one)
        assert len(c5["ms_each_iteration"]) == 10 and c5["untimed_iterations_before"] >= 3
        assert abs(c5["ms_per_iteration"] - sorted(c5["ms_each_iteration"])[5]) < 1e-3
        assert max(c5["host_enqueue_ms_per_iteration"]) < 5.0
