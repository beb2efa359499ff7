

def test_bench_headline_is_compact_and_complete():
    """bench.py prints ONE line on stdout: the compact headline of the full result (the legs in full go to bench_details.json and stderr).
    It has to carry the contract's keys and stay well inside the driver's 8 KB tail, whatever the legs hold."""
    import importlib.util, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    big = {"x%d" % i: list(range(50)) for i in range(40)}                     # a leg grown fat
    out = {"metric": "m", "value": 1.0, "unit": "landmarks/s", "n_gpus": 1, "steps": 50, "warmup": 5, "clock_settle_steps": 400, "ms_per_step": 0.18,
           "ms_per_step_cold": 0.21, "value_cold": 0.9, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "w"}, "roofline": {"bound": "fp64_valu", "kernel": "k", "achieved": 24.0, "peak": 78.6, "unit": "TFLOP/s", "frac": 0.3, "traffic": 1,
                                                      "avg_launch_ms": 0.098, "hbm": {"achieved": 980.0, "peak": 8000.0, "frac": 0.12}, "note": "x" * 3000},
           "cpu_baseline": {"value": 2.9e5, "unit": "landmarks/s", "cores": 1, "kind": "port", "sample": "s", "all_cores": {"value": 4e6}, "ba": {"gn_iters_per_s": 14.0},
                            "parity": {"rel_err_p99.9": 5e-15}, "configs0": big, "match": big},
           "ba": dict(big, gn_iters_per_s=6700.0, ms_per_iter=0.149, landmarks_total=1000000, shard_proxy={"ms_per_iter": 0.033, "landmarks": 125000}),
           "rooflines": {"linear_ls": dict(big, frac=0.66)}, "kernels": big, "match": dict(big, frac_of_peak=0.66, packed_bits_fp4={"frac_of_peak": 0.49}),
           "replay": big, "frontend": {"end_to_end_loop_device_resident": {"frames_per_s": 4000.0, "frames_per_s_with_upload": 3900.0},
                                       "end_to_end_loop_device_resident_ba_per_keyframe":
                                       dict(big, frames_per_s=2500.0, frames_per_s_with_upload=2450.0,
                                            bundle_adjust_per_keyframe={"engine": "device", "ms_per_adjustment_median": {"adjust_ms": 0.4}}),
                                       "cpu_baseline": None, "cpu_baseline_note": "n" * 500},
           "asymptote": dict(big, linear_ls={"GBps": 5600.0}, iterative_ls={"ms": 0.73}, ba_gn_iteration={"ms_per_iter": 1.17}),
           "sparse_ba": big, "transport": "t" * 1000, "ba_strong": None}
    h = bench.headline(out, None)
    line = json.dumps(h)
    assert len(line) < 4096
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "value_cold"):
        assert key in h, key
    assert h["roofline"]["frac"] == 0.3 and h["roofline"]["hbm"]["frac"] == 0.12 and "note" not in h["roofline"]
    assert h["cpu_baseline"]["cores"] == 1 and h["cpu_baseline"]["ba_gn_iters_per_s_all_cores"] == 14.0 and h["cpu_baseline"]["all_cores"] == 4e6
    # the loop legs: frames/s WITH the frames arriving inside the timed loop, the resident figure beside it; no CPU baseline, said so
    assert h["loop"]["rendered_60_frames"] == {"plain_frames_per_s": 3900.0, "plain_frames_per_s_resident": 4000.0, "ba_per_keyframe_frames_per_s": 2450.0,
                                               "ba_per_keyframe_frames_per_s_resident": 2500.0, "engine": "device", "adjust_ms_median": 0.4,
                                               "rmse_plain": None, "rmse_ba": None}
    assert h["loop"]["cpu_baseline"] is None and h["loop"]["cpu_baseline_note"]
    assert h["asymptote_1e7"]["linear_ls_GBps"] == 5600.0 and h["asymptote_1e7"]["ba_ms_per_iter"] == 1.17
    assert len(h["transport"]) <= 160 and h["details"] == "stderr"
