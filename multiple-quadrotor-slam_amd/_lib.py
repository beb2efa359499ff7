"""
ctypes loader of libmqslam_hip.so -- the drop-in counterpart of the reference's
`convert_c_to_ext_lib.py` + `triangulation_c/__init__.py:1-11` pattern: try the prebuilt
library, otherwise build it (hipcc, gfx950), otherwise record `loaded = False`.

Unlike the reference there is NO slower fallback: every compute entry point of this package
raises RuntimeError when the HIP library (or a GPU) is missing.
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("MQS_LIB_PATH", os.path.join(_PKG, "libmqslam_hip.so"))   # override: kernel A/B builds

loaded = False
load_error = None
_lib = None

c_f64p = ctypes.POINTER(ctypes.c_double)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u16p = ctypes.POINTER(ctypes.c_uint16)
c_i64 = ctypes.c_int64
c_vp = ctypes.c_void_p

MQS_OK = 0
MQS_PEER_MAX_WORLD = 8          # include/mqslam.h
MQS_PEER_HANDLE_BYTES = 128

# name -> (restype, argtypes); must list every symbol include/mqslam.h declares
# (tests/test_abi.py parses the header and checks this table and the .so against it).
SIGNATURES = {
    "mqs_last_error": (ctypes.c_char_p, []),
    "mqs_version": (ctypes.c_char_p, []),
    "mqs_device_count": (ctypes.c_int, []),
    "mqs_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_vp)]),
    "mqs_destroy": (None, [c_vp]),
    "mqs_synchronize": (ctypes.c_int, [c_vp]),
    "mqs_triangulate_linear_ls": (ctypes.c_int, [c_vp, c_f64p, c_f64p, ctypes.c_int, c_i64, c_f64p]),
    "mqs_triangulate_iterative_ls": (ctypes.c_int, [c_vp, c_f64p, c_f64p, ctypes.c_int, c_i64, ctypes.c_double,
                                                    ctypes.c_int, c_f64p, c_i32p]),
    "mqs_triangulate_linear_eigen": (ctypes.c_int, [c_vp, c_f64p, c_f64p, ctypes.c_int, c_i64, ctypes.c_double,
                                                    c_f64p, c_u8p]),
    "mqs_linear_LS_triangulation": (ctypes.c_int, [c_vp, c_f64p, c_f64p, c_f64p, c_f64p, c_i64, c_f64p]),
    "mqs_iterative_LS_triangulation": (ctypes.c_int, [c_vp, c_f64p, c_f64p, c_f64p, c_f64p, c_i64, ctypes.c_double,
                                                      c_f64p, c_i32p]),
    "mqs_triangulate_linear_ls_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_i64, c_vp, c_vp]),
    "mqs_triangulate_iterative_ls_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_i64, ctypes.c_double,
                                                        ctypes.c_int, c_vp, c_vp, c_vp]),
    "mqs_triangulate_f32_dev": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp, ctypes.c_int, c_i64, ctypes.c_double, ctypes.c_int,
                                               ctypes.c_double, c_vp, c_vp, c_vp, c_vp]),
    "mqs_triangulate_f32": (ctypes.c_int, [c_vp, ctypes.c_int, c_f32p, c_f64p, ctypes.c_int, c_i64, ctypes.c_double, ctypes.c_int,
                                           ctypes.c_double, c_f64p, c_i32p, c_u8p]),
    "mqs_triangulation_2view_f32": (ctypes.c_int, [c_vp, ctypes.c_int, c_f32p, c_f64p, c_f32p, c_f64p, c_i64, ctypes.c_double,
                                                   ctypes.c_double, c_f64p, c_i32p, c_u8p]),
    "mqs_triangulate_ls_and_iterative_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_i64, ctypes.c_double, ctypes.c_int,
                                                            c_vp, c_vp, c_vp, c_vp]),
    "mqs_triangulate_linear_eigen_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_i64, ctypes.c_double, c_vp,
                                                        c_vp, c_vp]),
    "mqs_triangulate_pixels_dev": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, ctypes.c_double,
                                                  ctypes.c_int, ctypes.c_double, c_vp, c_vp, c_vp, c_vp]),
    "mqs_match_knn2_f32": (ctypes.c_int, [c_vp, c_f32p, c_i64, c_f32p, c_i64, ctypes.c_int, c_i32p, c_f32p]),
    "mqs_match_knn2_f32_dev": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_vp]),
    "mqs_match_knn2_f16": (ctypes.c_int, [c_vp, c_u16p, c_i64, c_u16p, c_i64, ctypes.c_int, c_i32p, c_f32p]),
    "mqs_match_knn2_f16_dev": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mqs_match_knn2_f16_workspace_bytes": (c_i64, [c_i64, c_i64]),
    "mqs_ba_linearize_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                            ctypes.c_double, c_vp, c_vp, c_i64, c_vp]),
    "mqs_ba_workspace_bytes": (c_i64, [ctypes.c_int, c_i64]),
    "mqs_ba_solve_dev": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, ctypes.c_double, c_vp, c_vp,
                                        c_vp, c_vp]),
    "mqs_ba_backsub_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                          ctypes.c_double, c_vp, c_vp, c_vp]),
    "mqs_ba_solve_backsub_dev": (ctypes.c_int, [c_vp, ctypes.c_int] + [c_vp] * 8 + [c_i64, ctypes.c_double] + [c_vp] * 8),
    "mqs_ba_cost_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp,
                                       c_vp, c_i64, c_vp]),
    "mqs_ba_linearize": (ctypes.c_int, [c_vp, c_f64p, c_f64p, c_f64p, ctypes.c_int, c_f64p, c_f64p, c_u8p, c_f64p,
                                        c_f64p, c_i64, ctypes.c_double, c_f64p]),
    "mqs_ba_backsub": (ctypes.c_int, [c_vp, c_f64p, c_f64p, c_f64p, ctypes.c_int, c_f64p, c_f64p, c_u8p, c_f64p,
                                      c_f64p, c_i64, ctypes.c_double, c_f64p, c_f64p]),
    "mqs_ba_time_dev": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp, c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                       ctypes.c_double, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_int, c_vp,
                                       ctypes.POINTER(ctypes.c_float)]),
    "mqs_comm_unique_id": (ctypes.c_int, [c_u8p]),
    "mqs_comm_init_rank": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int]),
    "mqs_comm_world_size": (ctypes.c_int, [c_vp]),
    "mqs_comm_destroy": (ctypes.c_int, [c_vp]),
    "mqs_comm_all_reduce_sum_f64_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "mqs_comm_peer_export": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_u8p]),
    "mqs_comm_peer_open": (ctypes.c_int, [c_vp, c_u8p]),
    "mqs_comm_peer_close": (ctypes.c_int, [c_vp]),
    "mqs_comm_peer_state": (ctypes.c_int, [c_vp]),
    "mqs_comm_peer_timed_out": (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(ctypes.c_int)]),
    "mqs_ba_problem_create": (ctypes.c_int, [c_vp, ctypes.c_int, c_i64] + [c_vp] * 17 + [c_i64, ctypes.POINTER(c_vp)]),
    "mqs_ba_problem_destroy": (None, [c_vp]),
    "mqs_ba_problem_current": (ctypes.c_int, [c_vp]),
    "mqs_ba_problem_set_current": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "mqs_ba_gn_begin_dev": (ctypes.c_int, [c_vp, ctypes.c_double, c_vp]),
    "mqs_ba_gn_finish_dev": (ctypes.c_int, [c_vp, ctypes.c_double, ctypes.c_int, c_vp]),
    "mqs_ba_gn_iteration_dev": (ctypes.c_int, [c_vp, ctypes.c_double, c_vp]),
    "mqs_ba_gn_iterations_dev": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_double, c_vp]),
    "mqs_ba_problem_status": (ctypes.c_int, [c_vp, c_vp]),
    "mqs_debug_ba_withhold_flag": (ctypes.c_int, [ctypes.c_int]),
    "mqs_sba_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "mqs_sba_linearize_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp,
                                             c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_double,
                                             c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mqs_sba_linearize_grouped_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp,
                                                     c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int,
                                                     ctypes.c_double, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mqs_sba_group_pairs_workspace_bytes": (c_i64, [c_i64]),
    "mqs_sba_group_pairs_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "mqs_sba_sort_observations_workspace_bytes": (c_i64, [c_i64]),
    "mqs_sba_sort_observations_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mqs_sba_solve_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, ctypes.c_double, c_vp, c_vp, c_vp, c_vp]),
    "mqs_sba_solve_banded_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, ctypes.c_double, c_vp, c_vp, c_vp, c_vp]),
    "mqs_sba_solve_plan_dump": (ctypes.c_int64, [c_i64, c_i64, ctypes.c_int, c_vp, c_i64]),
    "mqs_sba_solve_plan_cache_size": (ctypes.c_int, []),
    "mqs_sba_backsub_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp,
                                           c_vp, ctypes.c_double, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mqs_sba_cost_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp,
                                        c_vp, c_vp, c_i64, c_vp]),
    "mqs_sba_between_dev": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "mqs_sba_worst_residual_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mqs_sba_lm_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "mqs_sba_optimize_lm_dev": (ctypes.c_int, [c_vp, c_vp, c_f64p, ctypes.c_int32, c_vp, c_vp]),
    "mqs_undistort_points": (ctypes.c_int, [c_vp, c_f64p, c_f64p, c_i64, c_f64p]),
    "mqs_undistort_points_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "mqs_project_points": (ctypes.c_int, [c_vp, c_f64p, c_f64p, c_f64p, c_f64p, c_i64, c_f64p, c_f64p, c_f64p]),
    "mqs_project_points_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mqs_project_workspace_bytes": (c_i64, []),
    "mqs_keyframe_step": (ctypes.c_int, [c_vp, c_f64p, c_f64p, c_i64, c_f64p, c_f64p, c_i64, c_f64p, c_f64p, c_f64p,
                                         ctypes.c_double, ctypes.c_int, ctypes.c_double, c_f64p, c_f64p, c_i32p, c_f64p]),
    "mqs_solve_pnp": (ctypes.c_int, [c_vp, c_f64p, c_f64p, c_i64, c_f64p, c_f64p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_double, c_f64p]),
    "mqs_pnp_refine_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, ctypes.c_int, c_vp, c_vp, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_double, c_vp, c_vp, c_vp]),
    "mqs_solve_pnp_ransac": (ctypes.c_int, [c_vp, c_f64p, c_f64p, c_i64, c_f64p, c_i32p, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_f64p,
                                            c_i32p, c_u8p, c_f64p]),
    "mqs_pnp_ransac_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_double, c_vp, c_vp, c_vp, c_vp, c_vp,
                                          c_i64, c_vp]),
    "mqs_pnp_workspace_bytes": (c_i64, [c_i64, ctypes.c_int]),
    "mqs_good_features_to_track": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                  ctypes.c_double, c_u8p, c_f32p, ctypes.c_int, c_i32p]),
    "mqs_good_features_to_track_dev": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                      ctypes.c_double, c_vp, c_vp, ctypes.c_int, c_vp, c_vp, c_i64, c_vp]),
    "mqs_gftt_workspace_bytes": (c_i64, [ctypes.c_int, ctypes.c_int]),
    "mqs_calc_optical_flow_pyr_lk": (ctypes.c_int, [c_vp, c_u8p, c_u8p, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                    ctypes.c_double, c_f32p, c_u8p, c_f32p]),
    "mqs_calc_optical_flow_pyr_lk_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int,
                                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_double, ctypes.c_double, c_vp, c_vp, c_vp, c_vp, c_i64,
                                                        c_vp]),
    "mqs_lk_workspace_bytes": (c_i64, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mqs_fast_detect": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_i32p,
                                       ctypes.c_int, c_i32p]),
    "mqs_fast_detect_dev": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp,
                                           ctypes.c_int, c_vp, c_vp, c_i64, c_vp]),
    "mqs_fast_workspace_bytes": (c_i64, [ctypes.c_int, ctypes.c_int]),
    "mqs_match_knn2_bits": (ctypes.c_int, [c_vp, c_u8p, c_i64, c_u8p, c_i64, ctypes.c_int, c_i32p, c_f32p]),
    "mqs_match_knn2_bits_dev": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mqs_match_knn2_bits_workspace_bytes": (c_i64, [c_i64, c_i64, ctypes.c_int]),
    "mqs_match_ratio_unique_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, ctypes.c_float, ctypes.c_double, c_vp, c_vp, c_vp,
                                                  c_vp, c_i64, c_vp]),
    "mqs_match_ratio_unique_workspace_bytes": (c_i64, [c_i64]),
    "mqs_match_radius_ratio_unique": (ctypes.c_int, [c_vp, c_f32p, c_i64, c_f32p, c_i64, ctypes.c_int, ctypes.c_float,
                                                     ctypes.c_double, c_f32p, c_i32p, c_f32p]),
    "mqs_slam_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f64p, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(c_vp)]),
    "mqs_slam_destroy": (None, [c_vp]),
    "mqs_slam_set_thresholds": (ctypes.c_int, [c_vp] + [ctypes.c_double] * 5 + [ctypes.c_int]),
    "mqs_slam_set_second_pass_screen": (ctypes.c_int, [c_vp, ctypes.c_double]),
    "mqs_slam_reassociate": (ctypes.c_int, [c_vp, ctypes.c_float, ctypes.c_double, c_i32p]),
    "mqs_slam_bundle_adjust": (ctypes.c_int, [c_vp, c_vp, c_f64p, c_f64p, ctypes.c_int32]),
    "mqs_slam_bundle_adjust_window": (ctypes.c_int, [c_vp, c_vp, c_vp, c_f64p, c_f64p, ctypes.c_int32]),
    "mqs_slam_ba_resident_groups": (ctypes.c_int, [c_vp, c_i32p]),
    "mqs_debug_slam_ba_fail_next": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "mqs_slam_read_ba_flags": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int, c_i32p]),
    "mqs_slam_read_ba_edges": (ctypes.c_int, [c_vp, c_i32p, c_i32p, c_f64p, ctypes.c_int, c_i32p]),
    "mqs_debug_slam_ba_stamps": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i64), ctypes.c_int, c_i32p]),
    "mqs_debug_factor32": (ctypes.c_int, [c_f64p, c_f64p, ctypes.c_int, c_i32p]),
    "mqs_slam_ingest_enable": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "mqs_slam_upload": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, ctypes.c_int]),
    "mqs_slam_wait_upload": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "mqs_slam_prepare_next": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_vp, ctypes.c_int]),
    "mqs_slam_set_next": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, ctypes.c_int]),
    "mqs_slam_pipeline": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "mqs_slam_log_enable": (ctypes.c_int, [c_vp, c_i64]),
    "mqs_slam_read_log": (ctypes.c_int, [c_vp, c_i32p, c_i32p, c_f64p, c_i64, ctypes.POINTER(c_i64)]),
    "mqs_slam_write_back": (ctypes.c_int, [c_vp, c_f64p, ctypes.c_int, c_f64p, c_f64p]),
    "mqs_slam_start": (ctypes.c_int, [c_vp, c_vp, c_f32p, c_f32p, ctypes.c_int, c_f64p]),
    "mqs_slam_track": (ctypes.c_int, [c_vp, c_vp, c_vp, c_f64p]),
    "mqs_slam_flush": (ctypes.c_int, [c_vp, c_f64p]),
    "mqs_slam_read_tracks": (ctypes.c_int, [c_vp, c_f32p, c_f32p, c_i32p, c_i32p, ctypes.c_int, c_i32p]),
    "mqs_slam_read_map": (ctypes.c_int, [c_vp, c_f32p, ctypes.c_int, c_i32p]),
    "mqs_time_triangulate_dev": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp, ctypes.c_int, c_i64, ctypes.c_double,
                                                ctypes.c_int, c_vp, c_vp, c_vp, ctypes.c_int, c_vp,
                                                ctypes.POINTER(ctypes.c_float)]),
}


def clean_child_env():
    """Environment for child processes started after this process may have initialised the GPU: without the profiler's
    preload (under `rocprofv3 --pmc` the preloaded tool initialises the GPU in every child before it can exec make / gcc /
    hipcc, which this pool forbids) and without its ROCP_* / ROCPROF* configuration."""
    env = {k: v for k, v in os.environ.items()
           if k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROF", "ROCTX_", "ROCPROFILER_"))}
    return env


def build():
    """Compile the gfx950 library in-tree (same recipe as `make`)."""
    subprocess.check_call(["make", "-s", "-j8", "-C", _ROOT, os.path.relpath(LIB_PATH, _ROOT)], env=clean_child_env())


def _try_load():
    global _lib, loaded, load_error
    try:
        try:
            # share PyTorch's HIP runtime when torch is importable: both libamdhip64 copies carry the
            # SONAME libamdhip64.so.7, the first one loaded wins, and device pointers must come
            # from ONE runtime when torch tensors are handed to the *_dev entry points.
            import torch  # noqa: F401
        except Exception:
            pass
        if not os.path.exists(LIB_PATH):
            build()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        loaded = True
    except Exception as e:     # mirrors triangulation_c/__init__.py:10-11, but see lib()
        load_error = e
        loaded = False


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it is unavailable."""
    if not loaded:
        raise RuntimeError("libmqslam_hip.so is not available (%r); this package has no CPU path. "
                           "Build it with `make` (needs hipcc)." % (load_error,))
    return _lib


def check(rc):
    if rc != MQS_OK:
        msg = lib().mqs_last_error()
        raise RuntimeError("libmqslam_hip call failed (%d): %s" % (rc, msg.decode() if msg else ""))


class Context:
    """Owns an mqs_ctx (device staging buffers + stream) for the host-pointer entry points."""

    def __init__(self, device_id=0):
        self._h = c_vp()
        check(lib().mqs_create(int(device_id), ctypes.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            lib().mqs_destroy(self._h)
            self._h = c_vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    """Lazily created context on device 0 (one per process; 'one ctx per thread' applies)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("MQS_DEVICE", "0")))
    return _default_ctx


_try_load()
