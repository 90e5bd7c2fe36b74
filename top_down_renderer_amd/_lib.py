"""ctypes binding of libtdr_hip.so (include/tdr.h).  There is no CPU fallback: if the library is missing or no HIP
device is present, the product path raises."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("TDR_LIB_PATH") or os.path.join(_PKG, "libtdr_hip.so")   # override: tuning variants only

TDR_ST_FIELDS = 7
TDR_MAX_CLASSES = 15


class FilterParamsC(C.Structure):
    _fields_ = [
        ("pos_cov", C.c_float), ("theta_cov", C.c_float), ("regularization", C.c_float),
        ("init_pos_px_x", C.c_float), ("init_pos_px_y", C.c_float), ("init_pos_px_cov", C.c_float),
        ("init_pos_m_x", C.c_float), ("init_pos_m_y", C.c_float),
        ("init_pos_deg_theta", C.c_float), ("init_pos_deg_cov", C.c_float),
        ("force_on_map", C.c_int32),
        ("fixed_scale", C.c_float), ("scale_log_min", C.c_float), ("scale_log_max", C.c_float),
        ("num_classes", C.c_int32),
        ("class_weights", C.c_float * 16),
    ]


class MapDescC(C.Structure):
    _fields_ = [("rec", C.c_void_p), ("ncls", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("rec_floats", C.c_int32), ("resolution", C.c_float),
                ("cwords", C.c_int32), ("dict_n", C.c_int32), ("crec", C.c_void_p), ("dict", C.c_void_p),
                ("rec16", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/tdr.h declares
_vp, _i, _i64, _f, _u64, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_uint32
SIGNATURES = {
    "tdr_last_error": (C.c_char_p, []),
    "tdr_version": (_i, []),
    "tdr_device_count": (_i, []),
    "tdr_rec_floats": (_i, [_i]),
    "tdr_map_rec_floats_total": (C.c_size_t, [_i, _i, _i]),
    "tdr_map_rec16_bytes": (C.c_size_t, [_i, _i, _i]),
    "tdr_config_rec16_min_particles": (_i64, [_i64]),
    "tdr_config_compact": (_i, [_i]),
    "tdr_config_shift_uniform": (_i, [_i]),
    "tdr_config_shift_uniform_span": (C.c_float, [C.c_float]),
    "tdr_config_ray_split": (_i, [_i]),
    "tdr_config_cart_skip": (_i, [_i]),
    "tdr_config_init_mfma": (_i, [_i]),
    "tdr_config_uw_waves": (_i, [_i]),
    "tdr_config_prefix_small": (_i, [_i]),
    "tdr_shift_uniform_launches": (_i64, []),
    "tdr_cmap_words": (_i, [_i]),
    "tdr_cmap_words_total": (C.c_size_t, [_i, _i, _i]),
    "tdr_cmap_tile_words": (C.c_size_t, [_i, _i, _i]),
    "tdr_cmap_plane_offset_words": (C.c_size_t, [_i, _i, _i]),
    "tdr_cmap_plane_words": (C.c_size_t, [_i, _i, _i]),
    "tdr_cmap_cmask_words": (C.c_size_t, [_i, _i, _i]),
    "tdr_k_compact_map": (_i, [C.POINTER(MapDescC), _vp, _vp, _vp, _vp]),
    "tdr_cmap_wide_words_total": (C.c_size_t, [_i, _i, _i]),
    "tdr_k_compact_map_wide": (_i, [C.POINTER(MapDescC), _vp, _vp, _vp, _vp]),
    "tdr_k_unpack_compact_map": (_i, [C.POINTER(MapDescC), _vp, _vp]),
    "tdr_k_selftest_atan2": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "tdr_libm_variant": (_i, []),
    "tdr_libm_force_variant": (_i, [_i]),
    "tdr_sincosf_host": (_i, [_vp, _i64, _i, _vp, _vp]),
    "tdr_k_selftest_sincos": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "tdr_k_selftest_round": (_i, [_vp, _i64, _f, _vp, _vp]),
    "tdr_k_pack_map": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "tdr_map_ingest_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "tdr_map_ingest_shape": (_i, [_i, _i, _f, C.POINTER(_i), C.POINTER(_i)]),
    "tdr_k_map_from_labels": (_i, [_vp, _i, _i, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "tdr_k_map_from_rasters": (_i, [_vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "tdr_map_save_rasters": (_i, [_vp, C.c_char_p]),
    "tdr_png_read_gray8_host": (_i, [C.c_char_p, _vp, _i64, _vp, _vp]),
    "tdr_png_write_gray8_host": (_i, [C.c_char_p, _vp, _i, _i]),
    "tdr_map_load_rasters": (_i, [_vp, C.c_char_p, _i, _f, _i, _i]),
    "tdr_k_unpack_map": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "tdr_polar_table_host": (_i, [_i, _i, _f, _f, _vp]),
    "tdr_raster_workspace_bytes": (_i64, [_i64]),
    "tdr_k_raster_polar": (_i, [_vp, _i, _i, _i64, _f, _f, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "tdr_k_raster_cart": (_i, [_vp, _i, _i, _i64, _f, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "tdr_raster_geo_workspace_bytes": (_i64, [_i64]),
    "tdr_k_raster_geo_polar": (_i, [_vp, _i, _i64, _i64, _f, _f, _i, _i, _vp, _vp, _vp]),
    "tdr_k_raster_geo_cart": (_i, [_vp, _i, _i64, _i64, _f, _i, _i, _vp, _vp]),
    "tdr_k_pack_scan": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "tdr_score_workspace_floats": (C.c_size_t, [_i, _i, _i, _i64, _i64]),
    "tdr_k_score_polar": (_i, [C.POINTER(MapDescC), _vp, _vp, _i, _i, _f, C.POINTER(FilterParamsC), _vp, _i64, _i64,
                               _i64, _vp, _f, _i, _vp, _vp, _vp]),
    "tdr_k_active_diffs": (_i, [C.POINTER(MapDescC), _vp, _i, _i, _f, _vp, _vp, _i, _i, _vp, _vp]),
    "tdr_active_candidates_host": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, C.POINTER(_i), C.POINTER(_i)]),
    "tdr_logf_host": (_i, [_vp, _i64, _vp]),
    "tdr_k_selftest_logf": (_i, [_vp, _i64, _vp, _vp]),
    "tdr_rng_pipe_create": (_i, [_i64, C.POINTER(_vp)]),
    "tdr_rng_pipe_destroy": (None, [_vp]),
    "tdr_rng_pipe_on_device": (_i, [_vp]),
    "tdr_rng_pipe_from_host": (_i, [_vp, _vp, _vp]),
    "tdr_rng_pipe_to_host": (_i, [_vp, _vp, _vp]),
    "tdr_rng_pipe_normals": (_i, [_vp, _i64, _i64, _i64, _i, C.POINTER(_vp), _vp]),
    "tdr_rng_pipe_uniform": (_i, [_vp, C.POINTER(_vp), _vp]),
    "tdr_rng_dev_workspace_bytes": (C.c_size_t, [_i64]),
    "tdr_k_rng_propagate_normals": (_i, [_vp, _i64, _i64, _i64, _i, _vp, _vp, _vp]),
    "tdr_k_rng_uniform": (_i, [_vp, _vp, _vp]),
    "tdr_rng_get_state_host": (_i, [_vp, _vp]),
    "tdr_rng_set_state_host": (_i, [_vp, _vp]),
    "tdr_k_resample_dev": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp]),
    "tdr_score_ctx_create": (_i, [C.POINTER(_vp)]),
    "tdr_score_ctx_destroy": (None, [_vp]),
    "tdr_score_ctx_span": (C.c_float, [_vp]),
    "tdr_score_ctx_trial_calls": (_i64, [_vp]),
    "tdr_config_tuning": (_i64, [C.c_char_p, _i64]),
    "tdr_selftest_score": (_i, []),
    "tdr_profile_variants": (_i, [C.POINTER(_i64)]),
    "tdr_score_ctx_set_polar_factors": (_i, [_vp, _vp, _i, _i]),
    "tdr_polar_factors_host": (_i, [_i, _i, _f, _f, _vp]),
    "tdr_k_score_polar_ctx": (_i, [C.POINTER(MapDescC), _vp, _vp, _i, _i, _f, C.POINTER(FilterParamsC), _vp, _i64, _i64,
                                   _i64, _vp, _f, _i, _vp, _vp, _vp, _vp]),
    "tdr_score_geo_workspace_floats": (C.c_size_t, [_i, _i, _i, _i64, _i64]),
    "tdr_k_score_polar_geo": (_i, [C.POINTER(MapDescC), C.POINTER(MapDescC), _vp, _vp, _vp, _f, _f, _i, _i, _f,
                                   C.POINTER(FilterParamsC), _vp, _i64, _i64, _i64, _vp, _f, _i, _vp, _vp, _vp]),
    "tdr_score_cart_workspace_floats": (C.c_size_t, [_i, _i, _i, _i64, _i64]),
    "tdr_k_score_cart": (_i, [C.POINTER(MapDescC), _vp, _i, _i, _f, C.POINTER(FilterParamsC), _vp, _i64, _i64, _i64,
                              _vp, _vp, _vp, _vp]),
    "tdr_k_propagate": (_i, [_vp, _i64, _i64, _vp, _f, _f, _f, _i, _f, _f, _vp, _u64, _u64, _i64, _vp]),
    "tdr_rng_create": (_vp, [_u32]),
    "tdr_rng_destroy": (None, [_vp]),
    "tdr_rng_uniform_host": (_f, [_vp]),
    "tdr_propagate_normals_host": (_i, [_vp, _i64, _i, _vp]),
    "tdr_k_update_weights": (_i, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "tdr_prefix_workspace_bytes": (_i64, [_i64]),
    "tdr_k_prefix": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "tdr_k_prefix_mode": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "tdr_k_resample": (_i, [_vp, _i64, _i64, _f, _i64, _i64, _vp, _vp]),
    "tdr_k_gather_states": (_i, [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "tdr_init_particles_host": (_i, [_vp, _vp, _i, _i, _i, _f, C.POINTER(FilterParamsC), _i, _vp, C.POINTER(C.c_int64)]),
    "tdr_k_mean_cov": (_i, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "tdr_k_sample_ml_states": (_i, [_vp, _i64, _i64, _i, _vp, _vp]),
    "tdr_gmm_fit_host": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "tdr_gmm_select_host": (_i, [_vp, _i, _i64, _vp, _i, _vp, _vp]),
    "tdr_adaptive_count_host": (_i64, [_vp, _i, _i64, _i64]),
    "tdr_filter_compute_gmm": (_i, [_vp]),
    "tdr_filter_get_gmm": (_i, [_vp, _i, _vp, _vp, _vp]),
    "tdr_filter_adaptive_count": (_i64, [_vp]),
    "tdr_k_save_ml_state": (_i, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "tdr_k_shard_pack2": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "tdr_k_shard_unpack2": (_i, [_vp, _i, _i64, _vp, _vp, _vp]),
    "tdr_k_unshard_states": (_i, [_vp, _i, _i64, _vp, _i64, _vp]),
    "tdr_comm_rccl_unique_id": (_i, [_vp]),
    "tdr_comm_create_rccl": (_i, [_i, _i, _vp, C.POINTER(_vp)]),
    "tdr_comm_create": (_i, [_i, _i, _vp, C.POINTER(_vp)]),
    "tdr_comm_destroy": (None, [_vp]),
    "tdr_comm_world": (_i, [_vp]),
    "tdr_comm_rank": (_i, [_vp]),
    "tdr_comm_all_gather": (_i, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "tdr_comm_broadcast": (_i, [_vp, _vp, C.c_size_t, _i, _vp]),
    "tdr_filter_create_sharded": (_i, [_vp, _i, C.POINTER(FilterParamsC), _u32, _vp, C.POINTER(_vp)]),
    "tdr_filter_num_local": (_i64, [_vp]),
    "tdr_k_set_scale": (_i, [_vp, _i64, _i64, _vp, _vp]),
    "tdr_k_shift_init": (_i, [_vp, _i64, _i64, _f, _f, _vp]),
    "tdr_k_states_aos_to_soa": (_i, [_vp, _i64, _vp, _i64, _vp]),
    "tdr_k_states_soa_to_aos": (_i, [_vp, _i64, _i64, _vp, _vp]),
    "tdr_profile_enable": (_i, [_i]),
    "tdr_profile_score_ms": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "tdr_profile_shares": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    # handle layer (csrc/tdr_host.cpp)
    "tdr_map_create": (_i, [C.POINTER(_vp)]),
    "tdr_map_destroy": (None, [_vp]),
    "tdr_map_set": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _i, _i]),
    "tdr_map_set_labels": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _f, _i, _i]),
    "tdr_filter_update_map_labels": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _f, _i, _i]),
    "tdr_map_sample_pts_polar": (_i, [_vp, _i, _i, _f]),
    "tdr_map_polar_shape": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "tdr_map_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), C.POINTER(_i)]),
    "tdr_map_center": (_i, [_vp, _vp, _vp]),
    "tdr_map_local_map": (_i, [_vp, _i, _f, _f, _f, _f, _i, _i, _vp, _vp]),
    "tdr_k_local_map_polar": (_i, [_vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp]),
    "tdr_k_local_map_cart": (_i, [_vp, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp]),
    "tdr_map_local_geo_map": (_i, [_vp, _i, _f, _f, _f, _f, _i, _i, _vp]),
    "tdr_map_load_cache": (_i, [_vp, C.c_char_p, C.c_char_p, _i, _f, _i, _i, C.POINTER(_i)]),
    "tdr_map_save_cache": (_i, [_vp, C.c_char_p, C.c_char_p]),
    "tdr_k_geo_map_from_map": (_i, [C.POINTER(MapDescC), _i, _vp, _vp, _vp]),
    "tdr_map_classes_at_point": (_i, [_vp, _i, _i, C.POINTER(_u32)]),
    "tdr_map_best_rel_pos": (_i, [_vp, _vp, _i, _vp, _vp]),
    "tdr_renderer_create": (_i, [_vp, C.POINTER(_vp)]),
    "tdr_renderer_destroy": (None, [_vp]),
    "tdr_renderer_render": (_i, [_vp, _i, _vp, _i, _i, _i64, _f, _f, _i, _i, _i, _vp]),
    "tdr_renderer_render_geo": (_i, [_vp, _i, _vp, _i, _i64, _i64, _f, _f, _i, _i, _vp]),
    "tdr_filter_create": (_i, [_vp, _i, C.POINTER(FilterParamsC), _u32, C.POINTER(_vp)]),
    "tdr_filter_destroy": (None, [_vp]),
    "tdr_filter_configure": (_i, [_vp, _i, _i]),
    "tdr_filter_initialize_particles": (_i, [_vp]),
    "tdr_filter_set_states": (_i, [_vp, _vp, _i64]),
    "tdr_filter_get_states": (_i, [_vp, _vp, _i64]),
    "tdr_filter_propagate": (_i, [_vp, _f, _f, _f]),
    "tdr_filter_update": (_i, [_vp, _vp, _vp, _f, _i64]),
    "tdr_filter_update_geo": (_i, [_vp, _vp, _vp, _f, _i64]),
    "tdr_filter_compute_weights": (_i, [_vp, _vp, _vp, _f]),
    "tdr_filter_get_raw_weights": (_i, [_vp, _vp, _i64]),
    "tdr_filter_get_last_dist": (_i, [_vp, _vp, _i64]),
    "tdr_filter_propagate_freeze": (_i, [_vp, _f, _f, _f, _i]),
    "tdr_filter_init_one": (_i, [_vp]),
    "tdr_filter_share_rng": (_i, [_vp, _vp]),
    "tdr_init_particle_host": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "tdr_filter_get_weights": (_i, [_vp, _vp, _i64]),
    "tdr_filter_get_resample_indices": (_i, [_vp, _vp, _i64]),
    "tdr_filter_mean_cov": (_i, [_vp, _i, _vp, _vp]),
    "tdr_filter_freeze_scale": (_i, [_vp]),
    "tdr_filter_is_scale_frozen": (_i, [_vp]),
    "tdr_filter_scale": (_f, [_vp]),
    "tdr_filter_num_particles": (_i64, [_vp]),
    "tdr_filter_update_map": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _i, _i]),
    "tdr_set_error": (_i, [_i, C.c_char_p]),
    "tdr_locality_tmp_ints": (C.c_size_t, [_i64, _i, _i]),
    "tdr_locality_pose_tmp_ints": (C.c_size_t, [_i64]),
    "tdr_k_locality_order_pose": (_i, [_vp, _i64, _i64, _i, _i, _f, _vp, _vp, _vp]),
    "tdr_k_locality_order": (_i, [_vp, _i64, _i64, _i, _i, _vp, _vp, _vp]),
}

_lib = None


class TdrError(RuntimeError):
    pass


def load():
    """Loads libtdr_hip.so and binds every entry point; raises TdrError if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise TdrError(
            f"{SO_PATH} is missing: build the HIP extension first (python -m top_down_renderer_amd.build, or "
            "__graft_entry__.build()).  There is no CPU fallback.")
    L = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise TdrError(f"libtdr_hip error {rc}: {load().tdr_last_error().decode()}")
