"""ctypes loader for libt2gpu.so (built in-tree by ``__graft_entry__.build()`` / ``csrc/Makefile``)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class T2GpuError(RuntimeError):
    pass


def library_path():
    return os.path.join(_HERE, "libt2gpu.so")


_i8p = ctypes.POINTER(ctypes.c_int8)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_ip = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); must list every function include/t2gpu.h declares (tests/test_capi_symbols.py).
PROTOTYPES = {
    "t2gpu_version": (ctypes.c_int, []),
    "t2gpu_last_error": (ctypes.c_char_p, []),
    "t2gpu_host_pin": (ctypes.c_int, [_vp, ctypes.c_size_t]),
    "t2gpu_host_unpin": (ctypes.c_int, [_vp]),
    "t2gpu_handoff_enable": (ctypes.c_int, [ctypes.c_int]),
    "t2gpu_twin_attach": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_int]),
    "t2gpu_twin_detach": (ctypes.c_int, [_vp]),
    "t2gpu_twin_copy": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_int]),
    "t2gpu_device_count": (ctypes.c_int, []),
    "t2gpu_ldpc_create": (_vp, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "t2gpu_ldpc_destroy": (None, [_vp]),
    "t2gpu_ldpc_configure": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "t2gpu_ldpc_info": (ctypes.c_int, [_vp, _ip, _ip, _ip, _ip]),
    "t2gpu_ldpc_graph_stats": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _ip, _ip, _ip, _ip]),
    "t2gpu_ldpc_execute_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "t2gpu_ldpc_execute": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "t2gpu_ldpc_submit": (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    "t2gpu_ldpc_submit_add": (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    "t2gpu_ldpc_submit_go": (ctypes.c_int, [_vp]),
    "t2gpu_ldpc_collect": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp]),
    "t2gpu_ldpc_status": (ctypes.c_int, [_vp]),
    "t2gpu_ldpc_wait_resident": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_ldpc_set_submit_cu_reserve": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_ldpc_occupancy": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_ldpc_launch_workgroups": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_ldpc_set_max_slots": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_ldpc_set_plain_launch": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_ldpc_profile": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_ldpc_profile_layers": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_demap_create": (_vp, [ctypes.c_int] * 6),
    "t2gpu_demap_destroy": (None, [_vp]),
    "t2gpu_demap_configure": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_demap_execute_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_float, _vp, _vp, _vp]),
    "t2gpu_demap_execute": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "t2gpu_demap_stats_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_float, _vp, _vp]),
    "t2gpu_demap_llr_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "t2gpu_demap_llr_batch_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp]),
    "t2gpu_ti_create": (_vp, [ctypes.c_int] * 4),
    "t2gpu_ti_destroy": (None, [_vp]),
    "t2gpu_ti_cells_per_fec": (ctypes.c_int, [_vp]),
    "t2gpu_ti_begin": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_ti_push_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "t2gpu_ti_push": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp]),
    "t2gpu_ti_push_async": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp]),
    "t2gpu_ti_wait": (ctypes.c_int, [_vp]),
    "t2gpu_demap_stats_batch_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp, ctypes.c_int, _vp]),
    "t2gpu_ti_execute_blocks_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_long, _vp, ctypes.c_long, ctypes.c_int, _vp]),
    "t2gpu_ti_execute_blocks_stats_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_long, _vp, ctypes.c_long, ctypes.c_int, ctypes.c_float, _vp, ctypes.c_int, _vp]),
    "t2gpu_ti_execute_blocks_terms_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_long, _vp, ctypes.c_long, ctypes.c_int, _vp]),
    "t2gpu_demap_stats_terms_dev": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp, ctypes.c_int, _vp]),
    "t2gpu_bch_descramble_dev": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp]),
    "t2gpu_bch_descramble": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp]),
    "t2gpu_bch_info": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "t2gpu_table_bch_minpoly": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp]),
    "t2gpu_bch_decode_dev": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp]),
    "t2gpu_bch_decode": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp]),
    "t2gpu_l1_pre_parse": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_l1_post_parse": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int]),
    "t2gpu_front_reset_loops": (ctypes.c_int, [_vp]),
    "t2gpu_front_set_frequency_nco": (ctypes.c_int, [_vp, ctypes.c_float]),
    "t2gpu_front_set_iq": (ctypes.c_int, [_vp, ctypes.c_float, ctypes.c_float]),
    "t2gpu_front_hold_iq": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_front_set_chain": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_front_commit_iq": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_sync_reset": (None, [_vp, ctypes.c_float]),
    "t2gpu_sync_clear_frequency": (None, [_vp]),
    "t2gpu_sync_correct_resample": (None, [_vp, ctypes.c_double]),
    "t2gpu_demod_create": (_vp, [ctypes.c_int, ctypes.c_float, ctypes.c_int]),
    "t2gpu_demod_destroy": (None, [_vp]),
    "t2gpu_demod_connect": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_demod_execute": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp]),
    "t2gpu_demod_set_tuner": (ctypes.c_int, [_vp, ctypes.c_double]),
    "t2gpu_demod_set_chain_one": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_demod_set_copy_ahead": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_demod_flush": (ctypes.c_int, [_vp]),
    "t2gpu_demod_set_trace": (ctypes.c_int, [_vp, _vp, ctypes.c_long]),
    "t2gpu_demod_trace_count": (ctypes.c_long, [_vp]),
    "t2gpu_demod_set_device_loop": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_demod_status": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_eq_p2_frames_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_long, ctypes.c_int, _vp, _vp]),
    "t2gpu_eq_fc_frames_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_long, ctypes.c_long, _vp, _vp]),
    "t2gpu_eq_p2_frames_l1_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_long, ctypes.c_int, _vp, _vp, _vp]),
    "t2gpu_bch_descramble_pack_dev": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp]),
    "t2gpu_bbdh_execute_packed": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp]),
    "t2gpu_rx_flush_dev": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_rx_carry": (ctypes.c_int, [_vp]),
    "t2gpu_rx_ldpc_occupancy": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_rx_reset": (ctypes.c_int, [_vp]),
    "t2gpu_rx_fetch_packed": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp]),
    "t2gpu_rx_ts_enable": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "t2gpu_rx_ts_read": (ctypes.c_long, [_vp, _vp, ctypes.c_long, ctypes.c_int]),
    "t2gpu_rx_ts_counters_get": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "t2gpu_rx_create": (_vp, [_vp, ctypes.c_int]),
    "t2gpu_rx_pool_create": (_vp, [_vp, _vp, ctypes.c_int, ctypes.c_int]),
    "t2gpu_rx_pool_destroy": (None, [_vp]),
    "t2gpu_rx_pool_frame_alignment": (ctypes.c_int, [_vp]),
    "t2gpu_rx_pool_info": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_rx_pool_share": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "t2gpu_rx_pool_execute": (ctypes.c_long, [_vp, _vp, _vp, ctypes.c_int]),
    "t2gpu_rx_pool_ts_read": (ctypes.c_long, [_vp, _vp, ctypes.c_long]),
    "t2gpu_rx_pool_counters": (ctypes.c_int, [_vp, _vp, _vp]),
    "t2gpu_rx_destroy": (None, [_vp]),
    "t2gpu_rx_info": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_rx_set_overlap": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_rx_wait": (ctypes.c_int, [_vp]),
    "t2gpu_rx_front_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, _vp]),
    "t2gpu_rx_back_dev": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp]),
    "t2gpu_rx_execute_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, _vp, _vp, _vp]),
    "t2gpu_rx_results": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    "t2gpu_rx_fetch": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp]),
    "t2gpu_rx_stage_ms": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_rx_fft_eq_demap_dev": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "t2gpu_rx_sync_sums": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "t2gpu_rx_set_outer_code": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_rx_outer_code_status": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "t2gpu_ti_frame_plan": (ctypes.c_int, [ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int]),
    "t2gpu_bbdh_create": (_vp, [ctypes.c_int]),
    "t2gpu_bbdh_destroy": (None, [_vp]),
    "t2gpu_bbdh_execute": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp]),
    "t2gpu_bbdh_mode": (ctypes.c_int, [_vp]),
    "t2gpu_bbdh_resync_count": (ctypes.c_int, [_vp]),
    "t2gpu_bbdh_reset": (ctypes.c_int, [_vp]),
    "t2gpu_bbdh_execute_packed_rows": (ctypes.c_long, [_vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_long, ctypes.c_long, _vp, ctypes.c_int, _vp,
                                                        ctypes.c_long, _vp]),
    "t2gpu_ofdm_create": (_vp, [ctypes.c_int] * 8),
    "t2gpu_ofdm_destroy": (None, [_vp]),
    "t2gpu_fft_execute_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp]),
    "t2gpu_fft_execute": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int]),
    "t2gpu_fft_execute_strided_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp]),
    "t2gpu_eq_data_execute_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "t2gpu_eq_data_frames_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_long,
                                                ctypes.c_long, _vp, _vp]),
    "t2gpu_eq_p2_execute_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "t2gpu_eq_data_publish_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint, _vp, _vp]),
    "t2gpu_eq_fc_execute_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "t2gpu_ofdm_set_one_launch": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_fft_sym_sync_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint, _vp, _vp]),
    "t2gpu_sym_sync_dev": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_uint, _vp, _vp]),
    "t2gpu_front_loop_dev": (_vp, [_vp]),
    "t2gpu_front_loop_begin": (ctypes.c_int, [_vp, _vp, _vp]),
    "t2gpu_front_execute_loop_dev": (ctypes.c_long, [_vp, ctypes.c_int32, ctypes.c_double, _vp, _vp, _vp, ctypes.c_long, _vp]),
    "t2gpu_front_loop_follow": (ctypes.c_int, [_vp, ctypes.c_float, ctypes.c_float]),
    "t2gpu_front_loop_pending": (ctypes.c_int, [_vp]),
    "t2gpu_front_nco": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_front_loop_read": (ctypes.c_int, [_vp, _vp, _vp]),
    "t2gpu_sync_export": (None, [_vp, _vp]),
    "t2gpu_sync_candidates": (None, [_vp, _vp]),
    "t2gpu_eq_data_execute": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "t2gpu_ofdm_mode_info": (ctypes.c_int, [ctypes.c_int] * 6 + [_vp]),
    "t2gpu_table_symbol_carriers": (ctypes.c_int, [ctypes.c_int] * 7 + [_vp, _vp]),
    "t2gpu_table_freq_deint": (ctypes.c_int, [ctypes.c_int] * 7 + [_vp, _vp]),
    "t2gpu_table_bitdeint": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "t2gpu_table_cell_deint": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp]),
    "t2gpu_table_bb_prbs": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_front_create": (_vp, [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int]),
    "t2gpu_front_destroy": (None, [_vp]),
    "t2gpu_front_reset": (ctypes.c_int, [_vp]),
    "t2gpu_front_resample": (ctypes.c_int, [_vp, _vp, _vp]),
    "t2gpu_front_execute_dev": (ctypes.c_long, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_long, _vp, _vp]),
    "t2gpu_front_execute": (ctypes.c_long, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_long, _vp]),
    "t2gpu_front_state": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_front_committed_state": (ctypes.c_int, [_vp, _vp]),
    "t2gpu_front_debug_stream": (ctypes.c_long, [_vp, ctypes.c_int, _vp, ctypes.c_long]),
    "t2gpu_decim_execute": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp]),
    "t2gpu_farrow_execute": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_double, _vp, ctypes.c_int]),
    "t2gpu_cp_correlate_dev": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "t2gpu_cp_correlate_stream_dev": (ctypes.c_int, [_vp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "t2gpu_sync_create": (_vp, [ctypes.c_float]),
    "t2gpu_sync_destroy": (None, [_vp]),
    "t2gpu_sync_frequency": (None, [_vp, ctypes.c_float, ctypes.c_int]),
    "t2gpu_sync_symbol": (None, [_vp, ctypes.c_float, ctypes.c_float]),
    "t2gpu_sync_get": (None, [_vp, _vp]),
    "t2gpu_p1_create": (_vp, [ctypes.c_int, ctypes.c_int]),
    "t2gpu_p1_destroy": (None, [_vp]),
    "t2gpu_p1_reset": (ctypes.c_int, [_vp]),
    "t2gpu_p1_set_serial_detector": (ctypes.c_int, [_vp, ctypes.c_int]),
    "t2gpu_p1_execute_dev": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp]),
    "t2gpu_p1_execute": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp]),
    "t2gpu_p1_execute_batch_dev": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_float, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "t2gpu_p1_debug": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp]),
    "t2gpu_plan_nco": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_float, _vp, _vp]),
    "t2gpu_plan_farrow": (ctypes.c_long, [_vp, ctypes.c_int, ctypes.c_double, _vp, _vp, _vp]),
}

_lib = None


def lib():
    """Load libt2gpu.so once. Raises T2GpuError (never falls back) when the HIP library is not built."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise T2GpuError(
                "%s is missing: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)"
                % path)
        # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's). Import torch first so that the
        # process holds exactly one runtime -- the one that owns the tensors whose pointers cross this ABI.
        import torch  # noqa: F401
        l = ctypes.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().t2gpu_last_error()
        raise T2GpuError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
