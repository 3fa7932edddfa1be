"""ctypes views of the L1 signalling structs (include/t2gpu.h) and the two parse calls; mirrors
p2_symbol::l1_pre_info / l1_post_info (/root/reference/src/DVB_T2/p2_symbol.cpp:301-532,534-718)."""
import ctypes

import numpy as np

from ._lib import lib

_PRE = ("type bwt_ext s1 s2_field1 s2_field2 l1_repetition_flag guard_interval papr l1_post_mod l1_cod l1_fec_type l1_post_size "
        "l1_post_info_size pilot_pattern tx_id_availability cell_id network_id t2_system_id num_t2_frames num_data_symbols regen_flag "
        "l1_post_extension num_rf current_rf_index t2_version l1_post_scrambled t2_base_lite reserved").split()
_PLP = ("id plp_type plp_payload_type ff_flag first_rf_idx first_frame_idx plp_group_id plp_cod plp_mod plp_rotation plp_fec_type "
        "plp_num_blocks_max frame_interval time_il_length time_il_type in_band_a_flag in_band_b_flag reserved_1 plp_mode static_flag "
        "static_padding_flag").split()


class l1_presignalling(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in _PRE] + [("crc_32", ctypes.c_uint32)]


class l1_postsignalling_plp(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in _PLP]


class dynamic_plp(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("id", "start", "num_blocks", "reserved_2")]


class l1_postsignalling(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("sub_slices_per_frame", "num_plp", "num_aux", "aux_config_rfu")] + \
               [("rf_idx", ctypes.c_int32 * 8), ("frequency", ctypes.c_uint32 * 8)] + \
               [(n, ctypes.c_int32) for n in ("fef_type", "fef_length", "fef_interval", "fef_length_msb", "reserved_2", "frame_idx",
                                               "sub_slice_interval", "type_2_start", "l1_change_counter", "start_rf_idx",
                                               "dyn_reserved_1", "dyn_reserved_3")]


def l1_pre_info(p2_cells):
    """p2_cells: complex64 equalised P2 cells. Returns (crc_ok, l1_presignalling)."""
    c = np.ascontiguousarray(p2_cells, np.complex64)
    pre = l1_presignalling()
    rc = lib().t2gpu_l1_pre_parse(c.ctypes.data, ctypes.byref(pre))
    return rc == 1, pre


def l1_post_info(p2_cells, pre, max_plp=16):
    c = np.ascontiguousarray(p2_cells, np.complex64)[1840:]
    post = l1_postsignalling()
    plp = (l1_postsignalling_plp * max_plp)()
    dyn = (dynamic_plp * max_plp)()
    rc = lib().t2gpu_l1_post_parse(c.ctypes.data, ctypes.byref(pre), ctypes.byref(post), plp, dyn, max_plp)
    return rc == 1, post, plp, dyn
