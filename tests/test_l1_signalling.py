"""CPU tier: L1-pre / L1-post extraction (host code of the product, csrc/l1_signalling.cpp, restating p2_symbol.cpp:301-1089)
against the transmitter model: every field written by t2_tx comes back, for all four L1-post constellations, with and
without L1-post scrambling; a flipped systematic bit is rejected by the CRC-32."""
import numpy as np
import pytest

import t2_tx


@pytest.fixture(scope="module")
def l1(built):
    from sdr_receiver_dvb_t2_amd import l1 as l1mod
    return l1mod


PRE = dict(type=0, bwt_ext=1, s1=0, s2_field1=5, s2_field2=0, guard_interval=4, papr=0, l1_post_mod=1, l1_cod=0, l1_fec_type=0,
           l1_post_size=350, l1_post_info_size=0, pilot_pattern=6, cell_id=0x1234, network_id=0x3085, t2_system_id=0x8001,
           num_t2_frames=2, num_data_symbols=59, num_rf=1, t2_version=2)
PLP = dict(id=0, plp_type=1, plp_payload_type=3, plp_cod=3, plp_mod=3, plp_rotation=1, plp_fec_type=1, plp_num_blocks_max=202,
           frame_interval=1, time_il_length=1, time_il_type=0, plp_mode=2)


@pytest.mark.parametrize("mod,size,scrambled", [(0, 700, False), (1, 350, False), (1, 350, True), (2, 200, False), (3, 140, True)])
def test_l1_round_trip(l1, mod, size, scrambled):
    dyn = [dict(id=0, start=0, num_blocks=201)]
    info = t2_tx.l1_post_bits(dict(frame_idx=1, l1_change_counter=0), [PLP], dyn)
    pre_fields = dict(PRE, l1_post_mod=mod, l1_post_size=size, l1_post_info_size=len(info), l1_post_scrambled=int(scrambled))
    cells = np.concatenate([t2_tx.l1_pre_cells(pre_fields, seed=mod), t2_tx.l1_post_cells(info, mod, size, seed=mod, scrambled=scrambled)])
    rng = np.random.Generator(np.random.PCG64(mod))
    noisy = (cells + 0.03 * (rng.standard_normal(cells.size) + 1j * rng.standard_normal(cells.size))).astype(np.complex64)
    ok, pre = l1.l1_pre_info(noisy)
    assert ok
    for k, v in pre_fields.items():
        assert getattr(pre, k) == v, k
    ok2, post, plp, dynp = l1.l1_post_info(noisy, pre)
    assert ok2 and post.num_plp == 1 and post.num_aux == 0 and post.frame_idx == 1
    for k, v in PLP.items():
        assert getattr(plp[0], k) == v, k
    assert (dynp[0].id, dynp[0].start, dynp[0].num_blocks) == (0, 0, 201)
    bad = noisy.copy(); bad[17] = -bad[17]                      # a systematic L1-pre bit
    assert not l1.l1_pre_info(bad)[0]
    bad = noisy.copy(); bad[1840 + 3] = -bad[1840 + 3]          # a systematic L1-post bit (sign bits of the first cells)
    assert not l1.l1_post_info(bad, pre)[0]


# ---- against the reference's own p2_symbol (tests/golden/t2sym_golden.npz, made by tests/golden/make_t2_golden.py) ----
import os

import oracle_lib as ol
import ref_cases as rc


@pytest.mark.parametrize("name", list(rc.SYM_MODES))
def test_equals_the_reference_class(l1, name):
    """The reference's p2_symbol::execute equalised these P2 symbols and parsed them (l1_pre_info / l1_post_info); the product's
    parser on the same equalised cells returns the same L1-pre (all 29 fields incl. CRC_32) and the same L1-post: counts, RF list,
    every configurable and dynamic PLP field, frame index. (Fields the reference mis-reads -- RESERVED_2 as 2 of its 30 bits,
    RESERVED_3 through `=` instead of `|=`, p2_symbol.cpp:890-893,994-996 -- are zero on air and are not compared.)"""
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t2sym_golden.npz")) as z:
        cells, want_pre, want_post = z["sym/%s/p2_cells" % name], z["sym/%s/l1_pre" % name], z["sym/%s/l1_post" % name]
    ok, pre = l1.l1_pre_info(cells)
    assert ok
    for k, v in zip(ol.L1_PRE_NAMES, want_pre):
        assert getattr(pre, k) == int(v) & 0xffffffff if k == "crc_32" else getattr(pre, k) == int(v), k
    ok2, post, plp, dyn = l1.l1_post_info(cells, pre)
    assert ok2
    w = ol.unpack_l1_post(want_post)
    for k in ("sub_slices_per_frame", "num_plp", "num_aux", "aux_config_rfu", "fef_type", "fef_length", "fef_interval", "fef_length_msb",
              "frame_idx", "sub_slice_interval", "type_2_start", "l1_change_counter", "start_rf_idx", "dyn_reserved_1"):
        assert getattr(post, k) == w[k], k
    assert [(post.rf_idx[i], post.frequency[i]) for i in range(pre.num_rf)] == [(a, b & 0xffffffff) for a, b in w["rf"]]
    for i, wp in enumerate(w["plp"]):
        for k in l1mod_fields():
            assert getattr(plp[i], k) == wp[k], (i, k)
        assert (dyn[i].id, dyn[i].start, dyn[i].num_blocks) == (wp["dyn_id"], wp["dyn_start"], wp["dyn_num_blocks"])


def l1mod_fields():
    from sdr_receiver_dvb_t2_amd import l1 as l1mod
    return l1mod._PLP


def test_counts_larger_than_the_block_are_refused(l1):
    """NUM_PLP / NUM_AUX that do not fit L1_POST_INFO_SIZE (CRC-valid, crafted) are rejected before any field behind the
    block is touched."""
    plps = [dict(PLP, id=i) for i in range(3)]
    dyn = [dict(id=i, start=0, num_blocks=1) for i in range(3)]
    info = t2_tx.l1_post_bits(dict(), plps, dyn)
    info[15:23] = t2_tx.bits_of(200, 8)                           # NUM_PLP = 200 in a block that holds three
    pre_fields = dict(PRE, l1_post_mod=1, l1_post_size=350, l1_post_info_size=len(info))
    cells = np.concatenate([t2_tx.l1_pre_cells(pre_fields, seed=1), t2_tx.l1_post_cells(info, 1, 350, seed=1)]).astype(np.complex64)
    ok, pre = l1.l1_pre_info(cells)
    assert ok
    assert not l1.l1_post_info(cells, pre, max_plp=255)[0]
