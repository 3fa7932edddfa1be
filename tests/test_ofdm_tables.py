"""CPU tier: OFDM-side mode arithmetic and tables the product builds (csrc/ofdm_tables.cpp) against the reference-shaped
oracle (oracle/ofdm_oracle.c), and the oracle's fast_math against the reference header itself (oracle/_ref/libref_dsp.so)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol

MODES = [  # fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data
    (5, 1, 6, 4, 0, 59), (5, 0, 6, 4, 0, 59), (5, 1, 3, 0, 0, 60), (5, 1, 1, 2, 2, 20), (5, 0, 7, 1, 0, 30), (5, 1, 5, 1, 2, 41),
    (4, 1, 6, 4, 0, 100), (4, 0, 0, 3, 0, 17), (4, 1, 2, 2, 2, 33), (4, 1, 7, 1, 0, 64), (4, 0, 4, 1, 0, 45), (4, 1, 1, 3, 0, 9)]


@pytest.fixture(scope="module")
def l(built):
    import sdr_receiver_dvb_t2_amd as pkg
    return pkg.lib()


@pytest.mark.parametrize("mode", MODES)
def test_mode_arithmetic_and_carrier_tables(l, mode):
    m = ol.ora_mode(*mode)
    assert m is not None
    info = (ctypes.c_int * 12)()
    assert l.t2gpu_ofdm_mode_info(*mode, info) == 0
    want = [m.fft_size, m.k_total, m.k_ext, m.k_offset, m.l_nulls, m.c_p2, m.c_data, m.n_fc, m.c_fc, m.l_fc, m.len_frame]
    assert list(info)[:11] == want
    for idx in sorted({0, 1, 2, 3, 4, 5, 16, 17, m.len_frame - 2, m.len_frame - 1}):
        if idx >= m.len_frame:
            continue
        mp = np.zeros(m.k_total, np.uint8)
        rf = np.zeros(m.k_total, np.float32)
        assert l.t2gpu_table_symbol_carriers(*mode, idx, mp.ctypes.data, rf.ctypes.data) == m.k_total
        wmp, wrf = ol.ora_symbol_carriers(m, idx)
        assert np.array_equal(mp.astype(np.int32), wmp), idx
        assert np.array_equal(rf, wrf), idx
        ndata = int((wmp == 1).sum())
        if idx == 0:
            assert ndata == m.c_p2                                   # tables 47-49 of EN 302 755 via the reference's constants
        elif m.l_fc and idx == m.len_frame - 1:
            assert ndata == m.n_fc
        else:
            assert ndata == m.c_data


@pytest.mark.parametrize("mode", MODES)
def test_frequency_deinterleaver_tables(l, mode):
    m = ol.ora_mode(*mode)
    for kind in (0, 1, 2):
        cells = [m.c_p2, m.c_data, m.n_fc][kind]
        if cells == 0:
            continue
        he, ho = np.zeros(cells, np.int32), np.zeros(cells, np.int32)
        assert l.t2gpu_table_freq_deint(*mode, kind, he.ctypes.data, ho.ctypes.data) == cells
        whe, who = ol.ora_freq_deint(m, kind)
        assert np.array_equal(he, whe) and np.array_equal(ho, who)
        assert np.array_equal(np.sort(ho), np.arange(cells)) and np.array_equal(np.sort(he), np.arange(cells))
        if m.is32k:
            assert np.array_equal(he[ho], np.arange(cells))            # 32K: one permutation and its inverse


def _ref_dsp():
    import os
    path = os.path.join(ol.ROOT, "oracle", "_ref", "libref_dsp.so")
    if not os.path.exists(path):
        return None
    r = ctypes.CDLL(path)
    r.ref_lut_init()
    return r


@pytest.mark.skipif(_ref_dsp() is None, reason="oracle/_ref/libref_dsp.so not built here")
def test_fast_math_restatement_is_the_reference():
    """LUT contents, LUT lookups and atan2_approx of the oracle == the reference header compiled with its own flags."""
    o, r = ol.oracle(), _ref_dsp()
    for lib_, pre in ((o, "ora_"), (r, "ref_")):
        f = getattr(lib_, pre + "lut_table"); f.restype = ctypes.POINTER(ctypes.c_float * 65536); f.argtypes = [ctypes.c_int]
        for n in ("sin_lut", "cos_lut"):
            g = getattr(lib_, pre + n); g.restype = ctypes.c_float; g.argtypes = [ctypes.c_float]
        g = getattr(lib_, pre + "atan2_approx"); g.restype = ctypes.c_float; g.argtypes = [ctypes.c_float] * 2
    o.ora_lut_init()
    for w in (0, 1):
        assert np.array_equal(np.array(o.ora_lut_table(w).contents), np.array(r.ref_lut_table(w).contents))
    rng = np.random.Generator(np.random.PCG64(3))
    xs = rng.uniform(-9, 9, 4000).astype(np.float32)
    assert all(o.ora_sin_lut(float(x)) == r.ref_sin_lut(float(x)) and o.ora_cos_lut(float(x)) == r.ref_cos_lut(float(x)) for x in xs)
    ys = np.concatenate([rng.standard_normal(4000), [0, 0, 1, -1, 0]]).astype(np.float32)
    zs = np.concatenate([rng.standard_normal(4000), [1, -1, 0, 0, 0]]).astype(np.float32)
    assert all(o.ora_atan2_approx(float(y), float(x)) == r.ref_atan2_approx(float(y), float(x)) for y, x in zip(ys, zs))
