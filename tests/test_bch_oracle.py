"""The outer code (BCH) of DVB-T2: the CPU checker oracle/bch_oracle.c against known answers. The reference has no BCH decoder
(bch_decoder.cpp:136 "TODO BCH decode") and no vectors for one, so the checker is pinned by the published code instead: the
generator degrees that the reference's (k_bch, n_bch) pairs imply (bch_decoder.cpp:79-134), polynomials of EN 302 755 tables
6a / 6b, an independent big-integer encoder (tests/t2_tx.py) and round trips. The library's host tables are held to the same."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
import t2_tx
from sdr_receiver_dvb_t2_amd._lib import lib


def _poly(*exps):
    return sum(1 << e for e in exps)


# EN 302 755 table 6a (normal FEC frames, GF(2^16)) and 6b (short, GF(2^14)): g_1, g_2, g_3 = minimal polynomials of alpha^1,3,5
STANDARD = {
    (16, 1): _poly(0, 2, 3, 5, 16),
    (16, 3): _poly(0, 1, 4, 5, 6, 8, 16),
    (16, 5): _poly(0, 2, 3, 4, 5, 7, 8, 9, 10, 11, 16),
    (14, 1): _poly(0, 1, 3, 5, 14),
    (14, 3): _poly(0, 6, 8, 11, 14),
    (14, 5): _poly(0, 1, 2, 6, 9, 10, 14),
}


def test_minimal_polynomials_match_the_standard():
    for (m, j), want in STANDARD.items():
        assert ol.ora_bch_minpoly(m, j) == want, (m, j)
    l = lib()
    for m, t in ((16, 12), (16, 10), (14, 12)):
        out = (ctypes.c_uint32 * t)()
        assert l.t2gpu_table_bch_minpoly(m, t, out) == t
        for i in range(t):
            assert out[i] == ol.ora_bch_minpoly(m, 2 * i + 1), (m, i)       # the library's host tables against the checker
            assert out[i] >> m == 1                                          # every one of degree m: m * t parity bits
    assert l.t2gpu_table_bch_minpoly(15, 12, (ctypes.c_uint32 * 12)()) == -1


@pytest.mark.parametrize("cid", range(12))
def test_generator_fills_the_parity_field(cid):
    m, t, kb, nb = ol.bch_params(cid)
    assert (m, t) == ((14, 12) if cid < 6 else (16, 10) if cid in (8, 11) else (16, 12))     # r = 2/3 and 5/6: 160 parity bits
    g = ol.ora_bch_generator(m, t)
    assert len(g) - 1 == nb - kb and g[0] == 1 and g[-1] == 1
    gi, mi, ti = t2_tx.bch_generator(cid)                                   # the big-integer construction agrees bit for bit
    assert (mi, ti) == (m, t) and gi == sum(int(b) << i for i, b in enumerate(g))
    fm, ft, fk, fn = (ctypes.c_int() for _ in range(4))
    assert lib().t2gpu_bch_info(cid // 6, cid % 6, ctypes.byref(fm), ctypes.byref(ft), ctypes.byref(fk), ctypes.byref(fn)) == 0
    assert (fm.value, ft.value, fk.value, fn.value) == (m, t, kb, nb)


@pytest.mark.parametrize("cid", [0, 3, 5, 6, 9, 11])
def test_encoders_agree_and_round_trip(cid):
    m, t, kb, nb = ol.bch_params(cid)
    rng = np.random.default_rng(100 + cid)
    msg = rng.integers(0, 2, (4, kb), dtype=np.uint8)
    msg[3] = 0
    cw = ol.ora_bch_encode(cid, msg)
    assert (cw[:, kb:] == t2_tx.bch_parity(cid, msg)).all()
    assert not cw[3].any()                                                  # linear code: zero message, zero parity
    assert ((cw[0] ^ cw[1]) == ol.ora_bch_encode(cid, msg[0:1] ^ msg[1:2])[0]).all()
    bad = cw.copy()
    plan = [[], [0, nb - 1] + list(rng.choice(np.arange(1, nb - 1), t - 2, replace=False)),      # t errors incl. both ends
            list(rng.choice(nb, t + 1, replace=False)), list(range(kb, kb + 3))]                 # t+1; a burst in the parity
    for f, pos in enumerate(plan):
        bad[f, pos] ^= 1
    out, st = ol.ora_bch_decode(cid, bad)
    assert list(st) == [0, t, -1, 3]
    assert (out[[0, 1, 3]] == cw[[0, 1, 3]]).all() and (out[2] == bad[2]).all()


def test_no_gpu_no_result():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    words = np.zeros((1, 32400), np.uint8)
    st = np.zeros(1, np.int32)
    assert lib().t2gpu_bch_decode(1, 0, words.ctypes.data, 1, st.ctypes.data) == -1
