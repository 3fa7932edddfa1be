"""CPU tier: the P1 restatement (oracle/p1_oracle.c). Its delay lines and running sums are pinned against the reference's
buffers.hh compiled unmodified (oracle/_ref/libref_dsp*.so); the detector as a whole is exercised on P1 symbols built by the
transmitter model (tests/t2_tx.py, EN 302 755 9.8)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
import t2_tx

needs_ref = pytest.mark.skipif(ol.ref_dsp() is None or ol.ref_dsp(True) is None, reason="oracle/_ref/libref_dsp.so not built here")


def noise(rng, n, s):
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * s).astype(np.complex64)


@needs_ref
def test_delay_and_running_sum_vs_reference_classes():
    rng = np.random.Generator(np.random.PCG64(1))
    x = noise(rng, 5000, 1.0)
    o = ol.oracle()
    for lib, exact in ((ol.ref_dsp(True), True), (ol.ref_dsp(), False)):
        for delay in (482, 542, 964, 2):
            a, b = np.zeros_like(x), np.zeros_like(x)
            lib.ref_delay_run(delay, len(x), x.ctypes.data_as(ctypes.c_void_p), a.ctypes.data_as(ctypes.c_void_p))
            o.ora_delay_run(delay, len(x), x.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p))
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            assert np.array_equal(a[delay:], x[:-delay]) and not a[:delay].any()
        for which, ln in ((0, 482), (1, 542)):
            a, b = np.zeros_like(x), np.zeros_like(x)
            lib.ref_sum_run(which, len(x), x.ctypes.data_as(ctypes.c_void_p), a.ctypes.data_as(ctypes.c_void_p))
            o.ora_sum_run(ln, len(x), x.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p))
            if exact:
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            np.testing.assert_allclose(a, b, rtol=0, atol=2e-4)
            win = np.convolve(x.astype(np.complex128), np.ones(ln - 1))[:len(x)]        # LEN-1 terms (buffers.hh:33-39)
            np.testing.assert_allclose(b, win, rtol=0, atol=2e-3)


@pytest.mark.parametrize("s1,s2,shift_carriers", [(0, 10, 0), (0, 8, 0), (1, 11, 0), (0, 10, 3), (0, 10, -7), (3, 2, 9)])
def test_p1_detect_and_decode(s1, s2, shift_carriers):
    rng = np.random.Generator(np.random.PCG64(10 * s1 + s2))
    p1 = t2_tx.p1_symbol(s1, s2) * 0.3
    t = np.arange(2048)
    p1 = p1 * np.exp(2j * np.pi * (shift_carriers + 0.02) * t / 1024.0)                  # integer carrier offset + a fraction
    lead = 2500 + 100 * s2
    x = np.concatenate([noise(rng, lead, 0.05), p1.astype(np.complex64) + noise(rng, 2048, 0.02), noise(rng, 3000, 0.05)])
    level = np.mean(np.abs(x.real)) * np.mean(np.abs(x.imag))
    r = ol.OraP1().execute(x, 0, True, level, want_trace=True)
    assert r["detected"] and r["p1_decoded"] == 1
    assert (r["s1"], r["s2"]) == (s1, s2) and r["preamble"] == s1 and r["fft_mode"] == s2 >> 1
    assert r["shift"] == 86 + shift_carriers
    assert abs(r["consume"] - r["idx_buffer_sym"] - (lead + 2048)) <= 2                   # the arg-max sits on the last P1 sample
    spacing = (64e6 / 7) / 1024
    assert abs(r["coarse_freq_offset"] - (shift_carriers + 0.02) * spacing) < 0.02 * spacing
    assert abs(int(np.nanargmax(r["trace"])) - (lead + 2047)) <= 2


def test_p1_not_decoded_twice_unless_reset():
    rng = np.random.Generator(np.random.PCG64(5))
    p1 = (t2_tx.p1_symbol(0, 10) * 0.3).astype(np.complex64)
    frame = np.concatenate([noise(rng, 3000, 0.02), p1, noise(rng, 2500, 0.02)])
    x = np.concatenate([frame, frame])
    o = ol.OraP1()
    level = np.mean(np.abs(x.real)) * np.mean(np.abs(x.imag))
    a = o.execute(x, 0, True, level)
    b = o.execute(x, a["consume"], False, 0.0)
    assert a["detected"] and b["detected"] and a["shift"] == 86 and b["shift"] == -1 and b["p1_decoded"] == 1
    assert abs(b["consume"] - b["idx_buffer_sym"] - (len(frame) + 3000 + 2048)) <= 2
    c = ol.OraP1()
    c.execute(x, 0, True, level)
    d = c.execute(x, a["consume"], False, 0.0, reset=True)
    assert d["shift"] == 86
