"""The front-end restatement (oracle/front_oracle.c) against the reference's own Qt-free classes compiled unmodified
(oracle/_ref/libref_dsp.so: filter_decimator.h, interpolator_farrow.hh, loop_filters.hh, buffers.hh)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol

needs_ref = pytest.mark.skipif(ol.ref_dsp() is None or ol.ref_dsp(True) is None, reason="oracle/_ref/libref_dsp.so not built here")


def _sig(n, seed, scale=0.2):
    rng = np.random.Generator(np.random.PCG64(seed))
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * scale).astype(np.complex64)


@needs_ref
def test_decimator_bit_exact_vs_reference_class():
    x = _sig(20000, 1)
    a, b, c = ol.OraDecim(), ol.OraDecim(ref=True, strict=True), ol.OraDecim(ref=True)
    pos = 0
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 1000, 4097, 7001):          # odd chunk lengths move the decimation phase
        ya, yb, yc = a(x[pos:pos + n]), b(x[pos:pos + n]), c(x[pos:pos + n])
        assert len(ya) == len(yb) == len(yc)
        assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))      # source order of the AVX2 adds
        np.testing.assert_allclose(ya, yc, rtol=0, atol=3e-7)              # the -Ofast build re-associates them
        pos += n
    del b, c
    # impulse response = the taps, every second one
    imp = np.zeros(200, np.complex64)
    imp[0] = 1
    h = ol.OraDecim()(imp)
    assert abs(h.real.sum() + 0) > 0 and np.count_nonzero(h) == 32


def test_decimator_register_form_equals_the_scalar_statement():
    """ora_decim_execute forms the sums eight lanes at a time as the reference's AVX2 registers do; the scalar restatement of the same
    order (decim_sum) gives the same bits for any chunking."""
    x = _sig(30011, 7)
    a, b = ol.OraDecim(), ol.OraDecim(scalar=True)
    pos = 0
    for n in (1, 2, 3, 63, 64, 65, 1000, 4097, 12001, 12814):
        ya, yb = a(x[pos:pos + n]), b(x[pos:pos + n])
        assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
        pos += n


@pytest.mark.parametrize("n", [32768, 16384, 1024])
def test_fft_restatement_against_the_references_fftw_output(n):
    """oracle/fft_oracle.c against tests/golden/fft_golden.npz (outputs of the FFTW binary the reference ships, on the committed
    input generator) and against a float64 DFT: 2e-5 of the spectrum's rms, the tolerance the HIP kernel is held to."""
    import os
    k = np.arange(n, dtype=np.int64)
    x = (((k * 7919) % 251 - 125) / 64.0 + 1j * (((k * 104729) % 241 - 120) / 64.0)).astype(np.complex64)
    f = ol.OraFft(n)
    y = f(x)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "fft_golden.npz"))
    rel = lambda a, b: np.abs(a - b).max() / np.sqrt((np.abs(b) ** 2).mean())
    if "fft_%d" % n in gold:
        assert rel(y, gold["fft_%d" % n]) < 2e-5
    assert rel(y, np.fft.fftshift(np.fft.fft(x.astype(np.complex128)))) < 2e-5
    assert np.array_equal(f(x, shift=False), np.fft.ifftshift(y))      # the shift is the swap of the two halves


@needs_ref
@pytest.mark.parametrize("resample", [0.5, 0.5 - 3 * 8.0e-9, 0.5 + 5 * 8.0e-9, 0.5 * (1 + 1.0e-4), 0.4571, 0.73, 1.0])
def test_farrow_counts_exact_values_close(resample):
    x = _sig(30000, 2)
    a, b, c = ol.OraFarrow(), ol.OraFarrow(ref=True, strict=True), ol.OraFarrow(ref=True)
    pos = 0
    for n in (1, 2, 3, 500, 8191, 20000):
        ya, yb, yc = a(x[pos:pos + n], resample), b(x[pos:pos + n], resample), c(x[pos:pos + n], resample)
        assert len(ya) == len(yb) == len(yc)                                # output count = phase sequence is exact
        assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))       # source order
        np.testing.assert_allclose(ya, yc, rtol=0, atol=2e-6)               # -Ofast association only
        pos += n


@needs_ref
def test_exponential_averager_and_pi_filters_bit_exact():
    r, o = ol.ref_dsp(), ol.oracle()
    x = (_sig(50000, 3).real + 0.01).astype(np.float32)
    r.ref_avg_new.restype = ctypes.c_void_p
    r.ref_avg_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    h = r.ref_avg_new()
    yr = np.zeros_like(x)
    r.ref_avg_run(h, len(x), x.ctypes.data, yr.ctypes.data)
    o.ora_exp_avg.restype = ctypes.c_float
    o.ora_exp_avg.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
    st = ctypes.c_float(0.0)
    yo = np.array([o.ora_exp_avg(ctypes.byref(st), 1.0e-6, float(v)) for v in x[:5000]], np.float32)
    assert np.array_equal(yo.view(np.uint32), yr[:5000].view(np.uint32))

    r.ref_pi_new.restype = ctypes.c_void_p
    r.ref_pi_new.argtypes = [ctypes.c_int]
    r.ref_pi_step.restype = ctypes.c_float
    r.ref_pi_step.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
    o.ora_pi_init.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_int]
    o.ora_pi_step.restype = ctypes.c_float
    o.ora_pi_step.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
    fs = int(np.float32(1.0) / (np.float32(1.0e-6) * np.float32(7.0) / np.float32(64.0)))
    rng = np.random.Generator(np.random.PCG64(4))
    for which, damping, bw, lim in ((0, 0.3, 1000000, 6.2831855), (1, 0.7, 4000000, 1.0 / 32768)):
        hp = r.ref_pi_new(which)
        s = ol.OraPi()
        o.ora_pi_init(ctypes.byref(s), damping, bw, fs)
        for e in (rng.standard_normal(2000) * lim * 0.3).astype(np.float32):
            a = r.ref_pi_step(which, hp, float(e), lim)
            b = o.ora_pi_step(ctypes.byref(s), float(e), lim)
            assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32)


def test_front_chunk_properties():
    """Unpinned part: sanity of the restated front loop (dc removal, IQ-imbalance statistics, NCO rotation)."""
    rng = np.random.Generator(np.random.PCG64(5))
    n = 40000
    iq = (rng.standard_normal((2, n)) * 2000).astype(np.int16)
    f = ol.OraFront(0)
    out, theta = f.execute(iq[0], iq[1], [15000, 25000], [0.0, 0.0], [0.0, 0.0])
    x = (iq[0].astype(np.float32) + 1j * iq[1].astype(np.float32)) / 16384
    assert np.abs(out - x).max() < 1e-3                                       # no rotation, c1 = 0, c2 = 1
    st = f.state()
    assert abs(st["c1"]) < 0.05 and abs(st["c2"] - 1) < 0.05
    out2, _ = f.execute(iq[0], iq[1], [n], [0.0], [1.0e-3])                   # constant CFO correction: |out| unchanged
    assert np.allclose(np.abs(out2), np.abs(out2 * 0 + out2), atol=0)
    ph = np.unwrap(np.angle(out2[100:2000] / (x[100:2000] * st["c2"] + 1e-9)))
    assert abs(np.polyfit(np.arange(len(ph)), ph, 1)[0] + 1.0e-3) < 2e-4       # rotates by -frequency_est_filtered per sample


def test_restatement_against_committed_reference_vectors():
    """The same pinning without oracle/_ref: tests/golden/dsp_golden.npz holds what the reference's own classes produced
    (tests/golden/make_dsp_golden.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dsp_golden.npz"))
    x = g["x"]
    d, pos, parts = ol.OraDecim(), 0, []
    for n in g["decim_lens"]:
        parts.append(d(x[pos:pos + int(n)]))
        pos += int(n)
    got = np.concatenate(parts)
    assert np.array_equal(got.view(np.uint32), g["decim_strict"].view(np.uint32))
    np.testing.assert_allclose(got, g["decim_fast"], rtol=0, atol=3e-7)
    for k, r in enumerate(g["farrow_ratios"]):
        f = ol.OraFarrow()
        y = np.concatenate([f(x[:3000], float(r)), f(x[3000:], float(r))])
        assert len(y) == len(g["farrow%d_strict" % k]) == len(g["farrow%d_fast" % k])
        assert np.array_equal(y.view(np.uint32), g["farrow%d_strict" % k].view(np.uint32))
        np.testing.assert_allclose(y, g["farrow%d_fast" % k], rtol=0, atol=2e-6)
    o = ol.oracle()
    o.ora_exp_avg.restype = ctypes.c_float
    o.ora_exp_avg.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
    st = ctypes.c_float(0.0)
    avg = np.array([o.ora_exp_avg(ctypes.byref(st), 1.0e-6, float(v)) for v in g["avg_in"][:3000]], np.float32)
    assert np.array_equal(avg.view(np.uint32), g["avg_out"][:3000].view(np.uint32))
    o.ora_pi_init.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_int]
    o.ora_pi_step.restype = ctypes.c_float
    o.ora_pi_step.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
    fs = int(np.float32(1.0) / (np.float32(1.0e-6) * np.float32(7.0) / np.float32(64.0)))
    for which, damping, bw in ((0, 0.3, 1000000), (1, 0.7, 4000000)):
        s = ol.OraPi()
        o.ora_pi_init(ctypes.byref(s), damping, bw, fs)
        lim = float(g["pi%d_lim" % which])
        got = np.array([o.ora_pi_step(ctypes.byref(s), float(e), lim) for e in g["pi%d_in" % which]], np.float32)
        assert np.array_equal(got.view(np.uint32), g["pi%d_out" % which].view(np.uint32))
    for ln in (482, 542):
        b = np.zeros_like(x)
        o.ora_sum_run(ln, len(x), x.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(b.view(np.uint32), g["sum%d" % ln].view(np.uint32))
