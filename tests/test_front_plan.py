"""CPU tier: the run-table planner of the front end (csrc/front_plan.cpp, host only) reproduces the reference's two
sequential float accumulators exactly -- NCO phase against a plain float loop of dvbt2_demodulator.cpp:187-193, Farrow
position against the oracle (which is pinned to the reference class) -- and does so in few runs in the operating regime."""
import numpy as np
import pytest

import oracle_lib as ol

PI2 = np.float32(np.float32(3.14159274101257324219) * np.float32(2.0))


def nco_loop(acc, n, fe):
    acc, fe = np.float32(acc), np.float32(fe)
    out = np.zeros(n, np.float32)
    for i in range(n):
        acc = np.float32(acc - fe)
        while acc > PI2:
            acc = np.float32(acc - PI2)
        while acc < -PI2:
            acc = np.float32(acc + PI2)
        out[i] = acc
    return out, acc


@pytest.mark.parametrize("fe", [0.0, 1.0e-3, -1.0e-3, 3.7e-5, -2.2e-6, 0.0123, -0.4, 1.0e-9, 6.0, 2.9802322e-8 * 1.5, 4.76837158203125e-07 * 2.5])
def test_nco_runs_exact(built, fe):
    from sdr_receiver_dvb_t2_amd import front
    rng = np.random.Generator(np.random.PCG64(int(abs(fe) * 1e9) + 1))
    n = 60000
    acc = np.float32(rng.uniform(-6, 6))
    for rep in range(2):
        want, acc_w = nco_loop(acc, n, fe)
        got, acc_g, runs = front.plan_nco(float(acc), n, float(fe))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert np.float32(acc_g).view(np.uint32) == np.float32(acc_w).view(np.uint32)
        if 0 < abs(fe) < 0.02:
            assert runs < n // 20                                           # binade-long runs, not one per sample
        acc = acc_w


def test_nco_stationary_is_one_run(built):
    from sdr_receiver_dvb_t2_amd import front
    for acc, fe in ((0.0, 0.0), (0.3, 0.0), (5.0, 1.0e-9), (-6.0, -1.0e-8)):      # fe = 0 or below half an ulp of the accumulator
        got, acc_g, runs = front.plan_nco(acc, 100000, fe)
        assert runs == 1 and np.all(got == np.float32(acc)) and np.float32(acc_g) == np.float32(acc)


def test_nco_many_random(built):
    from sdr_receiver_dvb_t2_amd import front
    rng = np.random.Generator(np.random.PCG64(77))
    for _ in range(40):
        fe = np.float32(rng.standard_normal() * 10.0 ** rng.uniform(-7, -1))
        acc = np.float32(rng.uniform(-6.2, 6.2))
        n = int(rng.integers(1, 9000))
        want, acc_w = nco_loop(acc, n, fe)
        got, acc_g, _ = front.plan_nco(float(acc), n, float(fe))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (fe, acc, n)


def farrow_positions(x1, n, resample):
    """Per-input outputs and positions from the oracle (state forced through a dummy warm-up is not possible, so restate)."""
    d, x1 = np.float32(resample), np.float32(x1)
    cnt = np.zeros(n, np.int32)
    pos = np.zeros(n, np.float32)
    for i in range(n):
        pos[i] = x1
        c = 0
        while x1 < np.float32(0.5):
            x1 = np.float32(x1 + d)
            c += 1
        x1 = np.float32(x1 - np.float32(1.0))
        cnt[i] = c
    return cnt, pos, x1


@pytest.mark.parametrize("resample", [0.5, 0.5 - 8.0e-9, 0.5 - 3 * 8.0e-9, 0.5 + 8.0e-9, 0.5 + 7 * 8.0e-9, 0.5 * (1 + 1e-4), 0.5 * (1 - 1e-4),
                                      0.4571, 0.546875, 0.73, 1.0, 1.37, 0.25])
def test_farrow_runs_exact(built, resample):
    from sdr_receiver_dvb_t2_amd import front
    n = 50000
    x1 = np.float32(-0.5)
    for rep in range(2):
        wc, wp, wx = farrow_positions(x1, n, resample)
        gc, gp, gx, total, runs = front.plan_farrow(float(x1), n, resample)
        assert np.array_equal(gc, wc)
        assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
        assert total == int(wc.sum()) and np.float32(gx).view(np.uint32) == np.float32(wx).view(np.uint32)
        if abs(resample - 0.5) < 1e-3 or resample in (1.0, 0.25):
            assert runs < 400, runs                                          # operating regime: a handful of runs per chunk
        x1 = wx
    # and the oracle's own count agrees with this restatement (oracle is pinned to the reference class)
    o = ol.OraFarrow()
    y, ph = o(np.ones(3000, np.complex64), resample, want_phases=True)
    wc, wp, _ = farrow_positions(-0.5, 3000, resample)
    assert len(y) == int(wc.sum())
    starts = np.concatenate(([0], np.cumsum(wc)[:-1]))
    assert np.array_equal(ph[starts[wc > 0]].view(np.uint32), wp[wc > 0].view(np.uint32))


def test_farrow_long_drift_crosses_a_slip(built):
    """0.5 - 37 ulp: the position drifts 74 * 2^-25 per input and slips (3 outputs for one input) every ~226 k inputs."""
    from sdr_receiver_dvb_t2_amd import front
    resample = float(np.float32(0.5) - 37 * np.float32(2.0) ** -25)
    n = 700000
    wc, wp, wx = farrow_positions(-0.5, n, resample)
    gc, gp, gx, total, runs = front.plan_farrow(-0.5, n, resample)
    assert np.array_equal(gc, wc) and np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
    assert set(np.unique(wc)) == {2, 3} and runs < 400


def test_farrow_rejects_bad_resample(built):
    from sdr_receiver_dvb_t2_amd import front
    from sdr_receiver_dvb_t2_amd._lib import T2GpuError
    for r in (0.0, -0.5, 1.0e-4, 5.0):
        with pytest.raises(T2GpuError):
            front.plan_farrow(-0.5, 10, r)


def test_sync_loops_match_oracle(built):
    from sdr_receiver_dvb_t2_amd import front
    rng = np.random.Generator(np.random.PCG64(9))
    a, b = front.sync_loops(front.SAMPLE_RATE), ol.OraSync(front.SAMPLE_RATE)
    for k in range(3000):
        if k % 3 == 0:
            fe = float(np.float32(rng.standard_normal() * 1e-5))
            a.frequency(fe, 32768)
            b.frequency(fe, 32768)
        pe, sr = float(np.float32(rng.standard_normal() * 0.05)), float(np.float32(rng.standard_normal() * 0.01))
        a.symbol(pe, sr)
        b.symbol(pe, sr)
    ga, gb = a.get(), b.get()
    assert np.float32(ga["phase_est_filtered"]).view(np.uint32) == np.float32(gb["phase_est_filtered"]).view(np.uint32)
    assert np.float32(ga["frequency_est_filtered"]).view(np.uint32) == np.float32(gb["frequency_est_filtered"]).view(np.uint32)
    assert ga["sample_rate_est_filtered"] == gb["sample_rate_est_filtered"]
    assert ga["arbitrary_resample"] == b.resample()
