"""GPU tier: sample-rate front end through the C ABI against the oracle (oracle/front_oracle.c, pinned to the reference's own
Qt-free classes -- see tests/test_oracle_front.py).

Parity bars
  * decimator, Farrow resampler: BIT-EXACT (output counts, positions and values) for any chunking and resample ratio.
  * full front end (dc removal -> IQ-imbalance -> NCO -> Farrow -> decimator): output COUNTS per chunk exact; values within
    2e-6 absolute (signals are O(0.1..1)). The only inexact stage is the dc averager, a float IIR whose rounding noise no
    parallel order reproduces (the device evaluates it in double); fed with the device's own de-rotated samples, the oracle's
    Farrow + decimator reproduce the device output bit for bit.
  * sign statistics -> c2, level_detect: relative 1e-4; c1: absolute 1e-4 (see the comment at the assertion) (the reference adds 1e5..1e6 floats sequentially); since c1/c2
    scale the whole next call, the checker is handed the device's c1/c2 after each call so values stay comparable at 2e-6."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available()
    return torch


def sig(n, seed, scale=0.2):
    rng = np.random.Generator(np.random.PCG64(seed))
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * scale).astype(np.complex64)


def iq16(n, seed, dc=(90, -40), imbalance=0.03, rms=1500):
    rng = np.random.Generator(np.random.PCG64(seed))
    i = rng.standard_normal(n) * rms
    q = rng.standard_normal(n) * rms * (1 + imbalance) + 0.02 * i
    return (np.clip(i + dc[0], -32768, 32767).astype(np.int16), np.clip(q + dc[1], -32768, 32767).astype(np.int16))


def bits(x):
    return np.ascontiguousarray(x).view(np.uint32)


def test_decimator_bit_exact(torch_cuda):
    from sdr_receiver_dvb_t2_amd import front
    f, o = front.front_end(max_samples=1 << 16), ol.OraDecim()
    x = sig(140000, 11)
    pos = 0
    for n in (1, 1, 2, 3, 62, 63, 64, 65, 511, 512, 513, 4097, 65536, 60001):
        ya, yb = f.decimate(x[pos:pos + n]), o(x[pos:pos + n])
        assert len(ya) == len(yb)
        assert np.array_equal(bits(ya), bits(yb)), n
        pos += n


@pytest.mark.parametrize("resample", [0.5, 0.5 - 3 * 8.0e-9, 0.5 + 5 * 8.0e-9, 0.5 * (1 + 1.0e-4), 0.546875, 0.4571, 1.0])
def test_farrow_bit_exact(torch_cuda, resample):
    from sdr_receiver_dvb_t2_amd import front
    f = front.front_end(sample_rate=front.SAMPLE_RATE * 2 * min(resample, 0.5), max_samples=1 << 16)   # sizes the buffers
    o = ol.OraFarrow()
    x = sig(100000, 12)
    pos = 0
    for n in (1, 2, 3, 4, 255, 256, 257, 1000, 30000, 65536):
        ya, yb = f.farrow(x[pos:pos + n], resample), o(x[pos:pos + n], resample)
        assert len(ya) == len(yb), n
        assert np.array_equal(bits(ya), bits(yb)), n
        pos += n


def oracle_chain(id_device, i_in, q_in, chunk_len, pe, fe, rs):
    """The reference's execute(): per chunk front loop -> Farrow -> decimator (dvbt2_demodulator.cpp:151-226)."""
    fo, fa, de = ol.OraFront(id_device), ol.OraFarrow(), ol.OraDecim()
    return fo, fa, de


def run_oracle(objs, i_in, q_in, chunk_len, pe, fe, rs):
    fo, fa, de = objs
    derot, theta = fo.execute(i_in, q_in, chunk_len, pe, fe)
    outs, lens, interp = [], [], []
    pos = 0
    for n, r in zip(chunk_len, rs):
        y = fa(derot[pos:pos + n], r)
        interp.append(y)
        z = de(y)
        outs.append(z)
        lens.append(len(z))
        pos += n
    return derot, np.concatenate(interp), np.concatenate(outs), np.array(lens, np.int32)


@pytest.mark.parametrize("id_device,case", [(0, "nominal"), (0, "tracking"), (1, "tracking"), (2, "cfo")])
def test_front_end_matches_oracle(torch_cuda, id_device, case):
    from sdr_receiver_dvb_t2_amd import front
    n_calls, chunks = 3, [33024, 33024, 17000, 1, 40000, 2047]
    n = sum(chunks)
    f = front.front_end(id_device=id_device, max_samples=n)
    objs = oracle_chain(id_device, None, None, None, None, None, None)
    rng = np.random.Generator(np.random.PCG64(31 + id_device))
    for call in range(n_calls):
        i_in, q_in = iq16(n * f.stride, 100 + call)
        if case == "nominal":
            pe, fe, rs = np.zeros(len(chunks)), np.zeros(len(chunks)), np.full(len(chunks), f.resample)
        elif case == "tracking":
            pe = rng.standard_normal(len(chunks)) * 0.02
            fe = rng.standard_normal(len(chunks)) * 2e-5
            rs = f.resample - np.cumsum(rng.integers(-1, 2, len(chunks))) * 8.0e-9
        else:
            pe = rng.standard_normal(len(chunks)) * 0.3
            fe = np.full(len(chunks), 0.0123) + rng.standard_normal(len(chunks)) * 1e-3
            rs = np.full(len(chunks), f.resample * (1 + 5e-5))
        pe, fe = pe.astype(np.float32), fe.astype(np.float32)
        derot, interp, want, want_len = run_oracle(objs, i_in, q_in, chunks, pe, fe, rs)
        got, got_len = f.execute(i_in, q_in, chunks, pe, fe, rs)
        assert np.array_equal(got_len, want_len)
        assert len(got) == len(want)
        g_derot, g_interp = f.debug_stream(0, n), f.debug_stream(1, len(interp) + 8)
        assert len(g_interp) == len(interp)
        np.testing.assert_allclose(g_derot, derot, rtol=0, atol=2e-6)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
        so, sg = objs[0].state(), f.state()
        assert bits(np.float32(sg["phase_nco"])) == bits(np.float32(so["phase_nco"]))
        assert bits(np.float32(sg["frequency_nco"])) == bits(np.float32(so["frequency_nco"]))
        assert bits(np.float32(sg["x1"])) == bits(objs[1].phase())
        for k in ("c2", "level_detect"):
            assert abs(sg[k] - so[k]) <= 1e-4 * max(abs(so[k]), 1e-3), (k, sg[k], so[k])
        # c1 = theta1 / theta2 with theta1 -= imag * sign(real): whenever the dc estimate crosses one of the int16 levels, the
        # samples sitting on that level have real = +-1e-9 and their sign -- a full +-imag in theta1 -- follows the last bit of
        # the averager, which differs between the reference's float IIR and any other evaluation. ~1e-5 of c1 per such sample.
        assert abs(sg["c1"] - so["c1"]) <= 1e-4, (sg["c1"], so["c1"])
        # c1/c2 multiply every sample of the NEXT call: hand the checker the device's values (they agree to 1e-4, the accuracy
        # of the reference's sequential float sums) so the next call is again compared at 2e-6
        objs[0].set_iq(sg["c1"], sg["c2"])
        assert abs(sg["dc_re"] - so["dc_re"]) < 1e-6 and abs(sg["dc_im"] - so["dc_im"]) < 1e-6
    # downstream stages are exact: the oracle's Farrow + decimator on the DEVICE's de-rotated samples give the device output
    f2 = front.front_end(id_device=id_device, max_samples=n)
    i_in, q_in = iq16(n * f2.stride, 500)
    rs = np.full(len(chunks), f2.resample - 16.0e-9)
    got, got_len = f2.execute(i_in, q_in, chunks, None, np.full(len(chunks), 3e-4, np.float32), rs)
    g_derot = f2.debug_stream(0, n)
    fa, de = ol.OraFarrow(), ol.OraDecim()
    pos, outs = 0, []
    for c, r in zip(chunks, rs):
        outs.append(de(fa(g_derot[pos:pos + c], r)))
        pos += c
    assert np.array_equal(bits(got), bits(np.concatenate(outs)))


def test_short_calls_in_one_launch_equal_the_five_launches(torch_cuda, monkeypatch):
    """A call of up to a few OFDM symbols' worth of samples runs as ONE launch with one pass per workgroup (front_one_kernel: 1024
    samples per workgroup, the dc prefix from the workgroups before it through flags, de-rotation / Farrow / decimator windows in LDS, run
    tables in the kernel arguments); t2gpu_front_set_chain(h, 0) keeps the five launches. NCO phase, Farrow position, output counts and
    decimation phase are the same bit for bit; the cells agree to what the two association orders of the dc averager's double-precision
    scan leave (a float rounding now and then: 1e-6 here, both are within 2e-6 of the reference's float IIR) -- over calls of one symbol,
    of a few samples (what the slot-shaped path hands over when the chunk estimate was a sample short), of a P1-sized chunk, of several
    chunks, and of lengths that do not qualify; the carried state goes from either form into the other."""
    from sdr_receiver_dvb_t2_amd import front
    n_max = 1 << 19
    five = front.front_end(max_samples=n_max)
    assert five._l.t2gpu_front_set_chain(five.h, 0) == 0
    one = front.front_end(max_samples=n_max)
    rng = np.random.Generator(np.random.PCG64(77))
    calls = [[70001], [3], [66050], [1], [2047], [4096], [4097], [90000], [300000], [5], [33024], [35072], [1024], [1025], [2, 33022],
             [98304], [98305], [20000, 13024, 2]]
    for call, chunks in enumerate(calls):
        n = sum(chunks)
        i_in, q_in = iq16(n, 900 + call)
        pe = (rng.standard_normal(len(chunks)) * 0.05).astype(np.float32)
        fe = (rng.standard_normal(len(chunks)) * (3e-4 if call % 3 else 2e-6)).astype(np.float32)
        rs = five.resample - rng.integers(-3, 4, len(chunks)) * 8.0e-9
        a, al = five.execute(i_in, q_in, chunks, pe, fe, rs)
        b, bl = one.execute(i_in, q_in, chunks, pe, fe, rs)
        assert np.array_equal(al, bl) and len(a) == len(b), (call, chunks)
        np.testing.assert_allclose(b, a, rtol=0, atol=1e-6, err_msg=str((call, chunks)))
        sa, sb = five.state(), one.state()
        for k in ("phase_nco", "frequency_nco", "x1"):
            assert bits(np.float32(sa[k])) == bits(np.float32(sb[k])), (call, chunks, k)
        for k in ("dc_re", "dc_im"):
            assert abs(sa[k] - sb[k]) <= 1e-7, (call, chunks, k, sa[k], sb[k])
        for k in ("c1", "c2", "level_detect"):     # (a sample ON an int16 level takes its sign from the averager's last bit: ~1e-5 of c1 apiece)
            assert abs(sa[k] - sb[k]) <= 1e-4 * max(abs(sa[k]), 1e-3) + 1e-4 * (k == "c1"), (call, chunks, k, sa[k], sb[k])
    # the delay lines were carried alike: one more call, through the five launches on BOTH objects
    assert one._l.t2gpu_front_set_chain(one.h, 0) == 0
    i_in, q_in = iq16(5000, 999)
    a, al = five.execute(i_in, q_in, [5000], [np.float32(0.01)], [np.float32(1e-4)], [five.resample])
    b, bl = one.execute(i_in, q_in, [5000], [np.float32(0.01)], [np.float32(1e-4)], [five.resample])
    assert np.array_equal(al, bl)
    np.testing.assert_allclose(b, a, rtol=0, atol=1e-6)
    five.close(); one.close()


def test_one_launch_front_end_matches_oracle(torch_cuda):
    """The one-launch form against the oracle's restatement of the reference's sample loop, Farrow and decimator, call after call with the
    state carried (symbol-sized chunks with moving loop values, as t2gpu_demod_execute issues them): cells within 2e-6, NCO phase and
    Farrow position bit-equal."""
    from sdr_receiver_dvb_t2_amd import front
    f = front.front_end(id_device=0, max_samples=1 << 17)
    objs = oracle_chain(0, None, None, None, None, None, None)
    rng = np.random.Generator(np.random.PCG64(5))
    for call, n in enumerate([33024, 2, 33022, 35072, 33024, 1, 33023]):
        i_in, q_in = iq16(n, 300 + call)
        pe = np.array([rng.standard_normal() * 0.02], np.float32)
        fe = np.array([rng.standard_normal() * 2e-5], np.float32)
        rs = np.array([f.resample - rng.integers(-2, 3) * 8.0e-9])
        derot, interp, want, want_len = run_oracle(objs, i_in, q_in, [n], pe, fe, rs)
        got, got_len = f.execute(i_in, q_in, [n], pe, fe, rs)
        assert np.array_equal(got_len, want_len) and len(got) == len(want)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
        so, sg = objs[0].state(), f.state()
        assert bits(np.float32(sg["phase_nco"])) == bits(np.float32(so["phase_nco"]))
        assert bits(np.float32(sg["frequency_nco"])) == bits(np.float32(so["frequency_nco"]))
        assert bits(np.float32(sg["x1"])) == bits(objs[1].phase())
        assert abs(sg["dc_re"] - so["dc_re"]) < 1e-6 and abs(sg["dc_im"] - so["dc_im"]) < 1e-6
        objs[0].set_iq(sg["c1"], sg["c2"])
    f.close()


def test_front_end_dev_entry_and_errors(torch_cuda):
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd import front
    from sdr_receiver_dvb_t2_amd._lib import T2GpuError
    n = 200000
    f, g = front.front_end(max_samples=n), front.front_end(max_samples=n)
    i_in, q_in = iq16(n, 7)
    want, wl = f.execute(i_in, q_in, [n])
    out = torch.zeros(n + 64, dtype=torch.complex64, device="cuda")
    cells, gl = g.execute_dev(torch.from_numpy(i_in).cuda(), torch.from_numpy(q_in).cuda(), [n], out)
    torch.cuda.synchronize()
    assert cells == len(want) == n and gl[0] == wl[0]                       # resample 0.5 then /2: one cell per input sample
    assert np.array_equal(bits(out[:cells].cpu().numpy()), bits(want))
    with pytest.raises(T2GpuError):
        f.execute(np.zeros(n + 1, np.int16), np.zeros(n + 1, np.int16), [n + 1])
    with pytest.raises(T2GpuError):
        f.execute(i_in, q_in, [n], arbitrary_resample=[1.0e-5])
    f.reset()
    again, _ = f.execute(i_in, q_in, [n])
    assert np.array_equal(bits(again), bits(want))                           # reset() restores the constructor state


def test_cp_correlation_matches_oracle(torch_cuda):
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd import front
    rng = np.random.Generator(np.random.PCG64(3))
    for fft_size, guard in ((32768, 256), (16384, 4096), (32768, 4864)):
        syms = []
        for s in range(3):
            body = sig(fft_size, 40 + s, 1.0)
            cfo = rng.uniform(-0.4, 0.4) / fft_size
            x = np.concatenate((body[-guard:], body)) * np.exp(2j * np.pi * cfo * np.arange(fft_size + guard))
            syms.append((x + sig(fft_size + guard, 60 + s, 0.05)).astype(np.complex64))
        syms = np.stack(syms)
        out = front.cp_correlate_dev(torch.from_numpy(syms).cuda(), fft_size, guard).cpu().numpy()
        for s in range(3):
            fe, sm = ol.ora_cp_frequency_est(syms[s], fft_size, guard)
            assert abs(out[s, 0] - sm.real) <= 2e-5 * abs(sm) and abs(out[s, 1] - sm.imag) <= 2e-5 * abs(sm)
            assert abs(out[s, 2] - fe) <= 1e-9 + 1e-4 * abs(fe)


def test_decimator_and_farrow_against_reference_vectors(torch_cuda):
    """Straight against what the reference's own filter_decimator / interpolator_farrow classes produced (tests/golden/
    dsp_golden.npz, compiled from the reference sources without fast-math): bit for bit."""
    import os
    from sdr_receiver_dvb_t2_amd import front
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dsp_golden.npz"))
    x = g["x"]
    f = front.front_end(max_samples=1 << 14)
    pos, parts = 0, []
    for n in g["decim_lens"]:
        parts.append(f.decimate(x[pos:pos + int(n)]))
        pos += int(n)
    assert np.array_equal(bits(np.concatenate(parts)), bits(g["decim_strict"]))
    for k, r in enumerate(g["farrow_ratios"]):
        f = front.front_end(sample_rate=front.SAMPLE_RATE * 2 * min(float(r), 0.5), max_samples=1 << 14)
        y = np.concatenate([f.farrow(x[:3000], float(r)), f.farrow(x[3000:], float(r))])
        assert np.array_equal(bits(y), bits(g["farrow%d_strict" % k]))


def test_the_loop_on_the_device_plans_the_nco_as_the_host_does(torch_cuda):
    """t2gpu_front_execute_loop_dev (the chunk's NCO runs planned from the device's loop state, by every workgroup for itself since round 6) against
    t2gpu_front_execute_dev with the same loop values planned on the host: every cell bit for bit, and the device's accumulators where the
    host's are after t2gpu_front_loop_follow -- residual offsets from none to one that takes hundreds of runs per chunk (searched in memory),
    symbol-sized chunks and the few-sample ones that complete a symbol."""
    import torch
    from sdr_receiver_dvb_t2_amd import front
    host = front.front_end(max_samples=1 << 17)
    dev = front.front_end(max_samples=1 << 17)
    rng = np.random.Generator(np.random.PCG64(41))
    fes = [0.0, 2e-5, -3e-4, 7e-4, 0.0123, -6.9e-5, 1e-6]
    for call, n in enumerate([33024, 2, 33022, 35072, 1, 33023, 8000]):
        i_in, q_in = iq16(n, 700 + call)
        pe, fe = np.float32(rng.standard_normal() * 0.05), np.float32(fes[call])
        rs = host.resample - rng.integers(-2, 3) * 8.0e-9
        di, dq = torch.from_numpy(i_in).cuda(), torch.from_numpy(q_in).cuda()
        oa = torch.zeros(n + 64, dtype=torch.complex64, device="cuda")
        ob = torch.zeros(n + 64, dtype=torch.complex64, device="cuda")
        na, _ = host.execute_dev(di, dq, [n], oa, [pe], [fe], [rs])
        st = np.zeros(10, np.float32); st[0] = pe; st[1] = fe                 # phase_est_filtered, frequency_est_filtered (tuner 0)
        dev.loop_begin(st)
        nb = dev.execute_loop_dev(di, dq, n, ob, rs)
        assert nb == na, (call, na, nb)
        dev.loop_follow(pe, fe)
        got = dev.loop_read()
        torch.cuda.synchronize()
        assert torch.equal(oa[:na].view(torch.int32) if False else torch.view_as_real(oa[:na]).view(torch.int32), torch.view_as_real(ob[:nb]).view(torch.int32)), call
        sa, sb = host.state(), dev.state()
        for k in ("phase_nco", "frequency_nco", "x1", "dc_re", "dc_im"):
            assert bits(np.float32(sa[k])) == bits(np.float32(sb[k])), (call, k, sa[k], sb[k])
        assert got["error"] == 0
        assert bits(np.float32(got["phase_nco"])) == bits(np.float32(sb["phase_nco"])), (call, got, sb)
        assert bits(np.float32(got["frequency_nco"])) == bits(np.float32(sb["frequency_nco"])), (call, got, sb)
    host.close(); dev.close()


def test_the_loop_filters_on_the_device_are_the_hosts(torch_cuda):
    """sym_sync_kernel with the device's loop state advances the two PI loop filters as t2gpu_sync_frequency / t2gpu_sync_symbol do: after
    every symbol phase_est_filtered and frequency_est_filtered + tuner (published at h_small[6..7]) equal the host's, bit for bit."""
    import torch
    import sdr_receiver_dvb_t2_amd as pkg
    from sdr_receiver_dvb_t2_amd import front
    mode = (5, 1, 6, 4, 0, 59)
    ctx = pkg.t2_ofdm(*mode, max_symbols=2)
    fe_obj = front.front_end(max_samples=1 << 16)
    loops = front.sync_loops()
    st = loops.export(); st[2] = np.float32(1.25e-5)                          # an emulated tuner offset
    fe_obj.loop_begin(st)
    rng = np.random.Generator(np.random.PCG64(9))
    guard, n = 256, 32768
    h_small = torch.zeros(8, dtype=torch.float32).pin_memory()
    h_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
    for k in range(12):
        spec = torch.from_numpy((rng.standard_normal((n, 2)) * 0.3).astype(np.float32)).cuda()
        buffered = (rng.standard_normal((guard + n, 2)) * 0.25).astype(np.float32)
        buffered[n:] = buffered[:guard] * np.float32(0.9) + np.float32(0.02 * (k - 5))
        buf = torch.from_numpy(buffered).cuda()
        with_cp = k % 4 != 3
        cp4, sync = ctx.sym_sync_dev(0, 1 + k, spec, buf if with_cp else None, guard, host=(h_small, h_flag, k + 1), loop=fe_obj.loop_dev())
        torch.cuda.synchronize()
        cp4, sync = cp4.cpu().numpy(), sync.cpu().numpy()
        if with_cp:
            loops.frequency(float(cp4[2]), n)
        loops.symbol(float(sync[0]), float(sync[1]))
        g = loops.get()
        want_pe, want_fe = np.float32(g["phase_est_filtered"]), np.float32(np.float32(g["frequency_est_filtered"]) + st[2])
        assert bits(h_small.numpy()[6:7])[0] == bits(want_pe), (k, h_small[6], want_pe)
        assert bits(h_small.numpy()[7:8])[0] == bits(want_fe), (k, h_small[7], want_fe)
    ctx.close(); fe_obj.close(); loops.close()
