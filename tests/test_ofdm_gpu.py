"""GPU tier: OFDM-side stages through the C ABI.

FFT: the reference delegates to FFTW3f (binary dependency, not source in the reference tree). Pinned by
tests/golden/fft_golden.npz, produced with the FFTW library the reference ships; FFTW3f itself deviates from the exact
DFT by 4.5e-6 .. 6.6e-6 of the output rms (printed by make_fft_golden.py), so the tolerance against it and against a
float64 DFT is max |error| <= 2e-5 * rms.
Equaliser: restated float operation by float operation; the equalised cells must equal the oracle's bit for bit."""
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "fft_golden.npz"))


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available()
    return torch


def golden_input(n):
    k = np.arange(n, dtype=np.int64)
    return (((k * 7919) % 251 - 125) / 64.0 + 1j * (((k * 104729) % 241 - 120) / 64.0)).astype(np.complex64)


def rel_err(y, ref):
    return np.abs(y - ref).max() / np.sqrt((np.abs(ref) ** 2).mean())


@pytest.mark.parametrize("fft_mode,n", [(5, 32768), (4, 16384)])
def test_fft_against_reference_fftw_and_dft(torch_cuda, fft_mode, n):
    import sdr_receiver_dvb_t2_amd as pkg
    ctx = pkg.t2_ofdm(fft_mode, 1, 6, 4, 0, 59, max_symbols=8)
    x = golden_input(n)
    y = ctx.fft(x)[0]                                                        # host-buffer entry point
    assert rel_err(y, GOLD["fft_%d" % n]) < 2e-5
    rng = np.random.Generator(np.random.PCG64(n))
    xb = (rng.standard_normal((5, n)) + 1j * rng.standard_normal((5, n))).astype(np.complex64)
    xb[3] = 0; xb[3, 1] = 1.0                                                # impulse at n = 1: a pure phase ramp
    yb = ctx.fft_dev(torch_cuda.from_numpy(xb.view(np.float32).reshape(5, n, 2)).cuda()).cpu().numpy()
    yb = yb[..., 0] + 1j * yb[..., 1]
    ref = np.fft.fftshift(np.fft.fft(xb.astype(np.complex128), axis=1), axes=1)
    for b in range(5):
        assert rel_err(yb[b], ref[b]) < 2e-5
    ctx.close()


@pytest.mark.parametrize("fft_mode,n", [(5, 32768), (4, 16384)])
def test_fft_of_one_symbol_on_many_cus_is_the_batch_kernels_output(torch_cuda, fft_mode, n):
    """Calls of one or two symbols (the slot-shaped path) run as two launches over many CUs with the first exchange through memory
    (fft_stage_a_kernel / fft_stage_bc_kernel); every value takes the same operations in the same order as in the one-workgroup kernel
    a larger batch uses: the outputs are bit-identical."""
    import sdr_receiver_dvb_t2_amd as pkg
    ctx = pkg.t2_ofdm(fft_mode, 1, 6, 4, 0, 59, max_symbols=8)
    rng = np.random.Generator(np.random.PCG64(n + 1))
    xb = (rng.standard_normal((5, n)) + 1j * rng.standard_normal((5, n))).astype(np.complex64)
    xb[4] = golden_input(n)
    xd = torch_cuda.from_numpy(xb.view(np.float32).reshape(5, n, 2)).cuda()
    batch = ctx.fft_dev(xd).cpu().numpy()                                    # five symbols: one workgroup each
    for b in range(5):
        one = ctx.fft_dev(xd[b:b + 1].contiguous()).cpu().numpy()            # one symbol: the two-launch form
        assert np.array_equal(one[0].view(np.uint32), batch[b].view(np.uint32))
    two = ctx.fft_dev(xd[3:5].contiguous()).cpu().numpy()
    assert np.array_equal(two.view(np.uint32), batch[3:5].view(np.uint32))
    y = two[1, :, 0] + 1j * two[1, :, 1]
    assert rel_err(y, GOLD["fft_%d" % n]) < 2e-5
    ctx.close()


def test_fft_batch_properties_full_size(torch_cuda):
    """512 symbols of 32K (the batch size of the bench): Parseval and linearity hold for every symbol."""
    import sdr_receiver_dvb_t2_amd as pkg
    torch = torch_cuda
    n, nb = 32768, 512
    ctx = pkg.t2_ofdm(5, 1, 6, 4, 0, 59, max_symbols=8)
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randn((nb, n, 2), device="cuda", generator=g)
    b = torch.randn((nb, n, 2), device="cuda", generator=g)
    fa, fb, fab = ctx.fft_dev(a), ctx.fft_dev(b), ctx.fft_dev(a + 2.0 * b)
    ea = (a.double() ** 2).sum(dim=(1, 2)) * n
    assert torch.allclose((fa.double() ** 2).sum(dim=(1, 2)), ea, rtol=1e-5)
    lin = (fab - (fa + 2.0 * fb)).abs().amax() / fab.abs().amax()
    assert float(lin) < 2e-5
    ctx.close()


@pytest.fixture(params=["default", "3", "5"])
def eq_form(request, monkeypatch):
    """Equaliser kernel form, read when the context is created: default = output ranges in LDS (eq_split_kernel, fewest ranges that
    fit), "3" / "5" = that kernel with three / five ranges per symbol (the segment-group kernel of round 2 was retired in round 6)."""
    if request.param == "default":
        monkeypatch.delenv("T2GPU_EQ_SPLITS", raising=False)
    else:
        monkeypatch.setenv("T2GPU_EQ_SPLITS", request.param)
    return request.param


def make_symbol(m, idx_symbol, seed, snr_db=25.0):
    """One received data symbol after the FFT (fft-shifted): pilots per the carrier map, random unit-power data cells, a
    smooth two-path channel with a common phase, AWGN."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mp, rf = ol.ora_symbol_carriers(m, idx_symbol)
    k = np.arange(m.k_total)
    tx = np.where(mp == 1, (rng.standard_normal(m.k_total) + 1j * rng.standard_normal(m.k_total)) / np.sqrt(2), rf.astype(np.complex128))
    tx[mp == 4] = 0
    chan = (1.0 + 0.35 * np.exp(-2j * np.pi * k * 37 / m.fft_size)) * np.exp(1j * (0.3 + seed * 0.7)) * (0.8 + 0.05 * seed)
    sigma = np.sqrt(0.5 * 10 ** (-snr_db / 10))
    rx = tx * chan + sigma * (rng.standard_normal(m.k_total) + 1j * rng.standard_normal(m.k_total))
    full = np.zeros(m.fft_size, np.complex64)
    full[m.l_nulls:m.l_nulls + m.k_total] = rx.astype(np.complex64)
    return full


@pytest.mark.parametrize("mode", [(5, 1, 6, 4, 0, 59), (5, 0, 3, 0, 0, 20), (5, 1, 1, 2, 2, 12), (4, 1, 6, 4, 0, 30), (4, 0, 0, 3, 0, 17),
                                  (4, 1, 7, 1, 2, 40)])
def test_equaliser_matches_oracle(torch_cuda, mode, eq_form):
    import sdr_receiver_dvb_t2_amd as pkg
    torch = torch_cuda
    m = ol.ora_mode(*mode)
    ctx = pkg.t2_ofdm(*mode, max_symbols=16)
    rows = m.n_data - m.l_fc
    idxs = [1, 2, 3, 4, 5, 1 + (rows - 1), 1 + rows // 2, 2]
    syms = np.stack([make_symbol(m, i, seed=s) for s, i in enumerate(idxs)])
    cells, sync = ctx.eq_data_dev(torch.from_numpy(syms.view(np.float32).reshape(len(idxs), m.fft_size, 2)).cuda(),
                                  torch.tensor(idxs, dtype=torch.int32, device="cuda"))
    cells = cells.cpu().numpy(); sync = sync.cpu().numpy()
    for b, i in enumerate(idxs):
        want, pho, sro = ol.ora_data_symbol(m, i, syms[b])
        got = cells[b, :, 0] + 1j * cells[b, :, 1]
        assert np.array_equal(got.astype(np.complex64), want), (i, np.abs(got - want).max())
        assert sync[b, 0] == np.float32(pho) and sync[b, 1] == np.float32(sro)
    # reference-shaped host call
    got, sro, pho = ctx.eq_data(idxs[0], syms[0])
    want, wpho, wsro = ol.ora_data_symbol(m, idxs[0], syms[0])
    assert np.array_equal(got, want) and np.float32(sro) == np.float32(wsro) and np.float32(pho) == np.float32(wpho)
    ctx.close()


@pytest.mark.parametrize("mode", [(5, 1, 6, 4, 0, 59), (4, 1, 6, 4, 0, 30), (5, 0, 1, 2, 0, 12)])
def test_p2_equaliser_matches_oracle(torch_cuda, mode, eq_form):
    import sdr_receiver_dvb_t2_amd as pkg
    torch = torch_cuda
    m = ol.ora_mode(*mode)
    ctx = pkg.t2_ofdm(*mode, max_symbols=8)
    syms = np.stack([make_symbol(m, 0, seed=s) for s in range(3)])
    cells, sync = ctx.eq_p2_dev(torch.from_numpy(syms.view(np.float32).reshape(3, m.fft_size, 2)).cuda())
    cells = cells.cpu().numpy(); sync = sync.cpu().numpy()
    # P2 is read with the extended-carrier tables whatever the mode (p2_symbol::init -> dvbt2_p2_parameters_init,
    # dvbt2_definition.cpp:88-90; pinned by tests/golden/t2sym_golden.npz "n32k_pp4")
    mp2 = ol.ora_mode(mode[0], 1, *mode[2:])
    for b in range(3):
        want, pho, sro = ol.ora_data_symbol(mp2, 0, syms[b])
        got = (cells[b, :, 0] + 1j * cells[b, :, 1]).astype(np.complex64)
        assert np.array_equal(got, want)
        assert sync[b, 0] == np.float32(pho) and sync[b, 1] == np.float32(sro)
    ctx.close()


@pytest.mark.parametrize("mode", [(5, 1, 3, 2, 0, 20), (4, 1, 1, 3, 0, 9), (4, 0, 4, 1, 2, 45)])
def test_frame_closing_equaliser_matches_oracle(torch_cuda, mode, eq_form):
    import sdr_receiver_dvb_t2_amd as pkg
    torch = torch_cuda
    m = ol.ora_mode(*mode)
    assert m.l_fc == 1
    ctx = pkg.t2_ofdm(*mode, max_symbols=8)
    idx = m.len_frame - 1
    syms = np.stack([make_symbol(m, idx, seed=s) for s in range(3)])
    cells, sync = ctx.eq_fc_dev(torch.from_numpy(syms.view(np.float32).reshape(3, m.fft_size, 2)).cuda())
    cells = cells.cpu().numpy(); sync = sync.cpu().numpy()
    for b in range(3):
        want, pho, sro = ol.ora_data_symbol(m, idx, syms[b])
        got = (cells[b, :, 0] + 1j * cells[b, :, 1]).astype(np.complex64)
        assert np.array_equal(got, want)
        assert sync[b, 0] == np.float32(pho) and sync[b, 1] == np.float32(sro)
    ctx.close()


@pytest.mark.parametrize("mode", [(5, 1, 6, 4, 0, 59), (4, 1, 1, 3, 0, 9), (4, 0, 4, 1, 2, 45), (5, 0, 3, 0, 0, 20)])
def test_sync_floats_from_the_pilots_alone_equal_the_equalisers(torch_cuda, mode):
    """t2gpu_sym_sync_dev (what the slot-shaped path waits for per symbol) against the equaliser launches' d_sync and
    t2gpu_cp_correlate_dev's d_out4: the same floats bit for bit, data / P2 / frame-closing tables, and the copy the kernel itself
    stores into page-locked memory with the sequence word behind it."""
    import sdr_receiver_dvb_t2_amd as pkg
    from sdr_receiver_dvb_t2_amd import front
    torch = torch_cuda
    m = ol.ora_mode(*mode)
    ctx = pkg.t2_ofdm(*mode, max_symbols=4)
    rows = m.n_data - m.l_fc
    guard = m.fft_size // 128 if mode[0] == 5 else m.fft_size // 16
    rng = np.random.Generator(np.random.PCG64(11))
    cases = [(0, 1), (0, 2), (0, rows), (0, 1 + rows // 2), (1, 0)] + ([(2, m.len_frame - 1)] if m.l_fc else [])
    h_small = torch.zeros(8, dtype=torch.float32).pin_memory()
    h_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
    for n, (kind, idx) in enumerate(cases):
        sym = make_symbol(m, idx, seed=n)
        spec = torch.from_numpy(sym.view(np.float32).reshape(1, m.fft_size, 2)).cuda()
        if kind == 0:
            _, want = ctx.eq_data_dev(spec, torch.tensor([idx], dtype=torch.int32, device="cuda"))
        elif kind == 1:
            _, want = ctx.eq_p2_dev(spec)
        else:
            _, want = ctx.eq_fc_dev(spec)
        buffered = (rng.standard_normal((1, guard + m.fft_size, 2)) * 0.25).astype(np.float32)
        buffered[0, m.fft_size:, :] = buffered[0, :guard, :] * np.float32(0.9) + np.float32(0.01)
        buf = torch.from_numpy(buffered).cuda()
        want_cp = front.cp_correlate_dev(torch.view_as_complex(buf), m.fft_size, guard)
        cp4, sync = ctx.sym_sync_dev(kind, idx, spec[0], buf[0], guard, host=(h_small, h_flag, n + 1))
        torch.cuda.synchronize()
        assert np.array_equal(sync.cpu().numpy().view(np.uint32), want.cpu().numpy()[0].view(np.uint32)), (kind, idx)
        assert np.array_equal(cp4.cpu().numpy().view(np.uint32), want_cp.cpu().numpy()[0].view(np.uint32)), (kind, idx)
        assert int(h_flag[0]) == n + 1
        assert np.array_equal(h_small.numpy()[:4].view(np.uint32), cp4.cpu().numpy().view(np.uint32))
        assert np.array_equal(h_small.numpy()[4:6].view(np.uint32), sync.cpu().numpy().view(np.uint32))
        # without a buffered symbol: the floats alone
        _, sync2 = ctx.sym_sync_dev(kind, idx, spec[0])
        assert np.array_equal(sync2.cpu().numpy().view(np.uint32), sync.cpu().numpy().view(np.uint32))
    ctx.close()


@pytest.mark.parametrize("one_launch", [1, 0])
@pytest.mark.parametrize("mode", [(5, 1, 6, 4, 0, 59), (4, 1, 1, 3, 0, 9), (4, 0, 4, 1, 2, 45), (4, 0, 0, 3, 0, 17), (5, 1, 1, 2, 2, 12)])
def test_fft_with_the_sync_floats_in_its_last_launch(torch_cuda, mode, one_launch):
    """t2gpu_fft_sym_sync_dev (what t2gpu_demod_execute launches per symbol) against t2gpu_fft_execute_dev + t2gpu_sym_sync_dev: spectrum,
    guard correlation and synchronisation floats bit for bit -- 32K and 16K (whose 128-lane launch stands for the correlation's 256
    lanes), data / frame-closing tables inside the launch, P2 and a dense pilot pattern (PP1) through the separate launches inside. Both forms:
    the whole transform in ONE launch (default: stage A's workgroups in front of the others, which wait for them) and the two launches."""
    import sdr_receiver_dvb_t2_amd as pkg
    torch = torch_cuda
    m = ol.ora_mode(*mode)
    ctx = pkg.t2_ofdm(*mode, max_symbols=2)
    assert pkg.lib().t2gpu_ofdm_set_one_launch(ctx._h, one_launch) == 0        # a property of the handle
    rows = m.n_data - m.l_fc
    guard = m.fft_size // 128 if mode[0] == 5 else m.fft_size // 16
    rng = np.random.Generator(np.random.PCG64(23))
    cases = [(0, 1), (0, rows), (0, 2), (1, 0)] + ([(2, m.len_frame - 1)] if m.l_fc else [])
    h_small = torch.zeros(8, dtype=torch.float32).pin_memory()
    h_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
    for n, (kind, idx) in enumerate(cases * 2):                     # twice: the launch's counter is back at zero
        buffered = (rng.standard_normal((guard + m.fft_size, 2)) * 0.25).astype(np.float32)
        buffered[:guard] = buffered[m.fft_size:] * np.float32(0.95) + np.float32(0.003)
        buf = torch.from_numpy(buffered).cuda()
        want_spec = ctx.fft_dev(buf[guard:].contiguous().reshape(1, m.fft_size, 2))[0]
        want_cp, want_sync = ctx.sym_sync_dev(kind, idx, want_spec, buf, guard)
        with_cp = n % 3 != 2
        spec, cp4, sync = ctx.fft_sym_sync_dev(kind, idx, buf, guard, with_cp=with_cp, host=(h_small, h_flag, 100 + n))
        torch.cuda.synchronize()
        assert torch.equal(spec.view(torch.int32), want_spec.view(torch.int32)), (kind, idx)
        assert torch.equal(sync.view(torch.int32), want_sync.view(torch.int32)), (kind, idx)
        if with_cp:
            assert torch.equal(cp4.view(torch.int32), want_cp.view(torch.int32)), (kind, idx)
            assert np.array_equal(h_small.numpy()[:4].view(np.uint32), cp4.cpu().numpy().view(np.uint32))
        assert int(h_flag[0]) == 100 + n
        assert np.array_equal(h_small.numpy()[4:6].view(np.uint32), sync.cpu().numpy().view(np.uint32))
    ctx.close()
