"""GPU tier, end to end from the tuner interface: transport-stream packets -> transmitter model (tests/t2_tx.py) -> P1 + cyclic
prefixes + AWGN -> int16 I/Q as an SDR delivers it -> front end (dc / IQ / NCO / Farrow x2 / decimator) -> P1 detection ->
guard-interval correlation -> FFT with guard removal -> equalisers -> de-interleavers -> demapper -> LDPC -> descrambler ->
BBFRAME de-framing -> the same transport-stream bytes. Every stage runs in libt2gpu.so through the C ABI."""
import numpy as np
import pytest

import oracle_lib as ol
import t2_tx
from test_chain_gpu import ts_slice

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("name,mode,lps,mod,fec_type,code_rate,snr,saturate,s2", [
    ("32K normal PP4 GI1/32 64-QAM 64800 r2/3", (5, 0, 3, 0, 0, 20), 400, 2, 1, 2, 22.0, False, 10),
    ("16K ext PP2 GI1/8 frame-closing, 16-QAM 16200 r3/5", (4, 1, 1, 2, 0, 24), 150, 1, 0, 1, 12.0, False, 8),   # higher SNR wraps the int8 LLRs (reference cast)
    ("CFG-A 32K ext PP7 GI1/128 256-QAM 64800 r3/4 (clamped LLRs)", (5, 1, 6, 4, 0, 59), 350, 3, 1, 3, 27.0, True, 10),
])
def test_iq_to_ts(torch_cuda, name, mode, lps, mod, fec_type, code_rate, snr, saturate, s2):
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd.receiver import t2_receiver
    n_frames, seed = 2, 77
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    cpf = (64800 if fec_type else 16200) // (2 * (mod + 1))
    nb = t2_tx.plp_blocks_per_frame(m, lps, cpf)
    k_bch = t2_tx.K_BCH[cid]
    ts = t2_tx.ts_packets(n_frames * nb * (k_bch // 1496 + 1) + 8, seed)
    frames, pos = [], 0
    for f in range(n_frames):
        cells, _, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts_slice(ts, pos, nb, k_bch), nb)
        frames.append(t2_tx.build_frame(m, cells, lps, seed + f, snr_db=None, phase=0.0))
        pos += nb
    rx = t2_receiver((*mode, lps, mod, fec_type, code_rate, 1, nb), dict(saturate_llr=saturate), max_frames=n_frames)
    guard = rx.chain.ofdm.guard_interval_size
    i16, q16, frame_len = t2_tx.iq_stream(frames, guard, s2, snr, seed)
    assert frame_len == rx.frame_len
    out = rx.demod_iq_dev(torch.from_numpy(i16).cuda(), torch.from_numpy(q16).cuda(), n_frames, flush=True)
    torch.cuda.synchronize()
    # P1: both frames found, S1/S2 decoded on the first, P2 starts one P1 length (+ the filter delay) behind the frame start
    assert all(r.detected for r in out["p1"]) and out["p1"][0].s2 == s2 and out["p1"][0].shift == 86
    assert out["p1"][0].fft_mode == (5 if m.fft_size == 32768 else 4)
    delay = out["p2_start"] - (np.arange(n_frames) * frame_len + 2048)
    assert (np.abs(delay - delay[0]) <= 1).all() and 10 <= delay[0] <= 24, delay       # Farrow + 64-tap FIR group delay
    cfo = out["cp"][..., 2].cpu().numpy()
    assert np.abs(cfo).max() < 2e-6                                                    # synchronous source: no frequency offset
    trials = out["trials"].cpu().numpy()
    assert (trials >= 0).all(), trials
    got = rx.chain.ts_from_bits(out["bits"].cpu().numpy(), trials)
    dfl_bytes = (k_bch - 80) // 8
    per_frame = (nb * dfl_bytes) // 187 - 1                       # whole packets one frame's BBFRAMEs carry
    sent = np.concatenate([ts_slice(ts, f * nb, nb, k_bch).reshape(-1) for f in range(n_frames)])
    # frame 1 comes back byte for byte from the first packet on
    assert np.array_equal(got[:per_frame * 188], sent[:per_frame * 188])
    # the transmitter model starts every frame on a packet boundary (SYNCD = 0), so the packet cut by the frame change is lost at
    # the junction; everything else of frame 2 is there too
    n = (len(got) // 188) * 188
    sp = {bytes(p) for p in sent.reshape(-1, 188)}
    matched = sum(bytes(p) in sp for p in got[:n].reshape(-1, 188))
    assert matched >= n_frames * per_frame - 1 and n // 188 - matched <= 2, (matched, n // 188, per_frame)
    # the library's own batch receiver (t2gpu_rx_*: stage sequencing in C++, nothing above the ABI) gives the same buffer the same answer
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    nat = t2_rx(*mode, lps, mod, fec_type, code_rate, 1, nb, max_frames=n_frames, saturate_llr=saturate)
    assert nat.frame_len == frame_len
    count = nat.execute_dev(torch.from_numpy(i16).cuda(), torch.from_numpy(q16).cuda(), n_frames, first_call=True, flush=True)
    assert count == n_frames * nb
    nbits, ntrials = nat.fetch(count)
    info = nat.results(n_frames)
    assert np.array_equal(ntrials, trials) and np.array_equal(nbits, out["bits"].cpu().numpy())
    assert np.array_equal(info["p2_start"], out["p2_start"]) and np.array_equal(info["cp"], out["cp"].cpu().numpy())
    assert [(r.detected, r.s1, r.s2, r.shift) for r in info["p1"]] == [(r.detected, r.s1, r.s2, r.shift) for r in out["p1"]]
    assert info["ldpc_ms"] > 0
    nat.close()
    rx.close()


def test_outer_code_in_the_batch_receiver(torch_cuda):
    """t2gpu_rx_set_outer_code: with real BCH parity on air every decoded FEC frame checks clean and the output is what the
    reference-shaped path gives; with the parity field left zero (which the reference never notices, bch_decoder.cpp:136) the
    same frames are reported as beyond correction and still pass through untouched."""
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    mode, lps, mod, fec_type, code_rate, snr, s2 = (4, 1, 1, 2, 0, 24), 150, 1, 0, 1, 12.0, 8
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    nb = t2_tx.plp_blocks_per_frame(m, lps, 16200 // (2 * (mod + 1)))
    k_bch = t2_tx.K_BCH[cid]
    n_frames = 2
    ts = t2_tx.ts_packets(n_frames * nb * (k_bch // 1496 + 1) + 8, 5)
    outs = {}
    for bch in (True, False):
        frames = []
        for f in range(n_frames):
            cells, _, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts_slice(ts, f * nb, nb, k_bch), nb, bch=bch)
            frames.append(t2_tx.build_frame(m, cells, lps, 9 + f, snr_db=None, phase=0.0))
        rx = t2_rx(*mode, lps, mod, fec_type, code_rate, 1, nb, max_frames=n_frames)
        i16, q16, _ = t2_tx.iq_stream(frames, rx.geometry.guard_interval_size, s2, snr, 11)
        di, dq = torch.from_numpy(i16).cuda(), torch.from_numpy(q16).cuda()
        count = rx.execute_dev(di, dq, n_frames, first_call=True, flush=True)
        plain_bits, plain_trials = rx.fetch(count)
        assert (plain_trials >= 0).all()
        rx.set_outer_code(True)
        assert rx.execute_dev(di, dq, n_frames, first_call=True, flush=True) == count
        bits, trials = rx.fetch(count)
        status = rx.outer_code_status(count)
        assert np.array_equal(trials, plain_trials) and np.array_equal(bits, plain_bits)
        outs[bch] = (bits, status)
        rx.set_outer_code(False)
        with pytest.raises(Exception):
            rx.outer_code_status(count)
        rx.close()
    assert not outs[True][1].any()                                  # real parity: every frame a codeword
    assert (outs[False][1] == -1).all()                             # zero parity: not one
    assert np.array_equal(outs[True][0], outs[False][0])            # the message bits are the same stream either way


def test_pipelined_schedule_equals_plain_calls(torch_cuda):
    """receiver.pipeline_step (four streams, stages of neighbouring buffers overlapped) returns, one call late, exactly what the
    plain call sequence returns for the same buffers."""
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd.receiver import t2_receiver
    mode, lps, mod, fec_type, code_rate, snr, s2 = (4, 1, 6, 4, 0, 40), 200, 2, 0, 0, 16.0, 8
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    nb = t2_tx.plp_blocks_per_frame(m, lps, 16200 // (2 * (mod + 1)))
    k_bch = t2_tx.K_BCH[cid]
    bufs = []
    for b in range(3):
        ts = t2_tx.ts_packets(2 * nb * (k_bch // 1496 + 1) + 8, 40 + b)
        frames = []
        for f in range(2):
            cells, _, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts_slice(ts, f * nb, nb, k_bch), nb)
            frames.append(t2_tx.build_frame(m, cells, lps, 50 + 2 * b + f, snr_db=None, phase=0.0))
        bufs.append(frames)
    a = t2_receiver((*mode, lps, mod, fec_type, code_rate, 1, nb), dict(), max_frames=2)
    b_ = t2_receiver((*mode, lps, mod, fec_type, code_rate, 1, nb), dict(), max_frames=2)
    guard = a.chain.ofdm.guard_interval_size
    iq = [t2_tx.iq_stream(fr, guard, s2, snr, 60 + i)[:2] for i, fr in enumerate(bufs)]
    dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(q).cuda()) for i, q in iq]
    plain = []
    level = None
    for k, (di, dq) in enumerate(dev):
        r = a.demod_iq_dev(di, dq, 2, level_detect=level, first_call=(k == 0), flush=True)
        if level is None:
            level = float(a.front.state()["level_detect"])
        plain.append((r["bits"].cpu().numpy(), r["trials"].cpu().numpy(), r["p2_start"].copy()))
    # the pipelined receiver needs the same thresholds: take the level estimate the plain one saw after its first buffer
    lv0 = float(np.mean(np.abs(iq[0][0] / 16384.0)) * np.mean(np.abs(iq[0][1] / 16384.0)))
    outs = []
    for k, (di, dq) in enumerate(dev):
        r = b_.pipeline_step(di, dq, 2, lv0 if k == 0 else level, first_call=True)
        if r is not None:
            outs.append(r)
    outs.append(b_.pipeline_flush())
    b_.pipeline_sync()
    assert len(outs) == 3
    for (bits, trials, p2), r in zip(plain, outs):
        assert np.array_equal(r["p2_start"], p2)
        assert np.array_equal(r["trials"].cpu().numpy(), trials) and (trials >= 0).all()
        assert np.array_equal(r["bits"].cpu().numpy(), bits)
    a.close()
    b_.close()


def test_closed_loop_tracks_cfo_and_decodes(torch_cuda):
    """Symbol-by-symbol operation with every tracking loop closed, on a 16K frame structure with a frame-closing symbol: a
    100 Hz carrier offset (0.18 carrier spacings) and a static phase. P1 reports the offset frame after frame -- the reference's
    estimate is (deliberately or not) a fraction of the true value, :21,115, so the emulated tuner converges geometrically, as
    the real one does -- until it is below 10 Hz; then L1-pre/L1-post pass their CRCs, the guard-interval loop removes the
    residue, the phase loop and the sample-rate tracker run, and the transport stream of the frames after acquisition comes back."""
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd.receiver import t2_receiver, t2_closed_loop
    mode, lps, mod, fec_type, code_rate, snr, s2 = (4, 1, 1, 2, 0, 24), 400, 1, 0, 1, 12.0, 8
    n_frames, seed, cfo_hz = 7, 91, 100.0        # L1-post in BPSK: the reference takes hard decisions of the systematic bits (no L1 FEC)
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    cpf = 16200 // (2 * (mod + 1))
    nb = t2_tx.plp_blocks_per_frame(m, lps, cpf)
    k_bch = t2_tx.K_BCH[cid]
    ts = t2_tx.ts_packets(n_frames * nb * (k_bch // 1496 + 1) + 8, seed)
    pre = dict(type=0, bwt_ext=mode[1], s1=0, s2_field1=4, guard_interval=mode[3], papr=0, l1_post_mod=0, l1_cod=0, l1_fec_type=0,
               l1_post_size=lps, pilot_pattern=mode[2], num_t2_frames=2, num_data_symbols=mode[5], num_rf=1, t2_version=2)
    plp = [dict(id=0, plp_type=1, plp_cod=code_rate, plp_mod=mod, plp_rotation=1, plp_fec_type=fec_type, plp_num_blocks_max=nb,
                frame_interval=1, time_il_length=1, time_il_type=0, plp_mode=1)]
    info = t2_tx.l1_post_bits(dict(), plp, [dict(id=0, start=0, num_blocks=nb)])
    pre["l1_post_info_size"] = len(info)
    l1c = np.concatenate([t2_tx.l1_pre_cells(pre, 3), t2_tx.l1_post_cells(info, 0, lps, 4)])
    frames, bbframes = [], []
    for f in range(n_frames):
        cells, bb, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts_slice(ts, f * nb, nb, k_bch), nb)
        frames.append(t2_tx.build_frame(m, cells, lps, seed + f, snr_db=None, phase=0.0, l1_cells=l1c))
        bbframes.append(np.asarray(bb).reshape(nb, -1))
    rx = t2_receiver((*mode, lps, mod, fec_type, code_rate, 1, nb), dict(), max_frames=1)
    i16, q16, flen = t2_tx.iq_stream(frames, rx.chain.ofdm.guard_interval_size, s2, snr, seed)
    x = (i16.astype(np.float64) + 1j * q16.astype(np.float64)) * np.exp(1j * (2 * np.pi * cfo_hz / (64e6 / 7) * np.arange(len(i16)) + 0.7))
    i16, q16 = np.rint(x.real).astype(np.int16), np.rint(x.imag).astype(np.int16)
    tail = np.zeros(8192, np.int16)
    d_i, d_q = torch.from_numpy(np.concatenate([i16, tail])).cuda(), torch.from_numpy(np.concatenate([q16, tail])).cuda()
    cl = t2_closed_loop(rx)
    rx.front.execute_dev(d_i[:65536], d_q[:65536], [65536], torch.zeros(65600, dtype=torch.complex64, device="cuda"))   # a buffer to
    # settle level_detect (the AGC window of dvbt2_demodulator.cpp:235-251), as at start-up
    cl.level_detect = float(rx.front.state()["level_detect"])
    assert 0.01 < cl.level_detect < 0.08
    rx.front.reset()
    done = []
    step = 1 << 18                                                         # execute() buffers of 262144 samples
    for a in range(0, d_i.numel(), step):
        done += cl.execute(d_i[a:a + step], d_q[a:a + step], level_gain_changed=(a == 0))
    assert cl.p1_seen == n_frames
    hz = (64e6 / 7) / (2 * np.pi)
    assert abs(cl.tuner * hz - cfo_hz) < 30.0                                # the tuner stops once P1 REPORTS < 10 Hz (~0.4 x true)
    assert cl.l1 is not None and cl.l1[0].guard_interval == mode[3] and cl.l1[0].pilot_pattern == mode[2]
    assert cl.l1[2][0].plp_mod == mod and cl.l1[2][0].plp_cod == code_rate and cl.l1[3][0].num_blocks == nb
    assert abs((cl.tuner + cl.log[-1][1]) * hz - cfo_hz) < 1.0               # tuner + guard-interval loop settle on the offset
    assert {k for k, *_ in cl.log} == {"P2", "DATA", "FC"}
    assert 2 <= len(done) <= n_frames - 2                                    # the first frames went into acquisition (P1 -> tuner)
    first = n_frames - len(done)
    dfl_bytes = (k_bch - 80) // 8
    per_frame = (nb * dfl_bytes) // 187 - 1
    for f, (bits, trials) in enumerate(done, start=first):
        t = trials.cpu().numpy()
        assert (t >= 0).all(), (f, t)
        b = bits.cpu().numpy()
        assert np.array_equal(b[:, :k_bch], bbframes[f][:, :k_bch]), f          # every BBFRAME of the frame, bit for bit
        if f == first:                                                           # and as transport stream (the transmitter model
            got = rx.chain.ts_from_bits(b, t)                                    # restarts on a packet boundary every frame, so only
            sent = ts_slice(ts, f * nb, nb, k_bch).reshape(-1)                   # the first frame lines up with a fresh de-framer)
            assert np.array_equal(got[:per_frame * 188], sent[:per_frame * 188])
    rx.close()


def test_ordered_receiver_on_one_gpu_equals_one_call(torch_cuda):
    """shard.ordered_receiver (the multi-GPU form: ranks decode batch-aligned shares, rank 0 de-frames everything in frame order)
    driven with a real t2_rx as its decoder, world size 1: the TS equals that of one call over the whole buffer. The N > 1 merge is
    covered with gloo in tests/test_shard.py; what runs on each GPU is this decoder."""
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    from sdr_receiver_dvb_t2_amd.chain import ts_from_bits
    from sdr_receiver_dvb_t2_amd.shard import ordered_receiver
    mode, lps, mod, fec_type, code_rate, snr, s2 = (4, 1, 6, 4, 0, 40), 200, 2, 0, 0, 16.0, 8
    n_frames, seed = 3, 5
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    nb = t2_tx.plp_blocks_per_frame(m, lps, 2700)
    k_bch = t2_tx.K_BCH[cid]
    ts = t2_tx.ts_packets(n_frames * nb * (k_bch // 1496 + 1) + 8, seed)
    frames = []
    for f in range(n_frames):
        cells, _, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts_slice(ts, f * nb, nb, k_bch), nb)
        frames.append(t2_tx.build_frame(m, cells, lps, seed + f, snr_db=None, phase=0.0))
    i16, q16, frame_len = t2_tx.iq_stream(frames, m.fft_size // 128, s2, snr, seed)
    d_i, d_q = torch.from_numpy(i16).cuda(), torch.from_numpy(q16).cuda()
    rx = t2_rx(*mode, lps, mod, fec_type, code_rate, 1, nb, max_frames=n_frames)
    count = rx.execute_dev(d_i, d_q, n_frames, first_call=True, flush=True)
    bits, trials = rx.fetch(count)
    want = ts_from_bits(bits, trials)

    def decode(lo, hi):
        n = rx.execute_dev(d_i[lo * frame_len:], d_q[lo * frame_len:], hi - lo, first_call=True, flush=True)
        return rx.fetch(n)
    orx = ordered_receiver(decode, nb, 32, 0, None)
    got = orx.execute(n_frames)
    assert got.size > 50000 and np.array_equal(got, want)
    orx.close()
    rx.close()


def test_stage_timers_config2_entry_point_and_sync_sums(torch_cuda):
    """t2gpu_rx_stage_ms (HIP events between the stages of a call), t2gpu_rx_fft_eq_demap_dev (BASELINE config 2: FFT + equalisers +
    de-interleave + demap of the frames the last call left in the handle) and t2gpu_rx_sync_sums (the per-symbol synchronisation
    sums the reference always forms; they equal what the stand-alone equaliser entry points return for the same spectra)."""
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    mode, lps, mod, fec_type, code_rate, snr, s2 = (4, 1, 6, 4, 0, 40), 200, 2, 0, 0, 16.0, 8
    n_frames, seed = 2, 9
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    nb = t2_tx.plp_blocks_per_frame(m, lps, 2700)
    k_bch = t2_tx.K_BCH[cid]
    ts = t2_tx.ts_packets(n_frames * nb * (k_bch // 1496 + 1) + 8, seed)
    frames = []
    for f in range(n_frames):
        cells, _, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts_slice(ts, f * nb, nb, k_bch), nb)
        frames.append(t2_tx.build_frame(m, cells, lps, seed + f, snr_db=None, phase=0.0))
    i16, q16, frame_len = t2_tx.iq_stream(frames, m.fft_size // 128, s2, snr, seed)
    rx = t2_rx(*mode, lps, mod, fec_type, code_rate, 1, nb, max_frames=n_frames)
    count = rx.execute_dev(torch.from_numpy(i16).cuda(), torch.from_numpy(q16).cuda(), n_frames, first_call=True)
    ms = rx.stage_ms()
    assert set(ms) == set(rx.STAGES) and all(v > 0 for v in ms.values()), ms
    assert ms["ldpc"] == pytest.approx(rx.last_ldpc_ms(), rel=0.2)
    bits0, trials0 = rx.fetch(count)
    sync0 = rx.sync_sums(n_frames)
    assert sync0.shape == (n_frames * m.len_frame, 2) and np.isfinite(sync0).all() and np.abs(sync0[:, 0]).max() > 0
    rx.fft_eq_demap_dev(n_frames)
    ms2 = rx.stage_ms()
    assert all(ms2[k] > 0 for k in ("fft", "equalise", "ti", "demap")) and all(ms2[k] < 0 for k in ("front", "p1", "guard_corr", "ldpc", "descramble"))
    assert np.array_equal(rx.sync_sums(n_frames), sync0)                       # the same spectra equalised again: the same sums
    rx.close()


def test_overlapped_decode_equals_plain_calls(torch_cuda, monkeypatch):
    """t2gpu_rx_set_overlap: the decode of a call on a stream of the handle's own beside the next call's front half, small decodes two at
    a time (three LLR buffers rotate, the waiting frames carried from one to the next, two decode sets alternate, the L1 cells sent home
    by a kernel of the call's own stream) -- and with T2GPU_RX_PAIR=0 the form every larger decode takes: one set, one after the other. Six one-frame calls of a 16K / 64-QAM /
    16200 r1/2 stream (41 FEC frames per T2 frame: SIMD batches form across calls, 9 .. 27 frames wait in between) with the library's
    host end on: the same packed rows per call, the same verdicts, the same TS bytes as the plain schedule, and the flush at the end."""
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    mode, lps, mod, fec_type, code_rate, s2 = (4, 1, 6, 4, 0, 40), 200, 2, 0, 0, 8
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    cpf = 16200 // (2 * (mod + 1))
    nb_full = t2_tx.plp_blocks_per_frame(m, lps, cpf)
    nb = 41                                                        # not a multiple of 32: batches straddle the calls
    assert nb <= nb_full
    k_bch = t2_tx.K_BCH[cid]
    n_frames, seed = 6, 123
    ts = t2_tx.ts_packets(n_frames * nb * (k_bch // 1496 + 1) + 8, seed)
    frames, pos = [], 0
    for f in range(n_frames):
        cells, _, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts_slice(ts, pos, nb, k_bch), nb)
        l1 = t2_tx.l1_cells(mode, lps, mod, fec_type, code_rate, nb, frame_idx=f)
        frames.append(t2_tx.build_frame(m, cells, lps, seed + f, snr_db=None, phase=0.0, l1_cells=l1))
        pos += nb
    probe = t2_rx(*mode, lps, mod, fec_type, code_rate, 1, nb, max_frames=1)
    i16, q16, frame_len = t2_tx.iq_stream(frames, probe.geometry.guard_interval_size, s2, 16.0, seed)
    assert frame_len == probe.frame_len
    probe.close()
    di, dq = torch.from_numpy(i16).cuda(), torch.from_numpy(q16).cuda()

    def run(overlap, pair=True, collect=0):
        rx = t2_rx(*mode, lps, mod, fec_type, code_rate, 1, nb, max_frames=1)
        rx.ts_enable(0, l1_check=True)
        if overlap:
            monkeypatch.setenv("T2GPU_RX_PAIR", "1" if pair else "0")
            monkeypatch.setenv("T2GPU_RX_COLLECT", str(collect))         # 0: a decode per call (round 5's form); n: collect n batches per decode
            rx.set_overlap(True)
        rows, verdicts, counts = [], [], []
        for f in range(n_frames):
            a = f * frame_len
            n = rx.execute_dev(di[a:a + frame_len], dq[a:a + frame_len], 1, first_call=(f == 0))
            counts.append(n)
            if n:
                r, t = rx.fetch_packed(n)
                rows.append(r); verdicts.append(t)
        n = rx.flush_dev()
        counts.append(n)
        if n:
            r, t = rx.fetch_packed(counts[-2] + n)                  # the rows of the flush follow the last back half's
            rows.append(r[counts[-2]:]); verdicts.append(t[counts[-2] // 32:])
        ts_bytes = rx.ts_read(wait_all=True)
        c = rx.ts_counters()
        rx.close()
        return counts, rows, verdicts, ts_bytes, c

    pc, pr, pv, pts, pcnt = run(False)
    oc, orr, ov, ots, ocnt = run(True)
    assert pc == oc and sum(pc) == n_frames * nb and any(0 < x < nb for x in pc[:-1])
    assert len(pr) == len(orr) and all(np.array_equal(a, b) for a, b in zip(pr, orr))
    assert all(np.array_equal(a, b) for a, b in zip(pv, ov)) and all((v >= 0).all() for v in pv)
    assert pts.size > 0 and np.array_equal(pts, ots)
    for k in ("t2_frames", "fec_frames", "fec_frames_dropped_ldpc", "fec_frames_dropped_l1", "l1_pre_crc_errors", "l1_post_crc_errors", "ts_bytes"):
        assert pcnt[k] == ocnt[k], k
    assert pcnt["fec_frames_dropped_l1"] == 0 and pcnt["fec_frames"] == n_frames * nb
    sc, sr, sv, sts, scnt = run(True, pair=False)
    assert sc == pc and all(np.array_equal(a, b) for a, b in zip(pr, sr)) and all(np.array_equal(a, b) for a, b in zip(pv, sv))
    assert np.array_equal(pts, sts) and scnt["fec_frames"] == pcnt["fec_frames"]
    # round 6: small calls COLLECT -- their LLR frames wait in the handle until 3 (here) SIMD batches are there, every decode is a whole
    # number of rounds of that many resident batches, a call returns the rows whose decode it launched (0 while collecting). The same rows,
    # verdicts and TS bytes in the same order; only the calls they come back from differ
    cc, cr, cv, cts, ccnt = run(True, collect=3)
    assert sum(cc) == sum(pc) and all(x % (3 * 32) == 0 for x in cc[:-1]) and 0 in cc[:-1], cc
    assert np.array_equal(np.concatenate(cr), np.concatenate(pr)) and np.array_equal(np.concatenate(cv), np.concatenate(pv))
    assert np.array_equal(pts, cts)
    for k in ("t2_frames", "fec_frames", "fec_frames_dropped_ldpc", "fec_frames_dropped_l1", "ts_bytes"):
        assert pcnt[k] == ccnt[k], k


@pytest.mark.parametrize("caller", ["own stream", "null stream"])
def test_overlapped_calls_take_their_input_from_the_callers_stream(torch_cuda, caller):
    """Round 6: in the overlap mode a call runs on streams of the handle's own (csrc/t2gpu_rx.cpp, StreamBundle); its chain waits for the
    caller's stream at entry. ONE input buffer, refilled for every call on the caller's stream BEHIND a few milliseconds of other work
    there and handed over at once: the TS must be the TS of plain calls on whole inputs (without the wait the front end reads the previous
    frame, or half of this one). On a stream of the caller's own and on the legacy NULL stream (which the handle's streams, blocking
    streams in its sense, follow without an event)."""
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    mode, lps, mod, fec_type, code_rate, s2 = (4, 1, 6, 4, 0, 40), 200, 2, 0, 0, 8
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    nb, n_frames, seed = 41, 5, 321
    k_bch = t2_tx.K_BCH[cid]
    ts = t2_tx.ts_packets(n_frames * nb * (k_bch // 1496 + 1) + 8, seed)
    frames, pos = [], 0
    for f in range(n_frames):
        cells, _, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts_slice(ts, pos, nb, k_bch), nb)
        l1 = t2_tx.l1_cells(mode, lps, mod, fec_type, code_rate, nb, frame_idx=f)
        frames.append(t2_tx.build_frame(m, cells, lps, seed + f, snr_db=None, phase=0.0, l1_cells=l1))
        pos += nb
    probe = t2_rx(*mode, lps, mod, fec_type, code_rate, 1, nb, max_frames=1)
    i16, q16, frame_len = t2_tx.iq_stream(frames, probe.geometry.guard_interval_size, s2, 16.0, seed)
    probe.close()
    di, dq = torch.from_numpy(i16).cuda(), torch.from_numpy(q16).cuda()
    torch.cuda.synchronize()

    def run(overlap):
        rx = t2_rx(*mode, lps, mod, fec_type, code_rate, 1, nb, max_frames=1)
        rx.ts_enable(0, l1_check=True)
        if overlap:
            rx.set_overlap(True)
        st = torch.cuda.Stream() if caller == "own stream" else torch.cuda.default_stream()
        buf_i, buf_q = torch.zeros(frame_len, dtype=torch.int16, device="cuda"), torch.zeros(frame_len, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        for f in range(n_frames):
            a = f * frame_len
            with torch.cuda.stream(st):
                if overlap:
                    torch.cuda._sleep(20_000_000)                  # ~10 ms of someone else's work on the caller's stream, then the refill
                buf_i.copy_(di[a:a + frame_len], non_blocking=True)
                buf_q.copy_(dq[a:a + frame_len], non_blocking=True)
                rx.execute_dev(buf_i, buf_q, 1, first_call=(f == 0), stream=st.cuda_stream)
        rx.flush_dev(stream=st.cuda_stream)
        out = rx.ts_read(wait_all=True)
        c = rx.ts_counters()
        rx.close()
        return out, c

    pts, pcnt = run(False)
    ots, ocnt = run(True)
    assert pts.size > 0 and pcnt["fec_frames"] == n_frames * nb and pcnt["fec_frames_dropped_ldpc"] == 0
    assert np.array_equal(pts, ots)
    for k in ("t2_frames", "fec_frames", "fec_frames_dropped_ldpc", "fec_frames_dropped_l1", "ts_bytes"):
        assert pcnt[k] == ocnt[k], k


@pytest.mark.parametrize("pattern", [(1,) * 6, (2, 2, 2), (4, 4), (1, 4, 1, 16)])
def test_overlapped_decode_on_the_benchmark_mode_for_every_call_size(torch_cuda, monkeypatch, pattern):
    """bench.py's frames_sweep rows run CFG-A (32K / 256-QAM / 64800 r = 3/4, 202 FEC frames per T2 frame) at 1, 2, 4 ..
    frames per call with t2gpu_rx_set_overlap; this is their correctness twin. Calls of the given sizes -- the last pattern changes size
    from call to call, so the decode sets and the three LLR buffers rotate across the boundary where two decodes no longer fit the device
    together -- with pairs allowed and not: the same row counts, packed rows, verdicts and TS bytes as the plain schedule (clamped LLRs,
    so that batches decode and the rows carry BBFRAMEs)."""
    torch = torch_cuda
    import bench
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    w = bench.Workload(bench.CONFIGS[3])
    ui, uq, _ = bench.make_frames(w, 2, 22.0, 4242)
    total = sum(pattern)
    reps = (total + 1) // 2
    di = torch.from_numpy(np.concatenate([ui] * reps)[:total].reshape(-1)).cuda()
    dq = torch.from_numpy(np.concatenate([uq] * reps)[:total].reshape(-1)).cuda()
    flen = w.frame_samples

    def run(overlap, pair=True, collect=0):
        rx = t2_rx(*w.mode, w.lps, w.plp[0], w.plp[1], w.plp[2], w.plp[3], w.nb, max_frames=max(pattern), saturate_llr=True)
        rx.ts_enable(0, l1_check=True)
        if overlap:
            monkeypatch.setenv("T2GPU_RX_PAIR", "1" if pair else "0")
            monkeypatch.setenv("T2GPU_RX_COLLECT", str(collect))
            rx.set_overlap(True)
        counts, rows, verdicts, at = [], [], [], 0
        for c, nf in enumerate(pattern):
            n = rx.execute_dev(di[at * flen:(at + nf) * flen], dq[at * flen:(at + nf) * flen], nf, first_call=(c == 0))
            at += nf
            counts.append(n)
            if n:
                r, t = rx.fetch_packed(n)
                rows.append(r); verdicts.append(t)
        n = rx.flush_dev()
        counts.append(n)
        if n:
            r, t = rx.fetch_packed(counts[-2] + n)
            rows.append(r[counts[-2]:]); verdicts.append(t[counts[-2] // 32:])
        ts_bytes = rx.ts_read(wait_all=True)
        c = rx.ts_counters()
        rx.close()
        return counts, rows, verdicts, ts_bytes, c

    pc, pr, pv, pts, pcnt = run(False)
    assert sum(pc) == total * w.nb and pts.size > 0 and sum(int((v >= 0).sum()) for v in pv) > 0
    for pair in (True, False):
        oc, orr, ov, ots, ocnt = run(True, pair)
        assert oc == pc, (pair, oc, pc)
        assert len(orr) == len(pr) and all(np.array_equal(a, b) for a, b in zip(pr, orr)), pair
        assert all(np.array_equal(a, b) for a, b in zip(pv, ov)), pair
        assert np.array_equal(pts, ots), pair
        for k in ("t2_frames", "fec_frames", "fec_frames_dropped_ldpc", "fec_frames_dropped_l1", "ts_bytes"):
            assert pcnt[k] == ocnt[k], (pair, k)
    # the library's default since round 6: calls of fewer than 28 batches collect until 14 are there (one decode resident at a time, whole
    # rounds of 14 slots); larger calls (the 16-frame one of the last pattern) decode whole rounds of all 16 slots. Same rows in the same order, same TS
    monkeypatch.delenv("T2GPU_RX_COLLECT", raising=False)
    cc, cr, cv, cts, ccnt = run(True, collect=14)
    assert sum(cc) == sum(pc), (cc, pc)
    assert np.array_equal(np.concatenate(cr), np.concatenate(pr)) and np.array_equal(np.concatenate(cv), np.concatenate(pv))
    assert np.array_equal(pts, cts)
    for k in ("t2_frames", "fec_frames", "fec_frames_dropped_ldpc", "fec_frames_dropped_l1", "ts_bytes"):
        assert pcnt[k] == ccnt[k], k
    if max(pattern) <= 4:
        assert all(x % (14 * 32) == 0 for x in cc[:-1]), cc
    else:
        # larger calls: whole rounds of all 16 resident slots, what is left of a round waits for the next call (or the flush)
        big = [x for x, nf in zip(cc, pattern) if nf * w.nb >= 28 * 32]
        assert big and all(x > 0 and x % (16 * 32) == 0 for x in big), cc
