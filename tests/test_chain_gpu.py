"""GPU tier, end to end: transport-stream packets go through the transmitter model (tests/t2_tx.py: BBFRAMEs, scrambling,
LDPC, bit interleaver, rotated QAM, cell + time interleaver, frame builder with pilots, IFFT, AWGN) and come back out of
the GPU chain (FFT -> equaliser/freq de-interleave -> time/cell de-interleave -> demap -> LDPC -> descramble -> TS) byte for
byte. A round trip, not an independent pin: tests/t2_tx.py INVERTS the oracle's tables (carrier maps, de-interleaver permutations, bit
de-interleaver addresses), so it shows that the chain is the inverse of that transmitter; what pins payload bits to the reference
itself are tests/test_ref_pins_gpu.py (the reference's FEC chain objects, and for 256-QAM its ldpc_decoder::execute slot)."""
import numpy as np
import pytest

import oracle_lib as ol
import t2_tx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available()
    return torch


def run_chain(torch, mode, l1_post_size, mod, fec_type, code_rate, snr_db, n_frames, seed, saturate=False, expect_ok=True, bch=False):
    import sdr_receiver_dvb_t2_amd as pkg
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    cpf = (64800 if fec_type else 16200) // (2 * (mod + 1))
    nb = t2_tx.plp_blocks_per_frame(m, l1_post_size, cpf)
    k_bch = t2_tx.K_BCH[cid]
    ts = t2_tx.ts_packets(n_frames * nb * (k_bch // 1496 + 1) + 8, seed)
    syms, pos = [], 0
    for f in range(n_frames):
        stream, frames, used_bits = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts_slice(ts, pos, nb, k_bch), nb, bch=bch)
        syms.append(t2_tx.build_frame(m, stream, l1_post_size, seed + f, snr_db=snr_db, phase=0.4 * (f + 1)))
        pos += nb
    chain = pkg.t2_chain(*mode, l1_post_size, mod, fec_type, code_rate, 1, nb, max_frames=n_frames, saturate_llr=saturate,
                         outer_code=bch)
    x = torch.from_numpy(np.stack(syms).view(np.float32).reshape(n_frames, m.len_frame, m.fft_size, 2)).cuda()
    bits, trials = chain.demod_dev(x, flush=True)
    torch.cuda.synchronize()
    trials = trials.cpu().numpy()
    if expect_ok:
        assert (trials >= 0).all(), trials
    if bch:                                                       # real BCH parity on air: every decoded frame checks clean
        assert not chain.outer_code_status.cpu().numpy().any()
    got = chain.ts_from_bits(bits.cpu().numpy(), trials)
    chain.close()
    return got, ts, nb


def ts_slice(ts, frame_pos, nb, k_bch):
    """The transmitter model packs a continuous packet flow; each TI block re-packs from a packet boundary here (the
    de-framer re-synchronises through SYNCD), so give every frame its own run of packets."""
    per = nb * (k_bch // 1496 + 1)
    a = (frame_pos // nb) * per
    return ts[a:a + per]


@pytest.mark.parametrize("name,mode,lps,mod,fec_type,code_rate,snr,saturate", [
    ("CFG-B 16K ext PP7 64-QAM 16200 r1/2", (4, 1, 6, 4, 0, 40), 200, 2, 0, 0, 14.0, False),
    ("32K normal PP4 64-QAM 64800 r2/3", (5, 0, 3, 0, 0, 20), 400, 2, 1, 2, 18.0, False),
    ("16K ext PP2 GI1/8 with frame-closing symbol, 16-QAM 16200 r3/5", (4, 1, 1, 2, 0, 24), 150, 1, 0, 1, 12.0, False),
    # 256-QAM: the reference's int8 cast wraps on the outer points at every SNR (see include/t2gpu.h, t2gpu_demap_configure);
    # with the clamping extension the same chain decodes CFG-A
    ("CFG-A 32K ext PP7 256-QAM 64800 r3/4 (clamped LLRs)", (5, 1, 6, 4, 0, 59), 350, 3, 1, 3, 22.0, True),
    # BASELINE config 5's workload: CFG-A with r = 2/3 (demux_256_fec_size_normal_2_3, llr_demapper.cpp:677), same extension
    ("CFG-C 32K ext PP7 256-QAM 64800 r2/3 (clamped LLRs)", (5, 1, 6, 4, 0, 59), 350, 3, 1, 2, 22.0, True),
    # the rest of SURVEY.md 8(f)-4: QPSK, the remaining pilot patterns, tone reservation, the remaining code rates
    ("16K normal PP1 GI1/4 QPSK 16200 r4/5", (4, 0, 0, 3, 0, 17), 150, 0, 0, 4, 9.0, False),
    ("32K normal PP8 GI1/16 QPSK 64800 r5/6 (FEC block larger than LDS: per-cell TI path)", (5, 0, 7, 1, 0, 30), 300, 0, 1, 5, 10.0, False),
    ("32K ext PP6 GI1/16 tone reservation, 64-QAM 64800 r4/5", (5, 1, 5, 1, 2, 41), 400, 2, 1, 4, 20.0, False),
    ("16K ext PP3 GI1/8 tone reservation, 16-QAM 64800 r1/2", (4, 1, 2, 2, 2, 33), 200, 1, 1, 0, 11.0, False),
    ("16K normal PP5 GI1/16 16-QAM 16200 r2/3", (4, 0, 4, 1, 0, 45), 150, 1, 0, 2, 13.0, False),
    ("32K ext PP4 GI1/32 64-QAM 16200 r5/6", (5, 1, 3, 0, 0, 60), 350, 2, 0, 5, 21.0, False),
])
def test_transport_stream_round_trip(torch_cuda, name, mode, lps, mod, fec_type, code_rate, snr, saturate):
    # every second case also carries real BCH parity and runs the opt-in outer-code stage
    got, ts, nb = run_chain(torch_cuda, mode, lps, mod, fec_type, code_rate, snr, n_frames=1, seed=11, saturate=saturate,
                            bch=(mod + code_rate) % 2 == 0)
    cid = ol.code_id(fec_type, code_rate)
    dfl_bytes = ((t2_tx.K_BCH[cid] - 80) // 8)
    sent = ts.reshape(-1)
    # every whole packet carried by the frame's BBFRAMEs comes back, in order, byte for byte
    n_pkts = (nb * dfl_bytes) // 187 - 1
    assert got.size >= n_pkts * 188
    assert np.array_equal(got[:n_pkts * 188], sent[:n_pkts * 188])


def test_cfg_a_reference_semantics_drops_every_batch(torch_cuda):
    """CFG-A with the reference's wrapping cast: the LDPC gives up on every SIMD batch (-1) and the reference would emit
    nothing -- the chain reports exactly that instead of inventing output."""
    got, ts, nb = run_chain(torch_cuda, (5, 1, 6, 4, 0, 59), 350, 3, 1, 3, 22.0, n_frames=1, seed=11, saturate=False, expect_ok=False)
    assert got.size == 0


def test_three_ti_blocks_per_frame(torch_cuda):
    """time_il_type 0 with time_il_length 3: the frame's FEC blocks are split into three TI blocks (sizes as
    time_deinterleaver::l1_dyn_execute computes them, the later blocks take the remainder), each interleaved on its own and
    each with its own SNR estimate in the demapper."""
    torch = torch_cuda
    import sdr_receiver_dvb_t2_amd as pkg
    mode, lps, mod, fec_type, code_rate, snr = (4, 1, 6, 4, 0, 40), 200, 2, 0, 0, 14.0
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    cpf = 16200 // (2 * (mod + 1))
    nb = t2_tx.plp_blocks_per_frame(m, lps, cpf)
    assert nb % 3 != 0                                                       # uneven split: exercises the remainder rule
    k_bch = t2_tx.K_BCH[cid]
    ts = t2_tx.ts_packets(nb * (k_bch // 1496 + 1) + 8, 5)
    frames, used = t2_tx.bbframes_hem(ts, k_bch, nb)
    cells = t2_tx.cells_from_codewords(t2_tx.fec_encode(cid, t2_tx.scramble(frames)), mod, fec_type, code_rate, True)
    base = nb // 3
    sizes = [base + (1 if j >= 3 - nb % 3 else 0) for j in range(3)]
    stream, pos = [], 0
    for n in sizes:
        stream.append(t2_tx.interleave_ti_block(cells[pos:pos + n]))
        pos += n
    sym = t2_tx.build_frame(m, np.concatenate(stream), lps, 9, snr_db=snr, phase=0.5)
    chain = pkg.t2_chain(*mode, lps, mod, fec_type, code_rate, 1, nb, max_frames=1, time_il_length=3)
    assert chain.ti_blocks == sizes
    x = torch.from_numpy(sym[None].view(np.float32).reshape(1, m.len_frame, m.fft_size, 2)).cuda()
    bits, trials = chain.demod_dev(x, flush=True)
    torch.cuda.synchronize()
    t = trials.cpu().numpy()
    assert (t >= 0).all(), t
    got = chain.ts_from_bits(bits.cpu().numpy(), t)
    n_pkts = (nb * ((k_bch - 80) // 8)) // 187 - 1
    assert np.array_equal(got[:n_pkts * 188], ts.reshape(-1)[:n_pkts * 188])
    chain.close()


def test_two_plps_in_one_frame(torch_cuda):
    """Two PLPs of one modulation and code share the frame (PLP 0: one TI block; PLP 1: two TI blocks). The frame de-multiplexer follows time_deinterleaver::execute (time_deinterleaver.cpp:296-312,357-368), the
    FEC frames of both PLPs fill the SIMD batches in arrival order (llr_demapper.cpp:742-760) and only need_plp's BBFRAMEs
    become transport stream (bb_de_header.cpp:139-142)."""
    torch = torch_cuda
    import sdr_receiver_dvb_t2_amd as pkg
    mode, lps, mod, fec_type, code_rate, snr = (4, 1, 6, 4, 0, 40), 200, 2, 0, 0, 14.0
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    cpf = 16200 // (2 * (mod + 1))
    nb = t2_tx.plp_blocks_per_frame(m, lps, cpf)
    n0 = nb // 3
    n1 = nb - n0
    assert n1 % 2 == 1                                                       # uneven TI blocks in PLP 1
    k_bch = t2_tx.K_BCH[cid]
    ts0 = t2_tx.ts_packets(n0 * (k_bch // 1496 + 1) + 8, 21)
    ts1 = t2_tx.ts_packets(n1 * (k_bch // 1496 + 1) + 8, 22)
    fr0, _ = t2_tx.bbframes_hem(ts0, k_bch, n0)
    fr1, _ = t2_tx.bbframes_hem(ts1, k_bch, n1)
    c0 = t2_tx.cells_from_codewords(t2_tx.fec_encode(cid, t2_tx.scramble(fr0)), mod, fec_type, code_rate, True)
    c1 = t2_tx.cells_from_codewords(t2_tx.fec_encode(cid, t2_tx.scramble(fr1)), mod, fec_type, code_rate, True)
    a = n1 // 2
    stream = np.concatenate([t2_tx.interleave_ti_block(c0), t2_tx.interleave_ti_block(c1[:a]), t2_tx.interleave_ti_block(c1[a:])])
    sym = t2_tx.build_frame(m, stream, lps, 9, snr_db=snr, phase=-0.3)
    plps = [dict(num_blocks=n0, start=0, plp_rotation=1, time_il_length=1),
            dict(num_blocks=n1, start=n0 * cpf, plp_rotation=1, time_il_length=2)]
    x = torch.from_numpy(sym[None].view(np.float32).reshape(1, m.len_frame, m.fft_size, 2)).cuda()
    for need in (1, 0):
        chain = pkg.t2_chain(*mode, lps, mod, fec_type, code_rate, 1, nb, max_frames=1, plps=plps, need_plp=need)
        assert chain.plan == [(0, 0, n0, n0 * cpf), (1, n0 * cpf, a, a * cpf), (1, (n0 + a) * cpf, n1 - a, (n1 - a) * cpf)]
        bits, trials = chain.demod_dev(x, flush=True)
        torch.cuda.synchronize()
        t = trials.cpu().numpy()
        assert (t >= 0).all(), t
        b = bits.cpu().numpy()
        assert chain.last_tags == [0] * n0 + [1] * n1
        assert np.array_equal(b[:n0], fr0) and np.array_equal(b[n0:], fr1)  # every BBFRAME of both PLPs, bit for bit
        got = chain.ts_from_bits(b, t)
        ts, n = (ts1, n1) if need == 1 else (ts0, n0)
        n_pkts = (n * ((k_bch - 80) // 8)) // 187 - 1
        assert np.array_equal(got[:n_pkts * 188], ts.reshape(-1)[:n_pkts * 188])
        assert got.size <= (n_pkts + 2) * 188                               # nothing of the other PLP leaked in
        chain.close()


@pytest.mark.parametrize("q_delay,decodes", [(False, False), (True, True)])
def test_plain_constellation_plp_loses_its_q_alignment(torch_cuda, q_delay, decodes):
    """PLP_ROTATION = 0. The reference's time de-interleaver moves every Q one cell back whatever the PLP signals (no test of
    plp_rotation in time_deinterleaver.cpp:316-336; only the demapper looks at it, llr_demapper.cpp:167,237,373,544), so a
    standard non-rotated PLP -- which carries no cyclic Q delay -- comes out with I and Q of neighbouring cells paired and
    every SIMD batch is dropped. The chain reproduces that; a transmitter that delays Q without rotating decodes, which pins
    the cause."""
    torch = torch_cuda
    import sdr_receiver_dvb_t2_amd as pkg
    mode, lps, mod, fec_type, code_rate, snr = (4, 1, 6, 4, 0, 40), 200, 1, 0, 0, 14.0
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    cpf = 16200 // (2 * (mod + 1))
    nb = 64
    k_bch = t2_tx.K_BCH[cid]
    ts = t2_tx.ts_packets(nb * (k_bch // 1496 + 1) + 8, 31)
    fr, _ = t2_tx.bbframes_hem(ts, k_bch, nb)
    cells = t2_tx.cells_from_codewords(t2_tx.fec_encode(cid, t2_tx.scramble(fr)), mod, fec_type, code_rate, False, q_delay=q_delay)
    sym = t2_tx.build_frame(m, t2_tx.interleave_ti_block(cells), lps, 3, snr_db=snr, phase=0.2)
    chain = pkg.t2_chain(*mode, lps, mod, fec_type, code_rate, 0, nb, max_frames=1)
    x = torch.from_numpy(sym[None].view(np.float32).reshape(1, m.len_frame, m.fft_size, 2)).cuda()
    bits, trials = chain.demod_dev(x, flush=True)
    torch.cuda.synchronize()
    t = trials.cpu().numpy()[:2]                                            # the 64 frames of the PLP = two whole batches
    if decodes:
        assert (t >= 0).all(), t
        assert np.array_equal(bits.cpu().numpy()[:nb], fr)
    else:
        assert (t < 0).all(), t
    chain.close()
