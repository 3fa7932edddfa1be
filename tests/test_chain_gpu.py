"""GPU tier, end to end: transport-stream packets go through the transmitter model (tests/t2_tx.py: BBFRAMEs, scrambling,
LDPC, bit interleaver, rotated QAM, cell + time interleaver, frame builder with pilots, IFFT, AWGN) and come back out of
the GPU chain (FFT -> equaliser/freq de-interleave -> time/cell de-interleave -> demap -> LDPC -> descramble -> TS) byte for
byte. This is the 'bit-exact after BCH' criterion of the north star against an independent forward model."""
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


def run_chain(torch, mode, l1_post_size, mod, fec_type, code_rate, snr_db, n_frames, seed, saturate=False, expect_ok=True):
    import sdr_receiver_dvb_t2_amd as pkg
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    cpf = (64800 if fec_type else 16200) // (2 * (mod + 1))
    nb = t2_tx.plp_blocks_per_frame(m, l1_post_size, cpf)
    k_bch = t2_tx.K_BCH[cid]
    ts = t2_tx.ts_packets(n_frames * nb * (k_bch // 1496 + 1) + 8, seed)
    syms, pos = [], 0
    for f in range(n_frames):
        stream, frames, used_bits = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts_slice(ts, pos, nb, k_bch), nb)
        syms.append(t2_tx.build_frame(m, stream, l1_post_size, seed + f, snr_db=snr_db, phase=0.4 * (f + 1)))
        pos += nb
    chain = pkg.t2_chain(*mode, l1_post_size, mod, fec_type, code_rate, 1, nb, max_frames=n_frames, saturate_llr=saturate)
    x = torch.from_numpy(np.stack(syms).view(np.float32).reshape(n_frames, m.len_frame, m.fft_size, 2)).cuda()
    bits, trials = chain.demod_dev(x, flush=True)
    torch.cuda.synchronize()
    trials = trials.cpu().numpy()
    if expect_ok:
        assert (trials >= 0).all(), trials
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
])
def test_transport_stream_round_trip(torch_cuda, name, mode, lps, mod, fec_type, code_rate, snr, saturate):
    got, ts, nb = run_chain(torch_cuda, mode, lps, mod, fec_type, code_rate, snr, n_frames=1, seed=11, saturate=saturate)
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
