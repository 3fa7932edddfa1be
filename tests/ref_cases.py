"""Shared by tests/golden/make_t2_golden.py (runs the REFERENCE here in the build container) and by the CPU / GPU tiers (run the
oracle / the HIP path): the deterministic inputs of every pinned case, and how each side is driven. Inputs are integer-valued
(numpy PCG64 integer streams and values rounded onto a power-of-two grid), so they regenerate bit-identically anywhere; the
fixtures hold a SHA-256 of every input and the tests check it before comparing outputs.

Reference quirks that shape the case list (all seen by running the reference under AddressSanitizer here):
  * L1-post BPSK: p2_symbol.cpp:405-409 allocates l1_post_size * l1_post_mod * 2 bytes = 0 and l1_post_info writes n_post bits
    into it -> every case with real L1 signalling uses QPSK L1-post;
  * 16-QAM and 256-QAM on 16200-bit FEC frames: llr_demapper.cpp steps 16 / 32 LLRs at a time and tests idx_out == 16200, which
    never happens (16200 is no multiple) -> the address pointer runs off its table; such PLPs are outside what the reference runs;
  * qam16/qam64/qam256/qpsk keep `static int blocks; static int8_t* out` (llr_demapper.cpp:168,242,...): one demapper per
    process, so the generator runs every case in a process of its own."""
import hashlib
import zlib

import numpy as np

import oracle_lib as ol
import t2_tx

GRID_SPEC = 4096.0          # spectrum values are multiples of 1/4096
GRID_CELL = 8192.0          # cell values are multiples of 1/8192


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def crc_rows(a):
    a = np.ascontiguousarray(a)
    return np.array([zlib.crc32(row.tobytes()) for row in a.reshape(a.shape[0], -1)], np.uint32)


def quantise(x, grid):
    x = np.asarray(x, np.complex128)
    q = np.stack([np.rint(x.real * grid), np.rint(x.imag * grid)], axis=-1)
    assert np.abs(q).max() < 32767
    return q.astype(np.int16)


def dequantise(q, grid):
    q = np.asarray(q, np.float64)
    return (q[..., 0] + 1j * q[..., 1]).astype(np.complex64) / np.float32(grid)


# ------------------------------------------------------------------------------------------------ OFDM modes / symbol-level cases
# (fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data)
SYM_MODES = {
    "cfg_a": (5, 1, 6, 4, 0, 59),            # CFG-A and CFG-C (the code rate does not reach the OFDM side)
    "cfg_b": (4, 1, 6, 4, 0, 40),
    "fc16k": (4, 1, 1, 2, 0, 24),            # 16K extended PP2 GI 1/8: frame-closing symbol
    "n32k_pp4": (5, 0, 3, 0, 0, 20),         # normal carriers (the reference reads its P2 with the extended tables)
    "tr32k_pp6": (5, 1, 5, 1, 2, 41),        # tone reservation + frame-closing symbol
}
SYM_L1_POST_SIZE = {"cfg_a": 350, "cfg_b": 200, "fc16k": 400, "n32k_pp4": 400, "tr32k_pp6": 400}


def sym_frame(name, seed=5, snr_db=25.0):
    """One T2 frame of mode `name` with real L1 signalling (QPSK L1-post) and random 256-QAM payload cells, as fft-shifted spectra
    on the 1/4096 grid: returns (ora_mode, int16 [len_frame][fft_size][2], L1-pre dict, PLP dict list)."""
    mode, lps = SYM_MODES[name], SYM_L1_POST_SIZE[name]
    m = ol.ora_mode(*mode)
    nb = 3
    pre = dict(type=0, bwt_ext=mode[1], s1=0, s2_field1=(4 if mode[0] == 4 else 5), guard_interval=mode[3], papr=mode[4], l1_post_mod=1,
               l1_cod=0, l1_fec_type=0, l1_post_size=lps, pilot_pattern=mode[2], num_t2_frames=2, num_data_symbols=mode[5], num_rf=1,
               t2_version=2, cell_id=0x1234, network_id=0x3085, t2_system_id=0x8001, tx_id_availability=0)
    plp = [dict(id=0, plp_type=1, plp_payload_type=3, plp_group_id=1, plp_cod=3, plp_mod=3, plp_rotation=1, plp_fec_type=1,
                plp_num_blocks_max=nb, frame_interval=1, time_il_length=1, time_il_type=0, plp_mode=1)]
    info = t2_tx.l1_post_bits(dict(frame_idx=1, l1_change_counter=0), plp, [dict(id=0, start=0, num_blocks=nb)])
    pre["l1_post_info_size"] = len(info)
    l1c = np.concatenate([t2_tx.l1_pre_cells(pre, 3), t2_tx.l1_post_cells(info, 1, lps, 4)])
    rng = np.random.Generator(np.random.PCG64(seed))
    cap = (m.c_p2 - 1840 - lps) + (m.n_data - m.l_fc) * m.c_data + m.l_fc * m.n_fc
    stream = (rng.integers(0, 16, cap) * 2 - 15 + 1j * (rng.integers(0, 16, cap) * 2 - 15)) * t2_tx.NORM[3]
    fr = t2_tx.build_frame(m, stream, lps, seed, snr_db=snr_db, phase=0.3, l1_cells=l1c)
    spec = np.fft.fftshift(np.fft.fft(fr.astype(np.complex128), axis=1), axes=1)
    return m, quantise(spec, GRID_SPEC), pre, plp


def sym_symbols(m):
    """The symbols of a frame that the fixture keeps whole: P2, the first odd and even data symbols, the last data symbol, FC."""
    last = m.len_frame - 1 - m.l_fc
    return [0, 1, 2, last] + ([m.len_frame - 1] if m.l_fc else [])


def ora_equalise(m, l, spec_c64):
    """The oracle on symbol l of mode m. P2 symbols use the tables the reference uses for them: extended carriers whatever the
    signal's mode (p2_symbol::init -> dvbt2_p2_parameters_init, dvbt2_definition.cpp:88-90)."""
    if l < m.n_p2:
        m = ol.ora_mode(m.fft_mode, 1, m.pilot_pattern, m.guard_interval_mode, m.papr_mode, m.n_data)
    return ol.ora_data_symbol(m, l, spec_c64)


# ------------------------------------------------------------------------------------------------ P1
def p1_stream(seed=8):
    """Noise, a P1 symbol (S1 = 0, S2 = 8: 16K SISO), noise -- on the 1/8192 grid. Returns (int16 [n][2], level_detect)."""
    rng = np.random.Generator(np.random.PCG64(seed))

    def noise(n, s):
        return (rng.integers(-1000, 1001, n) + 1j * rng.integers(-1000, 1001, n)) * (s / 1000.0)
    x = np.concatenate([noise(2700, 0.05), t2_tx.p1_symbol(0, 8) * 0.3 + noise(2048, 0.02), noise(3000, 0.05)])
    q = quantise(x, GRID_CELL)
    xc = dequantise(q, GRID_CELL)
    return q, float(np.float32(np.mean(np.abs(xc.real)) * np.mean(np.abs(xc.imag))))


# ------------------------------------------------------------------------------------------------ FEC-side cases
# name: (modulation, fec_type, code_rate, FEC blocks in the TI block, SNR dB, seed). Block counts > 32 so that one SIMD batch of
# the demapper / LDPC stage completes; cfg_a shows the 256-QAM wrap (every batch dropped), the others decode to TS.
FEC_CASES = {
    "cfg_b": (2, 0, 0, 64, 16.0, 1),          # 64-QAM, 16200, r=1/2: two batches
    "cfg_b_12": (2, 0, 0, 64, 12.0, 21),      # the same at 12 dB: BASELINE.md section 3's point for config 4, the one bench.py's config_4 legs run at
    "cfg_a": (3, 1, 3, 34, 21.0, 2),          # 256-QAM, 64800, r=3/4 at SURVEY 8d's 21 dB: the reference still drops every batch
    "cfg_c": (3, 1, 2, 33, 21.0, 3),          # 256-QAM, 64800, r=2/3 (demux_256_fec_size_normal_2_3), 21 dB
    "q16_n12": (1, 1, 0, 33, 10.0, 4),        # 16-QAM, 64800, r=1/2
    "qpsk_s34": (0, 0, 3, 40, 6.0, 5),        # QPSK, 16200, r=3/4
    "q64_n35": (2, 1, 1, 32, 15.0, 6),        # 64-QAM, 64800, r=3/5 (demux_64_fec_size_normal_code_3_5)
    # at the decoding threshold (found with the reference itself): some SIMD batches of the TI block decode in the last sweeps, some are
    # dropped -- where a one-step LLR difference could flip a batch, decode / drop equality is tested, not assumed
    "q64_s12_edge": (2, 0, 0, 160, 10.2, 11),   # 64-QAM, 16200, r=1/2: the reference decodes 2 of 5 batches (1 trial left each)
    "qpsk_n12_edge": (0, 1, 0, 64, 1.0, 12),    # QPSK, 64800, r=1/2: 1 of 2 (2 trials left)
    "q16_n35_edge": (1, 1, 1, 96, 7.65, 13),    # 16-QAM, 64800, r=3/5: 2 of 3 (2 trials left)
    "q64_n23_edge": (2, 1, 2, 96, 13.8, 14),    # 64-QAM, 64800, r=2/3: 2 of 3 (0 trials left: decoded by the 25th sweep)
}
FEC_L1_POST_SIZE = 200


def fec_case(name):
    """(cells of the TI block in transmission order on the 1/8192 grid as int16 [n][2], BBFRAMEs sent, TS packets sent, l1_post ints)."""
    mod, fec_type, code_rate, nb, snr, seed = FEC_CASES[name]
    cid = ol.code_id(fec_type, code_rate)
    k_bch = t2_tx.K_BCH[cid]
    ts = t2_tx.ts_packets(nb * (k_bch // 1496 + 1) + 8, seed)
    stream, frames, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts, nb)
    rng = np.random.Generator(np.random.PCG64(seed))
    amp = 10 ** (-snr / 20) * np.sqrt(1.5)                       # uniform integer noise of the same variance as AWGN at `snr`
    noise = (rng.integers(-4096, 4097, stream.size) + 1j * rng.integers(-4096, 4097, stream.size)) * (amp / 4096.0)
    q = quantise(stream + noise, GRID_CELL)
    cfg = dict(plp_cod=code_rate, plp_mod=mod, plp_rotation=1, plp_fec_type=fec_type, plp_num_blocks_max=nb, frame_interval=1,
               time_il_length=1, plp_type=1, plp_payload_type=3, plp_mode=1)
    l1 = ol.pack_l1_post([(cfg, dict(start=0, num_blocks=nb))])
    return q, frames, ts, l1


def fec_geometry(name):
    mod, fec_type, code_rate, nb, _, _ = FEC_CASES[name]
    n = 64800 if fec_type else 16200
    return mod, fec_type, code_rate, nb, n, n // (2 * (mod + 1)), ol.code_id(fec_type, code_rate)


def ora_fec_chain(name, cells_c64):
    """The oracle's restatement of the same chain on the same cells: TI block, LLRs (with the oracle's own scale), LDPC batches
    of 32 (trials left per batch, hard bits), descrambled BBFRAMEs of the batches that decoded."""
    mod, fec_type, code_rate, nb, n, cpf, cid = fec_geometry(name)
    t = ol.OraTi(cpf, nb)
    t.begin(nb)
    ti = np.zeros(nb * cpf, np.complex64)
    assert t.push(cells_c64, ti) == 1
    llr, sums, _ = ol.ora_demap(mod, fec_type, code_rate, 1, ti)
    batches = nb // 32
    trials, bits = [], []
    for b in range(batches):
        r, hard, _ = ol.ora_decode(cid, llr[32 * b:32 * b + 32])
        trials.append(r)
        bits.append(hard)
    bb = [ol.ora_bch_descramble(cid, bits[b]) for b in range(batches) if trials[b] >= 0]
    return dict(ti=ti, llr=llr[:32 * batches], scale=sums, trials=np.array(trials, np.int32),
                ldpc_bits=bits, bbframes=np.concatenate(bb) if bb else np.zeros((0, t2_tx.K_BCH[cid]), np.uint8))


# ------------------------------------------------------------------------------------------------ front loop (a1)
def front_case(seed=7, n=4000):
    rng = np.random.Generator(np.random.PCG64(seed))
    i16 = (rng.integers(-3000, 3000, n) + 37).astype(np.int16)
    q16 = (rng.integers(-2800, 2800, n) - 21).astype(np.int16)
    loops = dict(c1=0.013, c2=0.97, phase_est_filtered=1.3e-3, frequency_est_filtered=-2.1e-5)
    return i16, q16, loops


# ------------------------------------------------------------------------------------------------ whole receiver
RX_STREAM = dict(n_frames=12, seed=191, cfo_hz=0.0, l1_post_mod=1, plp=(2, 0, 0, 18.0))


def rx_stream():
    return t2_tx.rx_test_stream(RX_STREAM["n_frames"], RX_STREAM["seed"], RX_STREAM["cfo_hz"], None, RX_STREAM["l1_post_mod"],
                                RX_STREAM["plp"])


# ------------------------------------------------------------------------------------------------ whole receiver, loops closed under offsets
# What VERDICT r5 found missing: the path `drop_in` times -- 32K, every loop closed -- held to the reference's own dvbt2_demodulator on the
# same samples, under a carrier offset and a sample-rate offset. name: OFDM mode (ora_mode arguments), L1-post size, P1 S2 field, PLP
# (modulation, FEC type, code rate, SNR dB: modes that DECODE in the reference's arithmetic), T2 frames, seed, carrier offset of the
# recording in Hz, sample_rate handed to the demodulator's constructor minus the true 64e6 / 7 in Hz (a receiver clock that is off: what
# the sample-rate tracker must find, dvbt2_demodulator.cpp:430-439; float32 sample rates are 1 Hz = 0.11 ppm apart), samples per
# execute() call, frequency step of the emulated tuner in Hz.
# The tuner: the reference asks the SDR thread to move the local oscillator by P1's coarse estimate whenever it is 10 Hz or more
# (dvbt2_demodulator.cpp:291-305, rx_sdrplay.cpp:158-176) and goes no further until it is less. A recording has no tuner, so the harness
# emulates one: the move is applied to the recording itself (t2_tx.rx_offset_rotate) from the next buffer on, rounded to a multiple of
# `tuner_step` -- a synthesizer of finite resolution -- which leaves a residual of a few Hz for the frequency loop to pull in (with a perfect
# tuner the loops would idle). The fixture records the moves the REFERENCE asked for; the product is fed the very same samples.
RX_OFFSET_CASES = {
    "rx32k": dict(mode=(5, 1, 6, 4, 0, 59), lps=350, s2=10, plp=(2, 1, 2, 20.0), n_frames=17, seed=3201, cfo_hz=77.0, rate_off_hz=18.0,
                  buf=172032, tuner_step=12.0),
    "rx16k": dict(mode=(4, 1, 6, 0, 0, 24), lps=400, s2=8, plp=(2, 0, 0, 18.0), n_frames=16, seed=1601, cfo_hz=-64.0, rate_off_hz=-9.0,
                  buf=172032, tuner_step=12.0),
}


RX_OFFSET_TI_STEP = 4099          # every 4099th cell of a TI block goes into the fixture (a prime: walks through all symbols / FEC blocks)


def rx_offset_case(name):
    """(case dict, mode object, base I, base Q, TS packets sent per frame) -- the recording before the tuner (t2_tx.rx_offset_tuned)."""
    c = RX_OFFSET_CASES[name]
    m, i16, q16, sent = t2_tx.rx_offset_base(c["mode"], c["lps"], c["s2"], c["plp"], c["n_frames"], c["seed"])
    return c, m, i16, q16, sent


def rx_offset_sample_rate(c):
    return float(np.float32(64.0e6 / 7.0) + np.float32(c["rate_off_hz"]))


# ------------------------------------------------------------------------------------------------ LDPC stage input (256-QAM payload pin)
# The reference's own 256-QAM demapper never hands a decodable batch to its LDPC stage: the hard-decision SNR estimate saturates and
# the truncating int8 cast wraps the outer constellation points (llr_demapper.cpp:564-737; DESIGN.md section 3 has the command that
# shows 0 batches at 19.5 ... 24 dB). What CAN be pinned for configs 3 and 5 is everything from the LDPC stage's public slot on:
# ldpc_decoder::execute -> bch_decoder -> bb_de_header fed with CLAMPED LLRs (the quantize() rule the reference itself uses for QPSK,
# llr_demapper.cpp:770-776 = the product's saturate_llr extension). name: (modulation, fec_type, code_rate, frames, SNR dB, seed)
LDPC_IN_CASES = {
    "cfg_a": (3, 1, 3, 64, 22.0, 12),         # 256-QAM, 64800, r=3/4: two SIMD batches
    "cfg_c": (3, 1, 2, 64, 22.0, 13),         # 256-QAM, 64800, r=2/3
}


def saturated_llr(mod, fec_type, code_rate, ti_cells, precision):
    """Clamped LLRs of de-rotated, time-de-interleaved cells: the per-axis recursion of llr_demapper.cpp (x, |x| - 8d, ...) times
    `precision`, round to nearest even, clamp to [-128, 127] (quantize(), llr_demapper.cpp:770-776), scattered through the bit
    de-interleaver address table. float32 IEEE basic operations only (bit-reproducible anywhere); int8 [frames][fec_size]."""
    n = 64800 if fec_type else 16200
    bpc = 2 * (mod + 1)
    cpf = n // bpc
    c = np.ascontiguousarray(ti_cells, np.complex64)
    frames = c.size // cpf
    addr = ol.ora_bitdeint_address(mod, fec_type, code_rate).reshape(cpf, bpc)
    v = c.view(np.float32).reshape(frames, cpf, 2).copy()
    out = np.zeros((frames, n), np.int8)
    thr = np.float32(t2_tx.NORM32[mod] * np.float32(1 << mod))
    p = np.float32(precision)
    rows = np.arange(frames)[:, None]
    for lvl in range(mod + 1):
        q = np.clip(np.rint(v * p), -128, 127).astype(np.int8)
        for ax in range(2):
            out[rows, addr[None, :, 2 * lvl + ax]] = q[:, :, ax]
        v = np.abs(v) - thr
        thr = np.float32(thr * np.float32(0.5))
    return out


def ldpc_in_case(name):
    """(int8 LLRs [frames][fec_size], BBFRAMEs sent, TS packets sent, l1_post ints): frames FEC blocks of one TI block, sent with
    uniform integer noise of the variance of AWGN at the case's SNR, time-de-interleaved and de-rotated by the oracle, clamped LLRs
    with the oracle's own scale estimate."""
    mod, fec_type, code_rate, nb, snr, seed = LDPC_IN_CASES[name]
    cid = ol.code_id(fec_type, code_rate)
    k_bch = t2_tx.K_BCH[cid]
    n = 64800 if fec_type else 16200
    cpf = n // (2 * (mod + 1))
    ts = t2_tx.ts_packets(nb * (k_bch // 1496 + 1) + 8, seed)
    stream, frames, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts, nb)
    rng = np.random.Generator(np.random.PCG64(seed))
    amp = 10 ** (-snr / 20) * np.sqrt(1.5)
    noise = (rng.integers(-4096, 4097, stream.size) + 1j * rng.integers(-4096, 4097, stream.size)) * (amp / 4096.0)
    cells = dequantise(quantise(stream + noise, GRID_CELL), GRID_CELL)
    t = ol.OraTi(cpf, nb)
    t.begin(nb)
    ti = np.zeros(nb * cpf, np.complex64)
    assert t.push(cells, ti) == 1
    _, sums, derot = ol.ora_demap(mod, fec_type, code_rate, 1, ti)
    llr = saturated_llr(mod, fec_type, code_rate, derot, sums[2])
    cfg = dict(plp_cod=code_rate, plp_mod=mod, plp_rotation=1, plp_fec_type=fec_type, plp_num_blocks_max=nb, frame_interval=1,
               time_il_length=1, plp_type=1, plp_payload_type=3, plp_mode=1)
    return llr, frames, ts, ol.pack_l1_post([(cfg, dict(start=0, num_blocks=nb))])


# ------------------------------------------------------------------------------------------------ SIMD batches across T2 frames
# The reference fills its 32-frame LLR buffers across TI blocks and T2 frames (llr_demapper.cpp:742-764: `static int blocks`) and
# never decodes a short batch. name: OFDM mode, l1_post_size, (modulation, fec_type, code_rate), FEC blocks per T2 frame (not a
# multiple of 32), T2 frames, cell-level SNR for the reference run, seed, P1 S2 field, SNR of the int16 I/Q stream.
CARRY_CASES = {
    "b40x3": dict(mode=(4, 1, 6, 4, 0, 8), lps=200, plp=(2, 0, 0), nb=40, frames=3, snr=16.0, seed=21, s2=8, iq_snr=20.0),
}


def carry_geometry(name):
    c = CARRY_CASES[name]
    mod, fec_type, code_rate = c["plp"]
    n = 64800 if fec_type else 16200
    return c, mod, fec_type, code_rate, n, n // (2 * (mod + 1)), ol.code_id(fec_type, code_rate)


def carry_payload(name):
    """Per T2 frame: (interleaved cell stream of the frame's one TI block, its BBFRAMEs [nb][k_bch], its TS packets)."""
    c, mod, fec_type, code_rate, n, cpf, cid = carry_geometry(name)
    k_bch = t2_tx.K_BCH[cid]
    per = c["nb"] * (k_bch // 1496 + 1)
    ts = t2_tx.ts_packets(c["frames"] * per + 8, c["seed"])
    out = []
    for f in range(c["frames"]):
        stream, frames, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts[f * per:(f + 1) * per], c["nb"])
        out.append((stream, frames, ts[f * per:(f + 1) * per]))
    return out


def carry_case(name):
    """What the reference's FEC chain is fed: per T2 frame the TI block's cells on the 1/8192 grid with uniform integer noise (int16
    [nb * cpf][2]); plus the BBFRAMEs sent and the l1_post ints."""
    c, mod, fec_type, code_rate, n, cpf, cid = carry_geometry(name)
    cells, sent = [], []
    for f, (stream, frames, _) in enumerate(carry_payload(name)):
        rng = np.random.Generator(np.random.PCG64(c["seed"] + 100 + f))
        amp = 10 ** (-c["snr"] / 20) * np.sqrt(1.5)
        noise = (rng.integers(-4096, 4097, stream.size) + 1j * rng.integers(-4096, 4097, stream.size)) * (amp / 4096.0)
        cells.append(quantise(stream + noise, GRID_CELL))
        sent.append(frames)
    cfg = dict(plp_cod=code_rate, plp_mod=mod, plp_rotation=1, plp_fec_type=fec_type, plp_num_blocks_max=c["nb"], frame_interval=1,
               time_il_length=1, plp_type=1, plp_payload_type=3, plp_mode=1)
    return cells, np.concatenate(sent), ol.pack_l1_post([(cfg, dict(start=0, num_blocks=c["nb"]))])


l1_cells = t2_tx.l1_cells


def carry_iq(name, l1_variant=None):
    """The same payload as whole T2 frames at the tuner interface (int16 I/Q, P1 + cyclic prefixes + AWGN), with real L1 signalling in
    every P2 symbol: what t2gpu_rx is fed. l1_variant: {frame: dict(num_blocks=.. | spoil_post=True)} for the L1-gating tests."""
    c, mod, fec_type, code_rate, n, cpf, cid = carry_geometry(name)
    m = ol.ora_mode(*c["mode"])
    guard = {0: m.fft_size // 32, 1: m.fft_size // 16, 2: m.fft_size // 8, 3: m.fft_size // 4, 4: m.fft_size // 128}[c["mode"][3]]
    frames = [t2_tx.build_frame(m, stream, c["lps"], c["seed"] + 200 + f, snr_db=None, phase=0.0,
                                l1_cells=l1_cells(c["mode"], c["lps"], mod, fec_type, code_rate, c["nb"], frame_idx=f, **((l1_variant or {}).get(f, {}))))
              for f, (stream, _, _) in enumerate(carry_payload(name))]
    i16, q16, flen = t2_tx.iq_stream(frames, guard, c["s2"], c["iq_snr"], c["seed"])
    return i16.reshape(c["frames"], flen), q16.reshape(c["frames"], flen)
