"""GPU tier: the HIP path (through the C ABI) against THE REFERENCE's own outputs -- the fixtures tests/golden/t2sym_golden.npz,
t2fec_golden.npz, t2rx_golden.npz that tests/golden/make_t2_golden.py produced by running the reference's compiled src/DVB_T2
classes. No oracle in between: what the reference computed is the expected value. Tolerances are the ones tests/test_ref_pins.py
establishes for the oracle, for the same stated reasons."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
import ref_cases as rc
import t2_tx
from test_ref_pins import _load, sub, u32, full_spectrum

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def gsym():
    return _load("t2sym_golden.npz")


@pytest.fixture(scope="module")
def gfec():
    return _load("t2fec_golden.npz")


@pytest.fixture(scope="module")
def grx():
    return _load("t2rx_golden.npz")


@pytest.fixture(scope="module")
def gbatch():
    return _load("t2batch_golden.npz")


def dev(torch, x):
    return torch.from_numpy(np.ascontiguousarray(x).view(np.float32).reshape(x.shape + (2,))).cuda()


def c64(t):
    return np.ascontiguousarray(t.cpu().numpy()).view(np.complex64).reshape(t.shape[:-1])


def close_cells(got, want, what):
    bad = np.nonzero((u32(got) != u32(want)).reshape(-1, 2).any(axis=1))[0]
    assert bad.size <= max(1, got.size // 2000), (what, bad.size)
    if bad.size:
        assert np.abs(got[bad] - want[bad]).max() < 3e-4, what


@pytest.mark.parametrize("name", list(rc.SYM_MODES))
def test_equalisers_against_the_reference(torch_cuda, gsym, name):
    """The equaliser (eq_split_kernel) with the data / P2 / frame-closing tables on the spectra the reference equalised: data symbols bit-exact
    (cells and both sync outputs); P2 and FC cells bit-exact except <= 0.05 % within 3e-4, phase within 1e-6, sample-rate offset
    within 2.5e-4 (see tests/test_ref_pins.py::test_equalisers_equal_the_reference). The P2 symbol is read with the extended-carrier
    tables whatever the mode, as the reference reads it."""
    torch = torch_cuda
    import sdr_receiver_dvb_t2_amd as pkg
    g = sub(gsym, "sym", name)
    mode = tuple(int(v) for v in g["mode"])
    m = ol.ora_mode(*mode)
    P1 = dict(zip(ol.DVBT2_FIELDS, (int(v) for v in g["params_p1"])))
    o = pkg.t2_ofdm(*mode, max_symbols=4)
    spec = full_spectrum(m, g["spec_0"], P1["l_nulls"], m.fft_size)
    cells, sync = o.eq_p2_dev(dev(torch, spec[None]))
    close_cells(c64(cells)[0], g["p2_cells"], "P2")
    s = sync.cpu().numpy()[0]                                                    # (phase_offset, sample_rate_offset)
    assert abs(s[1] - g["p2_sync"][0]) <= 2.5e-4 and abs(s[0] - g["p2_sync"][1]) <= 1e-6
    ls = [int(v) for v in g["symbols"][1:4]]
    specs = np.stack([full_spectrum(m, g["spec_%d" % l], m.l_nulls, m.fft_size) for l in ls])
    cells, sync = o.eq_data_dev(dev(torch, specs), torch.tensor(ls, dtype=torch.int32, device="cuda"))
    got, s = c64(cells), sync.cpu().numpy()
    for k, l in enumerate(ls):
        assert np.array_equal(u32(got[k]), u32(g["data_cells_%d" % l])), l
        assert s[k][1] == g["data_sync_%d" % l][0] and s[k][0] == g["data_sync_%d" % l][1], l
    if m.l_fc:
        l = m.len_frame - 1
        spec = full_spectrum(m, g["spec_%d" % l], m.l_nulls, m.fft_size)
        cells, sync = o.eq_fc_dev(dev(torch, spec[None]))
        close_cells(c64(cells)[0], g["fc_cells"], "FC")
        s = sync.cpu().numpy()[0]
        assert abs(s[1] - g["fc_sync"][0]) <= 2.5e-4 and abs(s[0] - g["fc_sync"][1]) <= 1e-6
    o.close()


@pytest.mark.parametrize("name", list(rc.SYM_MODES))
def test_host_tables_against_the_reference(built, gsym, name):
    """The product's table builders (csrc/ofdm_tables.cpp behind t2gpu_table_*) against the reference's pilot_generator and
    address_freq_deinterleaver: identical. (Host code; runs wherever the library loads.)"""
    import sdr_receiver_dvb_t2_amd as pkg
    l = pkg.lib()
    g = sub(gsym, "sym", name)
    mode = tuple(int(v) for v in g["mode"])
    m = ol.ora_mode(*mode)

    def carriers(md, idx):
        info = (ctypes.c_int * 12)()
        assert l.t2gpu_ofdm_mode_info(*md, info) == 0
        k = info[1]
        mp, rf = np.zeros(k, np.uint8), np.zeros(k, np.float32)
        assert l.t2gpu_table_symbol_carriers(*md, idx, mp.ctypes.data, rf.ctypes.data) == k
        return mp.astype(np.int8), rf

    def deint(md, kind, n):
        he, ho = np.zeros(n, np.int32), np.zeros(n, np.int32)
        assert l.t2gpu_table_freq_deint(*md, kind, he.ctypes.data, ho.ctypes.data) == n
        return he, ho
    ext = (mode[0], 1) + mode[2:]
    mp, rf = carriers(ext, 0)
    assert np.array_equal(mp, g["p2_map"]) and np.array_equal(u32(rf), u32(g["p2_refer"]))
    he, ho = deint(ext, 0, m.c_p2)
    assert np.array_equal(he, g["p2_h_even"]) and np.array_equal(ho, g["p2_h_odd"])
    maps, refs = zip(*[carriers(mode, idx) for idx in range(m.n_p2, m.len_frame - m.l_fc)])
    assert np.array_equal(rc.crc_rows(np.stack(maps)), g["data_map_crc"]) and np.array_equal(rc.crc_rows(np.stack(refs)), g["data_refer_crc"])
    he, ho = deint(mode, 1, m.c_data)
    assert np.array_equal(he, g["data_h_even"]) and np.array_equal(ho, g["data_h_odd"])
    if m.l_fc:
        mp, rf = carriers(mode, m.len_frame - 1)
        assert np.array_equal(mp, g["fc_map"]) and np.array_equal(u32(rf), u32(g["fc_refer"]))
        he, ho = deint(mode, 2, m.n_fc)
        assert np.array_equal(he, g["fc_h_even"]) and np.array_equal(ho, g["fc_h_odd"])


@pytest.mark.parametrize("tag,splits", [("whole", None), ("split", [1000, 1777, 2500, None])])
def test_p1_against_the_reference(torch_cuda, gsym, tag, splits):
    """K-p1 fed as the reference's p1_symbol was: the same calls detect, consume the same samples, leave the same idx_buffer_sym and
    decode the same S1 / S2; coarse_freq_offset within 0.5 Hz."""
    from sdr_receiver_dvb_t2_amd import p1
    g = sub(gsym, "p1", "p1")
    x = rc.dequantise(g["x"], rc.GRID_CELL)
    splits = [len(x)] if splits is None else splits[:-1] + [len(x) - sum(splits[:-1])]
    det = p1.p1_symbol(max_samples=len(x))
    rows, pos = [], 0
    for n in splits:
        piece, consume = x[pos:pos + n], 0
        while consume < n:
            hit, consume, r = det.execute(piece, consume, True, float(g["level"]))
            rows.append([pos, int(hit), consume, r.idx_buffer_sym, r.p1_decoded, r.preamble, r.fft_mode, int(round(r.coarse_freq_offset * 1000))])
        pos += n
    rows, want = np.array(rows, np.int64), g["p1_" + tag]
    assert rows.shape[0] == want.shape[0] and np.array_equal(rows[:, :5], want[:, :5])
    hit = want[:, 1] == 1
    assert np.array_equal(rows[hit, 5:7], want[hit, 5:7]) and np.abs(rows[hit, 7] - want[hit, 8]).max() <= 500
    det.close()


@pytest.mark.parametrize("name", list(rc.FEC_CASES))
def test_fec_chain_against_the_reference(torch_cuda, gfec, name):
    """K-ti -> K-snr + K-demap -> K-ldpc -> K-descramble on the cells the reference's stage objects processed:
    TI block bit-exact (per-FEC-block CRC-32); LLRs of the frames the fixture holds whole equal except on <= 1e-4 of the positions
    (north_star's "stated float tolerance on pre-FEC LLRs"; the device forms the scale 8*norm*sum_s/sum_e from the reference's own
    sequential float sums, bit-equal to the oracle and to the reference's strict build -- the residue is what -Ofast does to the
    reference binary's de-rotation, the same bound tests/test_ref_pins.py holds the oracle to); the same SIMD batches decode --
    including the four cases at the decoding threshold, where some batches of the TI block decode in the last sweeps and others are
    dropped; hard bits and descrambled BBFRAMEs bit-exact; the
    host de-framer turns them into the reference's TS bytes."""
    torch = torch_cuda
    import sdr_receiver_dvb_t2_amd as pkg
    g = sub(gfec, "fec", name)
    q, frames, ts, l1 = rc.fec_case(name)
    assert rc.sha(q) == str(g["in_sha"])
    mod, fec_type, code_rate, nb, n, cpf, cid = rc.fec_geometry(name)
    cells = rc.dequantise(q, rc.GRID_CELL)
    ti = pkg.time_deinterleaver(mod, fec_type, nb)
    assert ti.l1_dyn(nb) == nb * cpf
    out = torch.zeros((nb * cpf, 2), dtype=torch.float32, device="cuda")
    assert ti.execute_dev(dev(torch, cells), out)
    torch.cuda.synchronize()
    tic = c64(out)
    assert np.array_equal(rc.crc_rows(tic.reshape(nb, cpf)), g["ti_crc"])
    dm = pkg.llr_demapper(mod, fec_type, code_rate, 1, max_cells=nb * cpf)
    llr, sums = dm.execute_dev(out)
    L = llr.cpu().numpy().reshape(-1, n)
    batches = nb // 32
    for row, want in ((0, g["llr_first"]), (32 * batches - 1, g["llr_last"])):
        d = L[row].astype(np.int32) - want.astype(np.int32)
        d = np.minimum(np.abs(d), 256 - np.abs(d))
        assert d.max() <= 1 and np.count_nonzero(d) <= max(1, n // 10000), (row, np.count_nonzero(d))
    # against the STRICT build of the reference (every float operation in the order the reference writes it, oracle/Makefile): every
    # LLR of every frame the reference emitted, no tolerance (per-row CRC-32 of all rows, first and last row whole)
    assert np.array_equal(rc.crc_rows(L[:32 * batches]), g["llr_crc_strict"])
    assert np.array_equal(L[0], g["llr_first_strict"]) and np.array_equal(L[32 * batches - 1], g["llr_last_strict"])
    dec = pkg.ldpc_decoder(fec_type, code_rate, max_frames=32 * batches)
    bits, trials = dec.execute_dev(llr[:32 * batches].contiguous())
    torch.cuda.synchronize()
    t = trials.cpu().numpy()
    assert int((t >= 0).sum()) == int(g["ldpc_batches"])
    if mod == 3:
        assert (t < 0).all()                                                     # the reference drops every 256-QAM batch too
        return
    keep = np.repeat(t >= 0, 32)                                                 # frames of the batches that decoded (the reference emits only those)
    assert keep.any()
    B = bits.cpu().numpy()[keep]
    assert np.array_equal(rc.crc_rows(B), g["ldpc_crc"]) and np.array_equal(np.packbits(B[0]), g["ldpc_first"])
    bch = pkg.bch_decoder(fec_type, code_rate)
    D = bch.execute_dev(bits).cpu().numpy()[keep]
    assert np.array_equal(rc.crc_rows(D), g["bb_crc"]) and np.array_equal(np.packbits(D[0]), g["bb_first"])
    l = pkg.lib()
    h = l.t2gpu_bbdh_create(0)
    got = []
    for fr in D:
        o = np.zeros(fr.size // 8 + 400, np.uint8)
        k = l.t2gpu_bbdh_execute(h, 0, fr.size, np.ascontiguousarray(fr).ctypes.data, o.ctypes.data, o.size, None)
        if k > 0:
            got.append(o[:k])
    l.t2gpu_bbdh_destroy(h)
    assert np.array_equal(np.concatenate(got), g["ts"])
    for obj in (ti, dm, dec):
        obj.close()


def deframe_packed(rows, k_bch, need_plp=0):
    """packed BBFRAME rows -> TS bytes through t2gpu_bbdh_execute_packed"""
    import sdr_receiver_dvb_t2_amd as pkg
    l = pkg.lib()
    h = l.t2gpu_bbdh_create(need_plp)
    got = []
    for r in rows:
        o = np.zeros(k_bch // 8 + 400, np.uint8)
        k = l.t2gpu_bbdh_execute_packed(h, need_plp, k_bch, np.ascontiguousarray(r).ctypes.data, o.ctypes.data, o.size, None)
        if k > 0:
            got.append(o[:k])
    l.t2gpu_bbdh_destroy(h)
    return np.concatenate(got) if got else np.zeros(0, np.uint8)


@pytest.mark.parametrize("name", list(rc.LDPC_IN_CASES))
def test_ldpc_stage_256qam_payload_against_the_reference(torch_cuda, gbatch, name):
    """BASELINE configs 3 and 5 behind the demapper, against reference-built code: the clamped 256-QAM LLRs the reference's
    ldpc_decoder::execute slot was fed (ldpc_decoder.h:90; two SIMD batches) through K-ldpc -> K-descramble / K-descramble-pack ->
    t2gpu_bbdh: every hard bit, every descrambled BBFRAME bit and every TS byte equal what the reference's ldpc_decoder ->
    bch_decoder -> bb_de_header emitted. Packed and bit-per-byte forms agree."""
    torch = torch_cuda
    import sdr_receiver_dvb_t2_amd as pkg
    g = sub(gbatch, "ldpc_in", name)
    llr, frames, ts, l1 = rc.ldpc_in_case(name)
    assert rc.sha(llr) == str(g["in_sha"]), "regenerated LLRs differ from the ones the reference saw"
    mod, fec_type, code_rate, nb, snr, seed = rc.LDPC_IN_CASES[name]
    k_bch = t2_tx.K_BCH[ol.code_id(fec_type, code_rate)]
    dec = pkg.ldpc_decoder(fec_type, code_rate, max_frames=nb)
    bits, trials = dec.execute_dev(torch.from_numpy(llr).cuda())
    torch.cuda.synchronize()
    assert dec.status() == 0 and (trials.cpu().numpy() >= 0).all() and int(g["ldpc_batches"]) == nb // 32
    B = bits.cpu().numpy()
    assert np.array_equal(rc.crc_rows(B), g["ldpc_crc"]) and np.array_equal(np.packbits(B[0]), g["ldpc_first"])
    bch = pkg.bch_decoder(fec_type, code_rate)
    D = bch.execute_dev(bits).cpu().numpy()
    assert np.array_equal(rc.crc_rows(D), g["bb_crc"]) and np.array_equal(np.packbits(D[0]), g["bb_first"])
    P = bch.execute_packed_dev(bits).cpu().numpy()
    assert P.shape == (nb, k_bch // 8) and np.array_equal(P, np.packbits(D, axis=1))
    got = deframe_packed(P, k_bch)
    assert got.size == int(g["ts_len"]) and np.array_equal(rc.crc_rows(got[:got.size // 188 * 188].reshape(-1, 188)), g["ts_packet_crc"])
    assert np.array_equal(got[:752], g["ts_head"]) and np.array_equal(got[-376:], g["ts_tail"])
    dec.close()


@pytest.mark.parametrize("name", list(rc.CARRY_CASES))
def test_batch_receiver_carries_simd_batches_across_calls_like_the_reference(torch_cuda, gbatch, name):
    """t2gpu_rx against the reference's FEC chain over three T2 frames of 40 FEC blocks (llr_demapper.cpp:742-764: the 32-frame LLR
    buffer fills across frames, a short batch is never decoded). The reference emitted one batch after each frame (FEC blocks 0-31,
    32-63, 64-95) and kept 24. Three calls of one frame each: the same counts, the same BBFRAMEs in the same batches, and -- through
    the library's own host end (worker thread: L1 parse of every P2 symbol + de-framer) -- the reference's TS file byte for byte. One
    call over all three frames gives the same rows. The flush (an addition) then delivers the 24 frames the reference keeps."""
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    g = sub(gbatch, "carry", name)
    c, mod, fec_type, code_rate, n, cpf, cid = rc.carry_geometry(name)
    nb, nf = c["nb"], c["frames"]
    k_bch = t2_tx.K_BCH[cid]
    _, sent, _ = rc.carry_case(name)
    assert np.array_equal(rc.crc_rows(sent), g["sent_bbframes_crc"])
    i16, q16 = rc.carry_iq(name)
    d_i, d_q = torch.from_numpy(i16).cuda(), torch.from_numpy(q16).cuda()
    want_counts = np.diff(np.concatenate([[0], g["emitted_after_frame"][:, 2]]))          # BBFRAMEs the reference emitted per frame
    rx = t2_rx(*c["mode"], c["lps"], mod, fec_type, code_rate, 1, nb, max_frames=nf)
    rx.ts_enable(0, l1_check=True)
    rows, level = [], 0.0
    for f in range(nf):
        count = rx.execute_dev(d_i[f], d_q[f], 1, level_detect=level, first_call=(f == 0))
        assert count == want_counts[f] and rx.carry == (f + 1) * nb - int(g["emitted_after_frame"][f, 2])
        r, t = rx.fetch_packed(count)
        assert (t >= 0).all()
        rows.append(r)
        level = rx.results(1)["level_detect"]
    rows = np.concatenate(rows)
    bits = np.unpackbits(rows, axis=1)
    assert np.array_equal(rc.crc_rows(bits), g["bb_crc"])                                  # = the reference's bit_descramble output
    ts = rx.ts_read(wait_all=True)
    assert np.array_equal(ts, g["ts"])                                                     # = the reference's TS file
    n = rx.ts_counters()
    assert n["t2_frames"] == nf and n["fec_frames"] == len(rows) and n["ts_bytes"] == ts.size
    assert n["l1_pre_crc_errors"] == n["l1_post_crc_errors"] == n["l1_mismatches"] == n["fec_frames_dropped_l1"] == n["fec_frames_dropped_ldpc"] == 0
    # end of stream: the frames the reference never decodes
    rest = rx.flush_dev()
    assert rest == nf * nb - len(rows) and rx.carry == 0
    r2, t2 = rx.fetch_packed(int(want_counts[-1]) + rest)
    assert (t2 >= 0).all() and np.array_equal(r2[:int(want_counts[-1])], rows[-int(want_counts[-1]):])
    tail = np.unpackbits(r2[int(want_counts[-1]):], axis=1)
    assert np.array_equal(rc.crc_rows(tail), g["sent_bbframes_crc"][len(rows):])
    assert rx.ts_read(wait_all=True).size > 0
    rx.close()
    # one call over the whole buffer: the same batches
    one = t2_rx(*c["mode"], c["lps"], mod, fec_type, code_rate, 1, nb, max_frames=nf)
    count = one.execute_dev(d_i.reshape(-1), d_q.reshape(-1), nf, first_call=True)
    assert count == len(rows) and one.carry == nf * nb - len(rows)
    r1, t1 = one.fetch_packed(count)
    assert (t1 >= 0).all() and np.array_equal(r1, rows)
    assert np.array_equal(deframe_packed(r1, k_bch), g["ts"])
    one.close()


def test_batch_receiver_gates_frames_on_their_l1_signalling(torch_cuda):
    """The host end's per-frame L1 parse (p2_symbol.cpp:301-718 per P2 symbol): a T2 frame whose L1-post fails its CRC-32 and one
    whose dynamic PLP_NUM_BLOCKS differs from the configuration have their BBFRAMEs withheld and counted; the others flow."""
    torch = torch_cuda
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    name = "b40x3"
    c, mod, fec_type, code_rate, n, cpf, cid = rc.carry_geometry(name)
    nb, nf = c["nb"], c["frames"]
    i16, q16 = rc.carry_iq(name, l1_variant={1: dict(spoil_post=True)})
    rx = t2_rx(*c["mode"], c["lps"], mod, fec_type, code_rate, 1, nb, max_frames=nf)
    rx.ts_enable(0, l1_check=True)
    count = rx.execute_dev(torch.from_numpy(i16).cuda().reshape(-1), torch.from_numpy(q16).cuda().reshape(-1), nf, first_call=True, flush=True)
    assert count == nf * nb
    ts = rx.ts_read(wait_all=True)
    k = rx.ts_counters()
    assert k["t2_frames"] == nf and k["l1_post_crc_errors"] == 1 and k["l1_pre_crc_errors"] == 0 and k["l1_mismatches"] == 0
    assert k["fec_frames"] == nf * nb and k["fec_frames_dropped_l1"] == nb and k["ts_bytes"] == ts.size
    rx.close()
    i16, q16 = rc.carry_iq(name, l1_variant={2: dict(num_blocks=nb - 1)})
    rx = t2_rx(*c["mode"], c["lps"], mod, fec_type, code_rate, 1, nb, max_frames=nf)
    rx.ts_enable(0, l1_check=True)
    rx.execute_dev(torch.from_numpy(i16).cuda().reshape(-1), torch.from_numpy(q16).cuda().reshape(-1), nf, first_call=True, flush=True)
    k = rx.ts_counters()
    assert k["l1_mismatches"] == 1 and k["l1_post_crc_errors"] == 0 and k["fec_frames_dropped_l1"] == nb
    rx.close()


def test_front_loop_against_the_reference(torch_cuda, grx):
    """K-front on the chunk the reference's dvbt2_demodulator::execute processed with the same loop values: de-rotated stream within
    2e-6 (the dc averager is evaluated as a scan in double on the device), c1 / c2 / level within 1e-4 relative, NCO phases
    bit-exact; the Farrow + decimator output behind it within 3e-6."""
    from sdr_receiver_dvb_t2_amd import front
    g = sub(grx, "front", "front")
    i16, q16, loops = rc.front_case()
    assert rc.sha(np.stack([i16, q16])) == str(g["in_sha"])
    # both forms of the front end: the five launches (whose de-rotated stream is in memory to be looked at) and, for a chunk of this
    # size, the one launch with one pass per workgroup that t2gpu_demod_execute's symbol-sized calls take (its de-rotated samples never
    # leave the workgroups: pinned through the decimated stream behind them)
    for one_launch in (0, 1):
        fe = front.front_end(max_samples=len(i16))
        assert fe._l.t2gpu_front_set_chain(fe.h, one_launch) == 0
        fe.set_iq(loops["c1"], loops["c2"])
        got, _ = fe.execute(i16, q16, [len(i16)], [loops["phase_est_filtered"]], [loops["frequency_est_filtered"]])
        if not one_launch:
            der = fe.debug_stream(0, len(i16))
            assert np.abs(der - g["derotated"]).max() < 2e-6
        assert np.abs(got[:len(i16)] - g["decimated"][:len(got)]).max() < 3e-6
        st, want = fe.state(), dict(zip(ol.RX_STATE, g["state"]))
        assert np.float32(st["phase_nco"]) == np.float32(want["phase_nco"]) and np.float32(st["frequency_nco"]) == np.float32(want["frequency_nco"])
        for k in ("c1", "c2", "level_detect"):
            assert abs(st[k] - want[k]) <= 1e-4 * abs(want[k]), k
        fe.close()


def test_whole_receiver_against_the_reference(built, grx, tmp_path):
    """int16 I/Q -> TS through t2::dvbt2_demodulator and the stage classes in a plain C++ process (tests/cpp/stage_mirror_test.cpp
    rx) on the stream the REFERENCE's dvbt2_demodulator decoded for the fixture: same acquisition outcome (guard interval found,
    L1 parsed, de-interleaver started), and the TS packets are the reference's, packet for packet, wherever both produced output
    -- every packet of every frame behind the reference's own acquisition frame, none missing."""
    import zlib
    g = sub(grx, "rx", "rx")
    exe = str(tmp_path / "stage_mirror_test")
    pkg = os.path.join(ol.ROOT, "sdr_receiver_dvb_t2_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ol.ROOT, "include"),
                           os.path.join(ol.ROOT, "tests", "cpp", "stage_mirror_test.cpp"), "-L" + pkg, "-lt2gpu", "-Wl,-rpath," + pkg, "-o", exe])
    m, i16, q16, buf, marks = rc.rx_stream()
    i16.tofile(tmp_path / "i.s16")
    q16.tofile(tmp_path / "q.s16")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([exe, "rx", str(tmp_path / "i.s16"), str(tmp_path / "q.s16"), str(tmp_path / "out.ts"), str(buf), "0", str(tmp_path / "log.txt")],
                       env=env, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    log = open(tmp_path / "log.txt").read()
    last = [ln for ln in log.splitlines() if ln.startswith("buf ")][-1].split()
    last = dict(zip(last[2::2], last[3::2]))
    assert last["init"] == "1" and last["deint"] == "1" and last["crc"] == "1" and int(last["gi"]) == 512
    ts = np.fromfile(tmp_path / "out.ts", np.uint8)
    mine = set(int(c) for c in rc.crc_rows(ts[:ts.size // 188 * 188].reshape(-1, 188)))
    ref = [int(c) for c in g["ts_packet_crc"]]
    both = [c for c in ref if c in mine]
    print("whole receiver: %d of the reference's %d packets recovered (%d produced)" % (len(both), len(ref), len(mine)))
    per_frame = len(ref) // max(1, len(g["frames_found"]))                    # packets of one T2 frame (the reference emitted whole frames)
    missing = [c for c in ref[per_frame:] if c not in mine]                   # everything behind the reference's first (acquisition) frame
    assert not missing and len(both) >= len(ref) - per_frame, (len(missing), len(both), len(ref), len(mine))
    # and frame for frame: every frame the reference recovered after frame 5 is recovered here
    found = [f for f in range(len(marks)) if ts.tobytes().find(marks[f]) >= 0]
    assert set(int(f) for f in g["frames_found"] if f >= 6) <= set(found), (found, g["frames_found"])


# ------------------------------------------------------------------------------------------------ whole receiver, loops closed under offsets
def _build_stage_driver(tmp_path):
    exe = str(tmp_path / "stage_mirror_test")
    pkg = os.path.join(ol.ROOT, "sdr_receiver_dvb_t2_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ol.ROOT, "include"),
                           os.path.join(ol.ROOT, "tests", "cpp", "stage_mirror_test.cpp"), "-L" + pkg, "-lt2gpu", "-Wl,-rpath," + pkg, "-o", exe])
    return exe


# What the product may differ by from the reference's trajectory, and why (measured on an MI355X in round 6, `pytest -s` prints them):
#   The loop FILTERS are the reference's float operations bit for bit (tests/test_front_gpu.py); what enters them is not: the guard
#   correlation is a double-precision tree sum here and a sequential float sum there (2e-5 relative), the spectrum comes from a different FFT
#   (2e-5 rms against FFTW), the dc averager is a scan in double (2e-6). A symbol's raw estimates therefore agree with the reference's to
#   float precision at the start (|d phase_est| 2e-7 rad on the first symbols) and drift apart slowly: a difference of 1e-9 rad / sample in
#   frequency_est_filtered is 3e-5 rad of phase per 32K symbol. Two parts of the reference turn such last-digit differences into discrete
#   ones: the sample-rate tracker steps by 8e-9 on the SIGN of the difference of two successive noisy estimates (dvbt2_demodulator.cpp:
#   430-439) and never corrects -- it walks -- and the equaliser's phase unwrapping (data_symbol.cpp:189-191) is discontinuous at pi. So the
#   two receivers are held to each other TIGHTLY over the first RXOFF_HEAD tracked symbols (same chunk lengths, same tracker steps, loop
#   values to 1e-4) -- which is what shows they are the same function of the samples -- and LOOSELY afterwards (both stay locked on the same
#   carrier and clock; the tracker's walk may end tens of steps apart), and exactly again behind the FEC: the BBFRAMEs and the transport
#   stream are the reference's, packet for packet.
RXOFF_HEAD = 100
RXOFF_TOL = dict(
    head=dict(chunk=0, rate_steps=0, phase=2.0e-4, freq_rel=1.0e-3, phase_est=3.0e-3),     # measured: 0, 0, 5.0e-5, 1.9e-4, 6.0e-4 (rx32k)
    chunk=4,               # input samples of a whole-symbol chunk (nearbyint of est_chunk * arbitrary_resample * 2: moves with the tracker's value); measured 2
    p1_samples=8,          # ... of the chunk that completes a frame's P2 symbol: it starts where the P1 detector put the frame, and the
                           # detector's correlation peak is flat-topped (recursive float running sums there, sliding sums here); measured 0
    phase=1.0,             # |phase_est_filtered - reference| in rad over the whole stream; measured 0.32 (rx32k), 0.039 (rx16k)
    freq_rel=0.1,          # |frequency_est_filtered - reference| relative to the tracked value (41 Hz / -40 Hz); measured 0.031, 0.012
    rate_steps=150,        # |sample_rate_est_filtered - reference| in the tracker's steps of 8e-9; measured 68, 8
    ti_cells_first=2.0e-3, # de-interleaved cells of the FIRST TI block against the reference's (unit-power constellation): measured 3.1e-4, 2.3e-4
    ti_cells=0.25,         # ... of every TI block: once the trackers have walked apart (68 steps = 5e-7 = one sample by the end of a 32K frame)
                           # the FFT windows sit a sample apart and the pilot interpolation leaves different residues; measured 7.6e-2, at a
                           # noise of 1e-1 (20 dB) -- what counts there is the LDPC's output, which is identical
)


@pytest.mark.parametrize("name", list(rc.RX_OFFSET_CASES))
def test_closed_loop_trajectory_against_the_reference(built, grx, tmp_path, name):
    """VERDICT r5's parity hole: the slot-shaped path with every loop closed -- 32K (and a 16K twin), the loops on the device and on the host
    -- against what the REFERENCE's dvbt2_demodulator::execute did on the same samples (fixture rxoff/*, made by tests/golden/
    make_t2_golden.py from the compiled reference): a recording with a 77 Hz (-64 Hz) carrier offset behind an emulated tuner, a
    receiver clock that is 2 ppm (-1 ppm) off, 172 032-sample buffers, nothing configured. Held to the reference: the re-tune requests
    (same buffers, the estimate within 1 Hz), the acquisition state after every execute(), the loop trajectory symbol by symbol (chunk
    lengths, phase_est_filtered, frequency_est_filtered, sample_rate_est_filtered), the de-interleaved cells of every TI block, the
    BBFRAMEs and the transport stream packet for packet. The two forms of the product (loops on the device / on the host) must agree
    with each other bit for bit."""
    g = sub(grx, "rxoff", name)
    c, m, bi, bq, sent = rc.rx_offset_case(name)
    same_input = rc.sha(np.stack([bi, bq])) == str(g["base_sha"])
    moves = g["moves"]
    i16, q16 = t2_tx.rx_offset_tuned(bi, bq, c["buf"], c["cfo_hz"], [(int(k), float(t)) for k, _, t in moves])
    i16.tofile(tmp_path / "i.s16")
    q16.tofile(tmp_path / "q.s16")
    exe = _build_stage_driver(tmp_path)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    env.update(STAGE_SAMPLE_RATE_OFF=repr(float(c["rate_off_hz"])), STAGE_TUNER_MOVES=",".join("%d:%.17g" % (int(k), r) for k, r, _ in moves),
               T2GPU_DEMOD_STRICT_LOOPS="1", STAGE_THREADS="1")
    runs = {}
    for loop in ("1", "0"):
        tag = "dev" if loop == "1" else "host"
        e = dict(env, STAGE_DEVICE_LOOP=loop, STAGE_TRACE=str(tmp_path / ("trace_%s.f64" % tag)), STAGE_DUMP=str(tmp_path / ("dump_%s" % tag)))
        p = subprocess.run([exe, "rx", str(tmp_path / "i.s16"), str(tmp_path / "q.s16"), str(tmp_path / ("out_%s.ts" % tag)), str(c["buf"]), "0",
                            str(tmp_path / ("log_%s.txt" % tag))], env=e, stderr=subprocess.PIPE, text=True, timeout=900)
        assert p.returncode == 0, p.stderr
        runs[tag] = dict(trace=np.fromfile(tmp_path / ("trace_%s.f64" % tag), np.float64).reshape(-1, 11), ts=np.fromfile(tmp_path / ("out_%s.ts" % tag), np.uint8),
                         log=open(tmp_path / ("log_%s.txt" % tag)).read())
    if os.environ.get("RXOFF_SAVE"):                                            # (development: keep what the runs wrote for a look on another machine)
        import shutil
        os.makedirs(os.environ["RXOFF_SAVE"], exist_ok=True)
        for tag in runs:
            for fn in ("trace_%s.f64" % tag, "log_%s.txt" % tag, "out_%s.ts" % tag):
                shutil.copy(tmp_path / fn, os.path.join(os.environ["RXOFF_SAVE"], name + "_" + fn))
        ti_all = np.fromfile(str(tmp_path / "dump_dev") + ".ti.c64", np.complex64)
        np.save(os.path.join(os.environ["RXOFF_SAVE"], name + "_ti_sample.npy"), ti_all[::rc.RX_OFFSET_TI_STEP])
    # ---- the product's two forms: the same trajectory, the same bytes
    assert np.array_equal(runs["dev"]["trace"].view(np.uint64), runs["host"]["trace"].view(np.uint64)), "loops on the device / on the host: different trajectories"
    assert np.array_equal(runs["dev"]["ts"], runs["host"]["ts"])
    r = runs["dev"]
    # ---- the tuner: asked to move at the buffers the reference asked at, by what the reference asked for (its P1 estimate) within 1 Hz
    asks = [ln.split() for ln in r["log"].splitlines() if ln.startswith("set_rf_ext")]
    assert [int(a[1]) for a in asks] == [int(k) for k, _, _ in moves] and all(a[3] == "listed" for a in asks), (asks, moves)
    worst_ask = max(abs(float(a[2]) - req) for a, (_, req, _) in zip(asks, moves))
    assert worst_ask < 1.0, (asks, moves)
    # ---- acquisition, execute() by execute()
    lines = [dict(zip(ln.split()[2::2], ln.split()[3::2])) for ln in r["log"].splitlines() if ln.startswith("buf ")]
    ref_state = [dict(zip(ol.RX_STATE, row)) for row in g["state_log"]]
    assert len(lines) == len(ref_state)
    for k, (a, b) in enumerate(zip(lines, ref_state)):
        mine = (int(a["next"]), int(a["p2_init"]), int(a["init"]), int(a["deint"]), int(a["crc"]))
        want = (int(b["next_symbol_type"]), int(b["p2_init"]), int(b["demodulator_init"]), int(b["deint_start"]), int(b["crc32_l1_pre"]))
        assert mine == want, (k, mine, want)
        if want[1]:
            assert int(a["gi"]) == int(b["guard_interval_size"]), (k, a["gi"], b["guard_interval_size"])
    # ---- the trajectory, symbol by symbol
    t, w = r["trace"], g["traj"]
    col = {n: i for i, n in enumerate(ol.RefRx.TRAJ)}
    assert t.shape[0] == w.shape[0], (t.shape, w.shape)
    assert np.array_equal(t[:, 1], w[:, col["next_symbol_type"]]) and np.array_equal(t[:, 2], w[:, col["idx_symbol"]])
    # chunks: a chunk cut short by the end of an execute() buffer says where the BUFFER ended relative to the symbol, which moves with the
    # P1 position by a few samples from frame to frame; whole-symbol chunks are compared (all but the ~1 in 5 that straddle two buffers)
    nominal = np.median(t[:, 3])
    whole = (t[:, 3] > nominal - 64) & (w[:, col["chunk"]] > nominal - 64)
    is_p2 = t[:, 2] == 1
    d_chunk = np.where(whole & ~is_p2, np.abs(t[:, 3] - w[:, col["chunk"]]), 0.0)
    d_p1 = np.where(whole & is_p2, np.abs(t[:, 3] - w[:, col["chunk"]]), 0.0)
    assert whole.mean() > 0.7, whole.mean()
    # a symbol's raw phase estimate follows from the reference's filter state: 2 (phase_est_filtered - integral) / k_p (DSP/loop_filters.hh:36-44)
    ref_phase_est = 2.0 * (w[:, col["phase_est_filtered"]] - w[:, col["phase_integral"]]) / w[:, col["phase_k_p"]]
    d_phase = np.abs(t[:, 4] - w[:, col["phase_est_filtered"]])
    d_freq = np.abs(t[:, 5] - w[:, col["frequency_est_filtered"]])
    d_rate = np.abs(t[:, 6] - w[:, col["sample_rate_est_filtered"]]) / 8.0e-9
    f_ref = np.abs(w[:, col["frequency_est_filtered"]]).max()
    exact = dict(chunk=float((d_chunk == 0).mean()), phase=float((d_phase == 0).mean()), freq=float((d_freq == 0).mean()), rate=float((d_rate < 0.5).mean()))
    print("closed loop %s: input %s; %d tracked symbols (%d whole-symbol chunks); re-tune estimates within %.3f Hz; max |d chunk| %d (P2: %d), |d phase_est_filtered| %.3e rad, "
          "|d frequency_est_filtered| %.3e (%.3e of the tracked %.3e), |d sample_rate_est_filtered| %.1f steps; equal to the last bit: %s"
          % (name, "identical to the fixture's" if same_input else "NOT the fixture's bits (this host's libm / FFT round differently)", t.shape[0], int(whole.sum()),
             worst_ask, int(d_chunk.max()), int(d_p1.max()), d_phase.max(), d_freq.max(), d_freq.max() / f_ref, f_ref, d_rate.max(), exact))
    H, tol_h = RXOFF_HEAD, RXOFF_TOL["head"]
    head = dict(chunk=float(np.abs(t[:H, 3] - w[:H, col["chunk"]]).max()), rate_steps=float(d_rate[:H].max()), phase=float(d_phase[:H].max()),
                freq_rel=float(d_freq[:H].max() / f_ref), phase_est=float(np.abs(t[:H, 8] - ref_phase_est[:H]).max()))
    print("closed loop %s: the first %d tracked symbols against the reference: %s" % (name, H, head))
    for k, v in head.items():
        assert v <= tol_h[k], (k, v, tol_h[k])
    assert np.array_equal(t[:H, 9].astype(np.float32) != 0, w[:H, col["old_sample_rate_est"]] != 0)
    assert np.abs(t[:H, 9] - w[:H, col["old_sample_rate_est"]]).max() <= 2e-6 * np.abs(w[:H, col["old_sample_rate_est"]]).max()   # the symbols' sample-rate estimates
    assert d_chunk.max() <= RXOFF_TOL["chunk"], d_chunk.max()
    assert d_p1.max() <= RXOFF_TOL["p1_samples"], d_p1.max()
    assert d_phase.max() <= RXOFF_TOL["phase"], d_phase.max()
    assert d_freq.max() <= RXOFF_TOL["freq_rel"] * f_ref, (d_freq.max(), f_ref)
    assert d_rate.max() <= RXOFF_TOL["rate_steps"], d_rate.max()
    # the derived resampling value the reference holds (resample - sample_rate_est_filtered, :157): same nominal value, to a float's last bits
    assert abs((t[-1, 7] + t[-1, 6]) - w[-1, col["resample"]]) < 1e-12
    # ---- the de-interleaved cells of every TI block
    ti = np.fromfile(str(tmp_path / "dump_dev") + ".ti.c64", np.complex64)
    sizes = [int(s) for s, _ in g["ti_meta"]]
    assert ti.size == sum(sizes), (ti.size, sizes)
    pos, per_block = 0, []
    for b, n in enumerate(sizes):
        mine = ti[pos:pos + n][::rc.RX_OFFSET_TI_STEP]
        per_block.append(float(np.abs(mine - g["ti_sample"][b][:len(mine)]).max()))
        pos += n
    print("closed loop %s: %d TI blocks, sampled cells within %s of the reference's" % (name, len(sizes), " ".join("%.1e" % v for v in per_block)))
    assert per_block[0] < RXOFF_TOL["ti_cells_first"] and max(per_block) < RXOFF_TOL["ti_cells"], per_block
    # ---- behind the FEC: the reference's BBFRAMEs and its transport stream, packet for packet
    ts = r["ts"]
    assert ts.size == int(g["ts_len"]), (ts.size, int(g["ts_len"]))
    assert np.array_equal(rc.crc_rows(ts[:ts.size // 188 * 188].reshape(-1, 188)), g["ts_packet_crc"])
    last = [ln for ln in r["log"].splitlines() if ln.startswith("bbframes ")][-1].split()
    assert int(last[1]) == int(g["bbframes"]), (last, int(g["bbframes"]))
