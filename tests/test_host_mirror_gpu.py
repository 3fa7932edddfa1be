"""GPU tier: the C++ host side above the C ABI (include/t2gpu_stages.hpp -- the reference's object / slot / signal shapes in the
reference's own language). tests/cpp/stage_mirror_test.cpp wires the classes the way the reference wires its objects with
connect() and is fed vectors from here; what it emits is compared with the oracle. Also proves the library is usable from a
plain C++ program with no Python / torch in the process."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
import t2_tx

pytestmark = pytest.mark.gpu
ROOT = ol.ROOT


@pytest.fixture(scope="module")
def driver(built, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "stage_mirror_test")
    pkg = os.path.join(ROOT, "sdr_receiver_dvb_t2_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "stage_mirror_test.cpp"), "-L" + pkg, "-lt2gpu",
                           "-Wl,-rpath," + pkg, "-o", out])
    return out


def run(driver, *args, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([driver] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr
    return p.stderr


def sig(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.2).astype(np.complex64)


def test_decimator_and_farrow_classes(driver, tmp_path):
    x = sig(50001, 1)
    x.tofile(tmp_path / "in.c64")
    run(driver, "decim", tmp_path / "in.c64", tmp_path / "dec.c64", 4097)
    got = np.fromfile(tmp_path / "dec.c64", np.complex64)
    o = ol.OraDecim()
    want = np.concatenate([o(x[p:p + 4097]) for p in range(0, len(x), 4097)])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for resample in (0.5, 0.5 - 24e-9):
        run(driver, "farrow", tmp_path / "in.c64", tmp_path / "far.c64", 10007, repr(resample))
        got = np.fromfile(tmp_path / "far.c64", np.complex64)
        f = ol.OraFarrow()
        want = np.concatenate([f(x[p:p + 10007], resample) for p in range(0, len(x), 10007)])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("fec_type,cod,threads", [(0, 0, 0), (1, 3, 0), (1, 3, 1), (1, 3, 2), (0, 0, 1)])
def test_ldpc_bch_chain_with_drop(driver, tmp_path, fec_type, cod, threads):
    """ldpc_decoder.bit_bch -> bch_decoder.execute -> bit_descramble, three SIMD batches, the middle one undecodable: the
    reference prints its message and drops that batch; the other two come out descrambled, frame by frame. threads 1: the stage on a
    thread of its own, the three batches decoded by ONE launch (t2gpu_ldpc_submit_add / _go: one verdict per batch); 2: that thread with
    one launch per batch (STAGE_MERGE=0)."""
    cid = ol.code_id(fec_type, cod)
    n, k, _, _ = ol.ldpc_params(cid)
    info, llr = ol.make_llr(cid, 96, 0.55 if fec_type else 0.7, 5)
    rng = np.random.Generator(np.random.PCG64(6))
    llr[32:64] = rng.integers(-20, 21, size=(32, n), dtype=np.int8)              # noise only: never converges
    llr.tofile(tmp_path / "llr.i8")
    err = run(driver, "fec", tmp_path / "llr.i8", tmp_path / "out.u8", fec_type, cod,
              env_extra={"STAGE_THREADS": str(min(threads, 1)), "STAGE_MERGE": "0" if threads == 2 else "1"})
    assert err.count("LDPC decoder could not recover the codeword!") == 1
    k_bch = t2_tx.K_BCH[cid]
    got = np.fromfile(tmp_path / "out.u8", np.uint8).reshape(-1, 1 + k_bch)
    assert got.shape[0] == 64 and (got[:, 0] == 0).all()                          # plp id of every emitted frame
    want = []
    for b in (0, 64):
        t, bits, _ = ol.ora_decode(cid, llr[b:b + 32])
        assert t >= 0
        want.append(ol.ora_bch_descramble(cid, bits))
    assert np.array_equal(got[:, 1:], np.concatenate(want))


def test_bch_class_with_outer_code(driver, tmp_path):
    """bch_decoder.outer_code (not in the reference, whose decoder is a TODO, bch_decoder.cpp:136). The LDPC stage hands over
    valid LDPC words only, so the damage is put where just the outer code sees it: the words that get LDPC-encoded differ from
    BCH codewords in a few bits. Frames with up to t such bits come out repaired, the others flagged and passed as they are."""
    fec_type, cod = 0, 2
    cid = ol.code_id(fec_type, cod)
    m, t, kb, nb = ol.bch_params(cid)
    rng = np.random.Generator(np.random.PCG64(16))
    msg = rng.integers(0, 2, (32, kb), dtype=np.uint8)
    words = np.concatenate([msg, t2_tx.bch_parity(cid, msg)], axis=1)
    errs = rng.integers(0, t + 3, 32)
    errs[:3] = (0, t, t + 1)
    sent = words.copy()
    for f in range(32):
        sent[f, rng.choice(nb, errs[f], replace=False)] ^= 1
    _, llr = ol.make_llr(cid, 32, 0.55, 17, info=sent)                            # LDPC-encodes the damaged words
    llr.tofile(tmp_path / "llr.i8")
    err = run(driver, "fec", tmp_path / "llr.i8", tmp_path / "out.u8", fec_type, cod, "outer")
    status = [int(l.split()[1]) for l in err.splitlines() if l.startswith("outer ")]
    assert status == [int(e) if e <= t else -1 for e in errs]
    got = np.fromfile(tmp_path / "out.u8", np.uint8).reshape(-1, 1 + kb)[:, 1:]
    want = np.where((errs <= t)[:, None], words[:, :kb], sent[:, :kb])
    assert np.array_equal(got, ol.ora_bch_descramble(cid, np.concatenate([want, np.zeros((32, nb - kb), np.uint8)], axis=1)))


def test_p1_class(driver, tmp_path):
    rng = np.random.Generator(np.random.PCG64(8))
    noise = lambda n, s: ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * s).astype(np.complex64)
    x = np.concatenate([noise(2700, 0.05), (t2_tx.p1_symbol(0, 8) * 0.3).astype(np.complex64) + noise(2048, 0.02), noise(3000, 0.05)])
    level = float(np.mean(np.abs(x.real)) * np.mean(np.abs(x.imag)))
    x.tofile(tmp_path / "p1.c64")
    run(driver, "p1", tmp_path / "p1.c64", tmp_path / "p1.txt", repr(level))
    hit, consume, idx_sym, preamble, fft_mode, decoded, cfo = open(tmp_path / "p1.txt").read().split()
    r = ol.OraP1().execute(x, 0, True, level)
    assert int(hit) == 1 and r["detected"]
    assert (int(consume), int(idx_sym), int(preamble), int(fft_mode), int(decoded)) == \
        (r["consume"], r["idx_buffer_sym"], r["preamble"], r["fft_mode"], r["p1_decoded"]) and int(fft_mode) == 4
    assert abs(float(cfo) - r["coarse_freq_offset"]) < 0.5


def test_fec_side_from_cells_two_plps(driver, tmp_path):
    """The whole FEC side in C++ as the reference wires it -- time_deinterleaver (two PLPs, the second with two TI blocks) ->
    llr_demapper -> ldpc_decoder -> bch_decoder -> bb_de_header(need_plp) -- fed symbol by symbol with the equalised cells of one
    T2 frame. Whole SIMD batches come out (the reference never flushes the tail); BBFRAMEs are bit-exact and tagged with their
    PLP, the transport stream holds only need_plp's packets."""
    mode, lps, mod, fec_type, code_rate = (4, 1, 6, 4, 0, 40), 200, 2, 0, 0
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    cpf = 16200 // (2 * (mod + 1))
    nb = t2_tx.plp_blocks_per_frame(m, lps, cpf)
    n0 = 40
    n1 = 63                                                                  # 103 frames: three SIMD batches and a tail of 7
    k_bch = t2_tx.K_BCH[cid]
    ts0 = t2_tx.ts_packets(n0 * (k_bch // 1496 + 1) + 8, 41)
    ts1 = t2_tx.ts_packets(n1 * (k_bch // 1496 + 1) + 8, 42)
    fr0, _ = t2_tx.bbframes_hem(ts0, k_bch, n0)
    fr1, _ = t2_tx.bbframes_hem(ts1, k_bch, n1)
    c0 = t2_tx.cells_from_codewords(t2_tx.fec_encode(cid, t2_tx.scramble(fr0)), mod, fec_type, code_rate, True)
    c1 = t2_tx.cells_from_codewords(t2_tx.fec_encode(cid, t2_tx.scramble(fr1)), mod, fec_type, code_rate, True)
    a = n1 // 2
    stream = np.concatenate([t2_tx.interleave_ti_block(c0), t2_tx.interleave_ti_block(c1[:a]), t2_tx.interleave_ti_block(c1[a:])])
    rng = np.random.Generator(np.random.PCG64(43))
    cap = (m.c_p2 - 1840 - lps) + m.n_data * m.c_data
    cells = np.zeros(1840 + lps + cap, np.complex64)                         # L1 cells (skipped by the stage), PLP cells, dummy cells
    cells[1840 + lps:1840 + lps + stream.size] = stream
    cells += (0.05 * (rng.standard_normal(cells.size) + 1j * rng.standard_normal(cells.size))).astype(np.complex64)
    cells.tofile(tmp_path / "cells.c64")
    sizes = [m.c_p2] + [m.c_data] * m.n_data
    assert sum(sizes) == cells.size
    # The two PLPs leave more than a TI block of dummy cells in the frame. No PLP starts behind PLP 1, so the reference stays in
    # PLP 1 and de-interleaves the dummy cells with its geometry (time_deinterleaver.cpp:357-371): the frames that come out of
    # them are noise, their SIMD batches are dropped by the LDPC stage -- together with the 7 good frames of the tail.
    assert cap - (n0 + n1) * cpf > (n1 - a) * cpf
    plps = [mod, fec_type, 1, n0, 1, 0, n0, mod, fec_type, 1, n1, 2, n0 * cpf, n1]
    run(driver, "cells", tmp_path / "cells.c64", tmp_path / "out.u8", 1, tmp_path / "ts.u8", lps, code_rate, len(sizes), *sizes, 2, *plps)
    got = np.fromfile(tmp_path / "out.u8", np.uint8).reshape(-1, 1 + k_bch)
    assert got.shape[0] == 96
    want = np.concatenate([fr0, fr1])[:96]
    assert np.array_equal(got[:, 0], np.array([0] * n0 + [1] * (96 - n0), np.uint8))
    assert np.array_equal(got[:, 1:], want)
    ts = np.fromfile(tmp_path / "ts.u8", np.uint8)
    n_pkts = ((96 - n0) * ((k_bch - 80) // 8)) // 187 - 1
    assert np.array_equal(ts[:n_pkts * 188], ts1.reshape(-1)[:n_pkts * 188])
    # the same chain with the hand-over of the stage outputs by address switched off again (t2::handoff(false): every stage copies its
    # input in, which is what the plain C ABI does by default): it changes where the bytes come from, not one of them
    run(driver, "cells", tmp_path / "cells.c64", tmp_path / "out0.u8", 1, tmp_path / "ts0.u8", lps, code_rate, len(sizes), *sizes, 2, *plps,
        env_extra={"STAGE_HANDOFF": "0"})
    assert np.array_equal(np.fromfile(tmp_path / "out0.u8", np.uint8), got.reshape(-1))
    assert np.array_equal(np.fromfile(tmp_path / "ts0.u8", np.uint8), ts)
    # ... and with the de-interleaver and the LDPC stage on threads of their own (t2::time_deinterleaver / t2::ldpc_decoder own_thread:
    # TI blocks alternate between two buffers, batches are awaited and emitted beside the caller): same frames in the same order
    run(driver, "cells", tmp_path / "cells.c64", tmp_path / "out1.u8", 1, tmp_path / "ts1.u8", lps, code_rate, len(sizes), *sizes, 2, *plps,
        env_extra={"STAGE_THREADS": "1"})
    assert np.array_equal(np.fromfile(tmp_path / "out1.u8", np.uint8), got.reshape(-1))
    assert np.array_equal(np.fromfile(tmp_path / "ts1.u8", np.uint8), ts)


def _unconfigured_stream(tmp_path, n_frames, seed, cfo_hz, spoil_frame=None):
    """t2_tx.rx_test_stream written to tmp_path as i.s16 / q.s16 in 2^18-sample buffers."""
    m, i16, q16, buf, marks = t2_tx.rx_test_stream(n_frames, seed, cfo_hz, spoil_frame)
    i16.tofile(tmp_path / "i.s16")
    q16.tofile(tmp_path / "q.s16")
    return m, buf, marks


def test_demodulator_class_acquires_and_decodes(driver, tmp_path):
    """t2::dvbt2_demodulator::execute(len, i, q, signal) -- the boundary slot -- fed buffer by buffer with int16 I/Q by a loop that is
    rx_sdrplay::start over a recording. Nothing about the signal is configured: FFT size and SISO come from P1, the guard
    interval from the reference's own search (first guess 1/4 fails, the brute-force list starts at 1/32 -- the signal's value; any
    other value costs seven frames per list entry there as here, because the equaliser's phase unwrapping, data_symbol.cpp:189-191,
    only follows one sign of slope and so needs the exact guard length), carrier mode, pilot pattern, frame length from L1-pre,
    the PLP from L1-post; a 60 Hz carrier offset makes P1 ask for re-tunes until it reports less than 10 Hz. Everything from
    the time de-interleaver down is the reference's connect() chain in C++. The transport stream of every frame after
    acquisition comes back byte for byte."""
    n_frames = 12
    m, buf, marks = _unconfigured_stream(tmp_path, n_frames, 191, 60.0)
    run(driver, "rx", tmp_path / "i.s16", tmp_path / "q.s16", tmp_path / "out.ts", buf, 0, tmp_path / "log.txt",
        env_extra={"STAGE_DEVICE_LOOP": "0"})                                  # the loops on the host for every symbol; the library's default follows below
    log = open(tmp_path / "log.txt").read()
    print(log.splitlines()[-1])                                                # throughput of the symbol-by-symbol form (pytest -s)
    lines = [ln for ln in log.splitlines() if ln.startswith("buf ")]
    last = dict(zip(lines[-1].split()[2::2], lines[-1].split()[3::2]))
    assert "amount_plp 1" in log, log
    assert log.count("set_rf") >= 2, log                                    # the initial tune of reset() and at least one re-tune
    assert last["init"] == "1" and last["deint"] == "1" and last["crc"] == "1" and last["resets"] == "0", log
    assert int(last["gi"]) == m.fft_size // 32, log                         # GUARD_INTERVAL of L1-pre confirmed the search
    got = np.fromfile(tmp_path / "out.ts", np.uint8).tobytes()
    found = [f for f in range(n_frames) if got.find(marks[f]) >= 0]
    # acquisition takes P1 (tune) .. P1 (init) .. P2 (guard search) .. P2 (L1-pre) .. P2 (L1-post): the frames after that are all
    # there; the last frames' FEC blocks wait in an unfinished SIMD batch
    assert len(found) >= 3 and found == list(range(found[0], found[0] + len(found))) and found[0] <= 8, (found, log)
    assert found[-1] >= n_frames - 2, (found, log)
    # the same stream with the tracking loops of every frame's data symbols ON THE DEVICE (t2gpu_demod_set_device_loop: sym_sync_kernel's
    # last lane runs the two loop filters, front_one_kernel's workgroup 0 plans the chunk's NCO, the next chunk is launched ahead of the
    # symbol's results and the host follows one symbol behind, comparing as it goes): the same transport stream, byte for byte, and the
    # same loop values at the end
    run(driver, "rx", tmp_path / "i.s16", tmp_path / "q.s16", tmp_path / "out_dev.ts", buf, 0, tmp_path / "log_dev.txt",
        env_extra={"STAGE_DEVICE_LOOP": "1"})
    assert np.fromfile(tmp_path / "out_dev.ts", np.uint8).tobytes() == got
    log_dev = open(tmp_path / "log_dev.txt").read()
    assert [ln for ln in log_dev.splitlines() if ln.startswith("buf ")] == lines
    # ... (that run had the chunk that completes a symbol and the symbol's transform in ONE launch, the default; here as two)
    run(driver, "rx", tmp_path / "i.s16", tmp_path / "q.s16", tmp_path / "out_two.ts", buf, 0, tmp_path / "log_two.txt",
        env_extra={"STAGE_DEVICE_LOOP": "1", "STAGE_CHAIN_ONE": "0"})
    assert np.fromfile(tmp_path / "out_two.ts", np.uint8).tobytes() == got
    assert [ln for ln in open(tmp_path / "log_two.txt").read().splitlines() if ln.startswith("buf ")] == lines
    # ... with page-locked I/Q buffers, which come over chunk by chunk inside the chunks' launches (t2gpu_demod_set_copy_ahead), and as one
    # copy per call
    for ahead in ("1", "0"):
        run(driver, "rx", tmp_path / "i.s16", tmp_path / "q.s16", tmp_path / "out_pin.ts", buf, 0, tmp_path / "log_pin.txt",
            env_extra={"STAGE_DEVICE_LOOP": "1", "STAGE_PIN": "1", "STAGE_COPY_AHEAD": ahead})
        assert np.fromfile(tmp_path / "out_pin.ts", np.uint8).tobytes() == got, ahead
        assert [ln for ln in open(tmp_path / "log_pin.txt").read().splitlines() if ln.startswith("buf ")] == lines, ahead


def test_demodulator_class_resets_and_recovers(driver, tmp_path):
    """A P2 symbol that cannot pass the L1-pre CRC after the demodulator has initialised: the reference calls reset(), raises
    signal->reset and signal->p1_reset (dvbt2_demodulator.cpp:418-424), the SDR thread re-tunes from scratch (rx_sdrplay.cpp:135-156,
    229-235) and the next decoded P1 restores p2_init / demodulator_init without a new guard search (:293-297). The stream keeps
    coming out afterwards."""
    n_frames, spoil = 16, 9
    m, buf, marks = _unconfigured_stream(tmp_path, n_frames, 291, 0.0, spoil_frame=spoil)
    run(driver, "rx", tmp_path / "i.s16", tmp_path / "q.s16", tmp_path / "out.ts", buf, 0, tmp_path / "log.txt",
        env_extra={"STAGE_DEVICE_LOOP": "0"})
    log = open(tmp_path / "log.txt").read()
    lines = [ln for ln in log.splitlines() if ln.startswith("buf ")]
    last = dict(zip(lines[-1].split()[2::2], lines[-1].split()[3::2]))
    assert last["resets"] == "1" and log.count("\nreset") >= 1, log            # reset() in the demodulator, reset() in the SDR loop
    assert last["init"] == "1" and last["crc"] == "1" and int(last["gi"]) == m.fft_size // 32, log
    got = np.fromfile(tmp_path / "out.ts", np.uint8).tobytes()
    found = [f for f in range(n_frames) if got.find(marks[f]) >= 0]
    assert any(f < spoil for f in found) and any(f > spoil + 1 for f in found), (found, log)
    assert spoil not in found, found
    # ... and with the loops on the device: a reset in the middle of it, the same bytes
    run(driver, "rx", tmp_path / "i.s16", tmp_path / "q.s16", tmp_path / "out_dev.ts", buf, 0, tmp_path / "log_dev.txt",
        env_extra={"STAGE_DEVICE_LOOP": "1"})
    assert np.fromfile(tmp_path / "out_dev.ts", np.uint8).tobytes() == got
    # ... page-locked I/Q coming over chunk by chunk, a reset in the middle of it
    run(driver, "rx", tmp_path / "i.s16", tmp_path / "q.s16", tmp_path / "out_pin.ts", buf, 0, tmp_path / "log_pin.txt",
        env_extra={"STAGE_DEVICE_LOOP": "1", "STAGE_PIN": "1"})
    assert np.fromfile(tmp_path / "out_pin.ts", np.uint8).tobytes() == got


def test_example_rx_file_program(driver, tmp_path):
    """examples/t2gpu_rx_file.cpp -- the Qt-free source + sink around the accelerated path -- writes the same
    transport stream to a file as the test driver collects, and the same bytes as UDP datagrams (one per BBFRAME, as the
    reference's bb_de_header sends them) to a local port."""
    import socket
    import threading
    exe = str(tmp_path / "t2gpu_rx_file")
    pkg = os.path.join(ROOT, "sdr_receiver_dvb_t2_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "t2gpu_rx_file.cpp"), "-L" + pkg, "-lt2gpu", "-Wl,-rpath," + pkg, "-o", exe])
    m, buf, marks = _unconfigured_stream(tmp_path, 10, 391, 0.0)
    run(driver, "rx", tmp_path / "i.s16", tmp_path / "q.s16", tmp_path / "ref.ts", buf, 0, tmp_path / "log.txt")
    want = np.fromfile(tmp_path / "ref.ts", np.uint8).tobytes()
    assert len(want) > 100000
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    subprocess.run([exe, str(tmp_path / "i.s16"), str(tmp_path / "q.s16"), "--out", str(tmp_path / "out.ts"), "--buf", str(buf)],
                   check=True, env=env, timeout=300, stderr=subprocess.PIPE)
    assert np.fromfile(tmp_path / "out.ts", np.uint8).tobytes() == want
    # UDP sink: collect the datagrams on a local port
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 1 << 24)
    rx.bind(("127.0.0.1", 0))
    rx.settimeout(2.0)
    port = rx.getsockname()[1]
    got = []

    def collect():
        try:
            while True:
                got.append(rx.recv(65536))
        except socket.timeout:
            pass

    t = threading.Thread(target=collect)
    t.start()
    subprocess.run([exe, str(tmp_path / "i.s16"), str(tmp_path / "q.s16"), "--udp", str(port), "--buf", str(buf)],
                   check=True, env=env, timeout=300, stderr=subprocess.PIPE)
    t.join()
    rx.close()
    joined = b"".join(got)
    # localhost UDP can drop under load: what arrived must be whole BBFRAME payloads of the stream, in order
    assert len(got) > 10 and len(joined) > len(want) // 2
    pos = 0
    for d in got:
        at = want.find(d, pos)
        assert at >= 0
        pos = at + len(d)


def test_example_program_on_the_benchmark_mode(tmp_path):
    """The slot-shaped path on BASELINE config 3's mode (32K extended PP7 GI 1/128, 59 data symbols, one 256-QAM 64800 r=3/4 PLP of 202 FEC
    blocks per frame), nothing configured. With the reference's int8 cast every SIMD batch of a 256-QAM PLP is lost in the LDPC stage
    (the outer constellation points wrap, llr_demapper.cpp:722-737) -- the program reports exactly that; with
    t2::llr_demapper::saturate_llr (the clamping extension) the transport stream of the frames behind the acquisition comes out,
    every packet one of those sent."""
    import json
    exe = str(tmp_path / "t2gpu_rx_file")
    pkg = os.path.join(ROOT, "sdr_receiver_dvb_t2_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "t2gpu_rx_file.cpp"), "-L" + pkg, "-lt2gpu", "-Wl,-rpath," + pkg, "-o", exe])
    mode, lps, mod, fec_type, code_rate = (5, 1, 6, 4, 0, 59), 350, 3, 1, 3
    m = ol.ora_mode(*mode)
    cid = ol.code_id(fec_type, code_rate)
    cpf = 64800 // (2 * (mod + 1))
    nb = t2_tx.plp_blocks_per_frame(m, lps, cpf)
    k_bch = t2_tx.K_BCH[cid]
    per = nb * (k_bch // 1496 + 1)
    frames, sent = [], []
    for f in range(2):
        ts = t2_tx.ts_packets(per, 700 + f)
        stream, _, _ = t2_tx.build_plp_frame_cells(cid, mod, fec_type, code_rate, ts, nb)
        l1 = t2_tx.l1_cells(mode, lps, mod, fec_type, code_rate, nb, frame_idx=f)
        frames.append(t2_tx.build_frame(m, stream, lps, 800 + f, snr_db=None, phase=0.0, l1_cells=l1))
        sent.append(ts)
    i16, q16, flen = t2_tx.iq_stream(frames, m.fft_size // 128, 10, 21.0, 9)
    n_frames = 10
    np.concatenate([i16] * (n_frames // 2)).tofile(tmp_path / "i.s16")
    np.concatenate([q16] * (n_frames // 2)).tofile(tmp_path / "q.s16")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    want = {bytes(p) for p in np.concatenate(sent)}
    for saturate in (0, 1):
        p = subprocess.run([exe, str(tmp_path / "i.s16"), str(tmp_path / "q.s16"), "--out", str(tmp_path / "out.ts"), "--json", "1", "--saturate", str(saturate)],
                           check=True, env=env, timeout=300, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        r = json.loads(p.stdout.strip().splitlines()[-1])
        assert r["deint_start"] == 1 and r["resets"] == 0, r
        dropped = p.stderr.count("LDPC decoder could not recover the codeword!")
        got = np.fromfile(tmp_path / "out.ts", np.uint8)
        if not saturate:
            assert r["bbframes"] == 0 and got.size == 0 and dropped >= 3 * nb // 32, (r, dropped)
            continue
        assert dropped == 0 and r["bbframes"] >= 3 * nb // 32 * 32, (r, dropped)
        pk = got[:got.size // 188 * 188].reshape(-1, 188)
        assert pk.shape[0] > 3 * nb * ((k_bch - 80) // 8 // 188) and (pk[:, 0] == 0x47).all()
        bad = sum(bytes(row) not in want for row in pk)
        # each T2 frame carries its own packet set: the packet that straddles two frames is half of one and half of the other
        assert bad <= n_frames, (bad, pk.shape[0])
        # that run took the round-5 forms (defaults): the chunk that completes a 32K symbol and the symbol's transform + floats as ONE launch
        # (front_fft_one_kernel), the SIMD batches of a TI block decoded by ONE launch (t2gpu_ldpc_submit_add / _go). The forms of before --
        # two launches per symbol's transform behind the chunk's, one decode per batch on streams of their own -- give the same file, byte
        # for byte
        for extra in (["--chain-one", "0"], ["--chain-one", "0", "--ldpc-merge", "0"], ["--device-loop", "0"]):
            subprocess.run([exe, str(tmp_path / "i.s16"), str(tmp_path / "q.s16"), "--out", str(tmp_path / "out2.ts"), "--json", "1", "--saturate", "1"] + extra,
                           check=True, env=env, timeout=300, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            assert np.array_equal(np.fromfile(tmp_path / "out2.ts", np.uint8), got), extra
