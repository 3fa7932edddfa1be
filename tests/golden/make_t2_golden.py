#!/usr/bin/env python3
"""Generate tests/golden/t2sym_golden.npz, t2fec_golden.npz and t2rx_golden.npz by RUNNING THE REFERENCE ITSELF: its src/DVB_T2
classes compiled where they lie under /root/reference by oracle/Makefile (targets _ref/libref_t2sym.so, _ref/libref_t2rx.so: g++
with the reference's flags, Qt 5.9.7 headers / moc / libraries of the base image under /opt/conda, the FFTW binary the reference
ships) and driven by oracle/ref_t2sym.cpp / ref_t2rx.cpp. Run in the build container:

    make -C oracle && python tests/golden/make_t2_golden.py

Every case runs in a process of its own (`--case kind:name`): the reference's demapper keeps function-local statics
(tests/ref_cases.py). Inputs are the deterministic ones of tests/ref_cases.py; the fixtures hold their SHA-256 (and the inputs
themselves where floating-point synthesis is involved), the reference's outputs whole where they are small and as per-row CRC-32
plus sample rows where they are large."""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402
import ref_cases as rc  # noqa: E402


def case_sym(name):
    m, spec_q, pre, plp = rc.sym_frame(name)
    mode = rc.SYM_MODES[name]
    out = {"mode": np.array(mode, np.int32)}
    r = ol.RefSym(0, mode[0])
    p0 = r.params()
    # tables the reference holds after init_dvbt2 (P2, always extended)
    mp, rf = r.carriers(0, 0)
    he, ho = r.freq_deint(0)
    out.update(p2_map=mp.astype(np.int8), p2_refer=rf, p2_h_even=he.astype(np.uint16), p2_h_odd=ho.astype(np.uint16),
               params_p1=np.array([p0[k] for k in ol.DVBT2_FIELDS], np.int32))
    keep = rc.sym_symbols(m)
    lo = {l: (p0["l_nulls"] if l == 0 else m.l_nulls) for l in keep}
    width = {l: (p0["k_total"] if l == 0 else m.k_total) for l in keep}
    spec = rc.dequantise(spec_q, rc.GRID_SPEC)
    # first P2 symbol: demodulator not initialised yet (dvbt2_demodulator.cpp:373, demodulator_init false)
    res = r.p2(spec[0], False)
    assert res["crc_pre"] and res["crc_post"], "reference failed the L1 CRCs"
    out.update(p2_cells=res["cells"], p2_sync=np.array([res["sample_rate_offset"], res["phase_offset"]], np.float32),
               l1_pre=np.array([res["l1_pre"][k] for k in ol.L1_PRE_NAMES], np.int64), l1_post=res["l1_post"].astype(np.int64))
    p1 = r.params()
    out["params_l1"] = np.array([p1[k] for k in ol.DVBT2_FIELDS], np.int32)
    r.data_init()                                                       # dvbt2_demodulator.cpp:386-395
    p2 = r.params()
    out["params_data"] = np.array([p2[k] for k in ol.DVBT2_FIELDS], np.int32)
    he, ho = r.freq_deint(1)
    out.update(data_h_even=he.astype(np.uint16), data_h_odd=ho.astype(np.uint16))
    maps, refs = [], []
    for l in range(m.n_p2, m.len_frame - m.l_fc):
        mp, rf = r.carriers(1, l)
        maps.append(mp.astype(np.int8))
        refs.append(rf)
    out["data_map_crc"] = rc.crc_rows(np.stack(maps))
    out["data_refer_crc"] = rc.crc_rows(np.stack(refs))
    for l in keep[1:1 + 3]:
        out["data_map_%d" % l] = maps[l - m.n_p2]
        out["data_refer_%d" % l] = refs[l - m.n_p2]
        cells, sro, ph = r.data(l, spec[l])
        out["data_cells_%d" % l] = cells
        out["data_sync_%d" % l] = np.array([sro, ph], np.float32)
    if m.l_fc:
        mp, rf = r.carriers(2, 0)
        he, ho = r.freq_deint(2)
        cells, sro, ph = r.fc(spec[m.len_frame - 1])
        out.update(fc_map=mp.astype(np.int8), fc_refer=rf, fc_h_even=he.astype(np.uint16), fc_h_odd=ho.astype(np.uint16), fc_cells=cells,
                   fc_sync=np.array([sro, ph], np.float32))
    out["symbols"] = np.array(keep, np.int32)
    for l in keep:
        out["spec_%d" % l] = spec_q[l, lo[l]:lo[l] + width[l]]          # active carriers only; the rest never reaches an equaliser
    return out


def case_p1(_):
    q, level = rc.p1_stream()
    x = rc.dequantise(q, rc.GRID_CELL)
    L = ol._ref_lib("ref_t2sym")
    import ctypes
    L.ref_p1_new.restype = ctypes.c_void_p
    L.ref_p1_execute.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    out = {"x": q, "level": np.float32(level)}
    for tag, splits in (("whole", [len(x)]), ("split", [1000, 1777, 2500, len(x) - 5277])):
        h = L.ref_p1_new()
        pos, rows, bufs = 0, [], []
        for n in splits:
            piece = np.ascontiguousarray(x[pos:pos + n])
            consume = 0
            while consume < n:                                           # symbol_acquisition's loop around p1 (:281-312)
                res = (ctypes.c_int * 7)()
                cfo = ctypes.c_double(0.0)
                bs = np.zeros(2048 + 8, np.complex64)
                L.ref_p1_execute(h, 1, level, n, piece.ctypes.data, consume, 0, res, ctypes.byref(cfo), bs.ctypes.data, bs.size)
                rows.append([pos] + list(res) + [int(round(cfo.value * 1000))])
                consume = res[1]
                if res[0]:
                    bufs.append(bs[:res[2]].copy())
            pos += n
        out["p1_%s" % tag] = np.array(rows, np.int64)
        out["p1_%s_buffer_sym" % tag] = bufs[0]
    return out


def case_fec(name):
    q, frames, ts, l1 = rc.fec_case(name)
    cells = rc.dequantise(q, rc.GRID_CELL)
    mod, fec_type, code_rate, nb, n, cpf, cid = rc.fec_geometry(name)
    tmp = tempfile.mkdtemp()
    r = ol.RefFec(os.path.join(tmp, "ref.ts"), 0)
    r.start(rc.FEC_L1_POST_SIZE, l1)
    r.frame(l1, np.concatenate([np.zeros(1840 + rc.FEC_L1_POST_SIZE, np.complex64), cells]))
    ti, llr, ld, bb, msg = r.taps(0), r.taps(1), r.taps(2), r.taps(3), r.taps(5)
    ts_out = r.ts()
    assert len(ti) == 1 and ti[0][0][0] == nb * cpf
    out = {"in_sha": np.array(rc.sha(q)), "ti_crc": rc.crc_rows(ti[0][1].reshape(nb, cpf)), "ti_first": ti[0][1][:cpf].copy(),
           "ti_last": ti[0][1][-cpf:].copy(), "sent_bbframes_crc": rc.crc_rows(frames), "sent_ts_crc": rc.crc_rows(ts)}
    L = np.concatenate([x[1] for x in llr]).reshape(-1, n)
    out.update(llr_crc=rc.crc_rows(L), llr_first=L[0].copy(), llr_last=L[-1].copy(), llr_batches=np.int32(len(llr)))
    k = ol.ldpc_params(cid)[1]
    if ld:
        B = np.concatenate([x[1] for x in ld]).reshape(-1, k)
        out.update(ldpc_crc=rc.crc_rows(B), ldpc_first=np.packbits(B[0]))
    out["ldpc_batches"] = np.int32(len(ld))
    if bb:
        D = np.stack([x[1] for x in bb])
        out.update(bb_crc=rc.crc_rows(D), bb_first=np.packbits(D[0]), bb_plp=np.array([x[0][1] for x in bb], np.int32))
    out["bb_count"] = np.int32(len(bb))
    out["ts"] = ts_out
    out["messages"] = np.array([bytes(b).decode() for _, b in msg])
    # the same cells through the STRICT build of the reference (oracle/Makefile: -O2 -ffp-contract=off, every float operation in the
    # order the reference writes it): the LLR rows the GPU tier holds the device to with no tolerance at all
    if ol.RefFec.available(strict=True):
        rs = ol.RefFec(os.path.join(tmp, "ref_strict.ts"), 0, strict=True)
        rs.start(rc.FEC_L1_POST_SIZE, l1)
        rs.frame(l1, np.concatenate([np.zeros(1840 + rc.FEC_L1_POST_SIZE, np.complex64), cells]))
        Ls = np.concatenate([x[1] for x in rs.taps(1)]).reshape(-1, n)
        rs.ts()
        out.update(llr_crc_strict=rc.crc_rows(Ls), llr_first_strict=Ls[0].copy(), llr_last_strict=Ls[-1].copy())
    return out


def case_ldpc_in(name):
    """The reference's LDPC stage driven through its public slot ldpc_decoder::execute (ldpc_decoder.h:90) with clamped 256-QAM
    LLRs, SIMD batch after SIMD batch; bch_decoder and bb_de_header follow through the reference's own signal connections."""
    llr, frames, ts, l1 = rc.ldpc_in_case(name)
    mod, fec_type, code_rate, nb, snr, seed = rc.LDPC_IN_CASES[name]
    cid = ol.code_id(fec_type, code_rate)
    tmp = tempfile.mkdtemp()
    r = ol.RefFec(os.path.join(tmp, "ref.ts"), 0)
    for b in range(nb // 32):
        r.ldpc_execute(l1, llr[32 * b:32 * b + 32])
    ld, bb, msg = r.taps(2), r.taps(3), r.taps(5)
    ts_out = r.ts()
    k = ol.ldpc_params(cid)[1]
    B = np.concatenate([x[1] for x in ld]).reshape(-1, k)
    D = np.stack([x[1] for x in bb])
    pk = ts_out[:ts_out.size // 188 * 188].reshape(-1, 188)
    return {"in_sha": np.array(rc.sha(llr)), "llr_first": llr[0].copy(), "saturated_frac": np.float32((np.abs(llr.astype(np.int32)) >= 127).mean()),
            "ldpc_batches": np.int32(len(ld)), "ldpc_crc": rc.crc_rows(B), "ldpc_first": np.packbits(B[0]),
            "bb_count": np.int32(len(bb)), "bb_crc": rc.crc_rows(D), "bb_first": np.packbits(D[0]),
            "sent_bbframes_crc": rc.crc_rows(frames), "ts_len": np.int64(ts_out.size), "ts_packet_crc": rc.crc_rows(pk),
            "ts_head": ts_out[:188 * 4].copy(), "ts_tail": ts_out[-188 * 2:].copy(),
            "messages": np.array([bytes(b).decode() for _, b in msg])}


def case_carry(name):
    """The reference's FEC chain (time_deinterleaver -> ... -> bb_de_header) over several T2 frames whose FEC block count is not a
    multiple of 32: what it has emitted after every frame shows the SIMD batches straddling the frames (llr_demapper.cpp:742-764)."""
    cells, sent, l1 = rc.carry_case(name)
    c, mod, fec_type, code_rate, n, cpf, cid = rc.carry_geometry(name)
    tmp = tempfile.mkdtemp()
    r = ol.RefFec(os.path.join(tmp, "ref.ts"), 0)
    r.keep(0, False)
    r.start(c["lps"], l1)
    after, llr, ld, bb = [], [], [], []
    for q in cells:
        r.frame(l1, np.concatenate([np.zeros(1840 + c["lps"], np.complex64), rc.dequantise(q, rc.GRID_CELL)]))
        llr += r.taps(1)
        ld += r.taps(2)
        bb += r.taps(3)
        after.append([len(llr), len(ld), len(bb)])
    ts_out = r.ts()
    k = ol.ldpc_params(cid)[1]
    B = np.concatenate([x[1] for x in ld]).reshape(-1, k)
    D = np.stack([x[1] for x in bb])
    L = np.concatenate([x[1] for x in llr]).reshape(-1, n)
    pk = ts_out[:ts_out.size // 188 * 188].reshape(-1, 188)
    return {"in_sha": np.array(rc.sha(np.stack(cells))), "emitted_after_frame": np.array(after, np.int32), "llr_crc": rc.crc_rows(L),
            "ldpc_crc": rc.crc_rows(B), "bb_crc": rc.crc_rows(D), "bb_plp": np.array([x[0][1] for x in bb], np.int32),
            "sent_bbframes_crc": rc.crc_rows(sent), "ts_len": np.int64(ts_out.size), "ts_packet_crc": rc.crc_rows(pk),
            "ts_head": ts_out[:188 * 4].copy(), "ts": ts_out}


def case_bbdh(_):
    """bb_de_header::execute alone on hand-made BBFRAMEs: HEM and NM streams, packets straddling frames, a frame for another
    PLP, a broken header CRC, SYNCD = 0xFFFF, a SYNCD that disagrees with the running packet (both directions)."""
    import bbdh_cases
    out = {}
    tmp = tempfile.mkdtemp()
    for name, (k_bch, frames, plps) in bbdh_cases.cases().items():
        r = ol.RefBbdh(os.path.join(tmp, name + ".ts"), 0, 2)
        for bits, plp in zip(frames, plps):
            r.execute(plp, bits)
        out["ts_" + name] = r.ts()
        out["msg_" + name] = np.array(r.messages())
    return out


def case_front(_):
    i16, q16, loops = rc.front_case()
    tmp = tempfile.mkdtemp()
    r = ol.RefRx(os.path.join(tmp, "f.ts"))
    for w in range(6):
        r.keep(w, False)
    r.set_loops(loops["c1"], loops["c2"], loops["phase_est_filtered"], loops["frequency_est_filtered"], 0.0, 0.0)
    r.sig[5] = 1
    r.sig[2] = 1
    r.execute(i16, q16)
    st = r.state()
    n = len(i16)
    return {"in_sha": np.array(rc.sha(np.stack([i16, q16]))), "derotated": r.buffer(0, n), "decimated": r.buffer(1, n),
            "state": np.array([st[k] for k in ol.RX_STATE], np.float64)}


def case_rx(_):
    m, i16, q16, buf, marks = rc.rx_stream()
    tmp = tempfile.mkdtemp()
    r = ol.RefRx(os.path.join(tmp, "rx.ts"))
    for w in (0, 1, 2, 4):
        r.keep(w, False)
    log = r.run_recording(i16, q16, buf)
    bb = r.taps(3)
    msgs = [bytes(b).decode() for _, b in r.taps(5)]
    ts = r.ts()
    pk = ts[:ts.size // 188 * 188].reshape(-1, 188)
    found = [f for f in range(len(marks)) if ts.tobytes().find(marks[f]) >= 0]
    return {"in_sha": np.array(rc.sha(np.stack([i16, q16]))), "buf": np.int32(buf), "ts_len": np.int64(ts.size), "ts_packet_crc": rc.crc_rows(pk),
            "ts_head": ts[:188 * 8].copy(), "frames_found": np.array(found, np.int32), "bbframes": np.int32(len(bb)),
            "bb_crc": rc.crc_rows(np.stack([x[1] for x in bb])), "messages": np.array(msgs),
            "state_log": np.array([[s[k] for k in ol.RX_STATE] for s in log], np.float64)}


def case_rxoff(name):
    """The reference's whole receiver, every loop closed, on a recording with a carrier offset behind an emulated tuner and with a receiver
    clock that is off (tests/ref_cases.py, RX_OFFSET_CASES): the tuner moves it asked for, the loop trajectory symbol by symbol (private
    members read when replace_null_indicator is emitted, dvbt2_demodulator.cpp:429-444), the state after every execute(), a sample of
    every TI block's cells, the BBFRAMEs and the TS."""
    c, m, bi, bq, sent = rc.rx_offset_case(name)
    tmp = tempfile.mkdtemp()
    r = ol.RefRx(os.path.join(tmp, "rx.ts"), sample_rate=rc.rx_offset_sample_rate(c))
    for w in (1, 2, 4):
        r.keep(w, False)
    log, moves = r.run_recording_tuned(bi, bq, c["buf"], c["cfo_hz"], c["tuner_step"])
    traj = r.traj()
    ti = r.taps(0)
    bb = r.taps(3)
    msgs = [bytes(b).decode() for _, b in r.taps(5)]
    ts = r.ts()
    pk = ts[:ts.size // 188 * 188].reshape(-1, 188)
    want = {bytes(p) for f in sent for p in f}
    sent_ok = np.array([bytes(p) in want for p in pk])
    step = rc.RX_OFFSET_TI_STEP
    states = np.array([[s[k] for k in ol.RX_STATE] for s in log], np.float64)
    col = {k: i for i, k in enumerate(ol.RX_STATE)}
    # dvbt2.fft_size / guard_interval_size are uninitialised memory in the reference until the first decoded P1 (init_dvbt2): not data
    states[states[:, col["p2_init"]] == 0, col["guard_interval_size"]] = 0
    states[states[:, col["p2_init"]] == 0, col["fft_size"]] = 0
    return {"base_sha": np.array(rc.sha(np.stack([bi, bq]))), "moves": np.array(moves, np.float64).reshape(-1, 3), "traj": traj,
            "state_log": states,
            "ti_meta": np.array([meta[:2] for meta, _ in ti], np.int32).reshape(-1, 2), "ti_sample": np.stack([cells[::step] for _, cells in ti]),
            "bbframes": np.int32(len(bb)), "bb_crc": rc.crc_rows(np.stack([x[1] for x in bb])), "messages": np.array(msgs),
            "ts_len": np.int64(ts.size), "ts_packet_crc": rc.crc_rows(pk), "ts_packets_sent": np.int64(sent_ok.sum())}


KINDS = {"sym": (case_sym, list(rc.SYM_MODES)), "p1": (case_p1, ["p1"]), "fec": (case_fec, list(rc.FEC_CASES)), "bbdh": (case_bbdh, ["bbdh"]),
         "ldpc_in": (case_ldpc_in, list(rc.LDPC_IN_CASES)), "carry": (case_carry, list(rc.CARRY_CASES)),
         "front": (case_front, ["front"]), "rx": (case_rx, ["rx"]), "rxoff": (case_rxoff, list(rc.RX_OFFSET_CASES))}
FILES = {"t2sym_golden.npz": ("sym", "p1"), "t2fec_golden.npz": ("fec", "bbdh"), "t2rx_golden.npz": ("front", "rx", "rxoff"),
         "t2batch_golden.npz": ("ldpc_in", "carry")}


def run_case(kind, name, path):
    out = KINDS[kind][0](name)
    np.savez(path, **out)


def run_case_subprocess(kind, name):
    """kind:name in a fresh process -> dict of arrays (also used by the CPU tier's live comparisons)."""
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "case.npz")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--case", "%s:%s" % (kind, name), "--out", path])
        with np.load(path) as z:
            return {k: z[k] for k in z.files}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case")
    ap.add_argument("--out")
    ap.add_argument("--only", help="regenerate one fixture file")
    a = ap.parse_args()
    if a.case:
        kind, name = a.case.split(":")
        run_case(kind, name, a.out)
        sys.exit(0)
    for fname, kinds in FILES.items():
        if a.only and a.only != fname:
            continue
        merged = {}
        for kind in kinds:
            for name in KINDS[kind][1]:
                for k, v in run_case_subprocess(kind, name).items():
                    merged["%s/%s/%s" % (kind, name, k)] = v
                print("  ", kind, name, "done")
        np.savez_compressed(os.path.join(HERE, fname), **merged)
        print("wrote", fname, "%.2f MB" % (os.path.getsize(os.path.join(HERE, fname)) / 1e6), len(merged), "arrays")
