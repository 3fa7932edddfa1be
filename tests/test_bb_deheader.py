"""CPU tier: BBFRAME de-framing (host code of the product, csrc/bb_deheader.cpp, restating bb_de_header.cpp:84-448) against
the transmitter model: TS packets packed into high-efficiency-mode BBFRAMEs come back byte for byte, across frame
boundaries, with the reference's drop rules for a bad BBHEADER CRC, a foreign PLP and SYNCD = 65535."""
import ctypes

import numpy as np
import pytest

import t2_tx


@pytest.fixture(scope="module")
def l(built):
    import sdr_receiver_dvb_t2_amd as pkg
    return pkg.lib()


def run(l, h, frame, plp=0, packed=False):
    """packed: the frame goes in as k_bch / 8 bytes, MSB first (t2gpu_bbdh_execute_packed: what K-descramble-pack delivers)."""
    out = np.zeros(frame.size // 8 + 400, np.uint8)
    err = ctypes.c_int(0)
    if packed:
        rows = np.packbits(frame)
        n = l.t2gpu_bbdh_execute_packed(h, plp, frame.size, rows.ctypes.data, out.ctypes.data, out.size, ctypes.byref(err))
    else:
        n = l.t2gpu_bbdh_execute(h, plp, frame.size, frame.ctypes.data, out.ctypes.data, out.size, ctypes.byref(err))
    return n, out[:max(n, 0)], err.value


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("cid", [9, 8, 0, 6])
def test_hem_round_trip(l, cid, packed):
    k_bch = t2_tx.K_BCH[cid]
    n_frames = 9
    ts = t2_tx.ts_packets(n_frames * (k_bch // (187 * 8) + 2), seed=cid)
    frames, used = t2_tx.bbframes_hem(ts, k_bch, n_frames)
    h = l.t2gpu_bbdh_create(0)
    got = []
    for f in range(n_frames):
        n, out, err = run(l, h, np.ascontiguousarray(frames[f]), packed=packed)
        assert n > 0 and err == 0 and l.t2gpu_bbdh_mode(h) == 1
        got.append(out)
    got = np.concatenate(got)
    whole = (got.size // 188) * 188
    assert whole // 188 >= used - 1
    assert np.array_equal(got[:whole], ts.reshape(-1)[:whole])
    l.t2gpu_bbdh_destroy(h)


@pytest.mark.parametrize("packed", [False, True])
def test_reset_starts_a_new_stream(l, packed):
    """t2gpu_bbdh_reset = the packet state of a freshly constructed bb_de_header: after frames of one stream (a packet is left split
    across the frame end), the first frame of ANOTHER stream gives exactly what a new handle gives -- no stale half packet."""
    k_bch = t2_tx.K_BCH[9]
    a, _ = t2_tx.bbframes_hem(t2_tx.ts_packets(120, 11), k_bch, 3)
    b, _ = t2_tx.bbframes_hem(t2_tx.ts_packets(120, 12), k_bch, 2)
    h, fresh = l.t2gpu_bbdh_create(0), l.t2gpu_bbdh_create(0)
    for f in range(3):
        assert run(l, h, np.ascontiguousarray(a[f]), packed=packed)[0] > 0
    stale = run(l, h, np.ascontiguousarray(b[0]), packed=packed)[1]          # without a reset: the old stream's tail leads
    assert l.t2gpu_bbdh_reset(h) == 0
    for f in range(2):
        n0, want, _ = run(l, fresh, np.ascontiguousarray(b[f]), packed=packed)
        n1, got, _ = run(l, h, np.ascontiguousarray(b[f]), packed=packed)
        assert n0 == n1 and np.array_equal(got, want)
        if f == 0:
            assert not np.array_equal(got, stale)
    l.t2gpu_bbdh_destroy(h); l.t2gpu_bbdh_destroy(fresh)


def test_drop_rules(l):
    k_bch = t2_tx.K_BCH[9]
    frames, _ = t2_tx.bbframes_hem(t2_tx.ts_packets(80, 1), k_bch, 2)
    h = l.t2gpu_bbdh_create(0)
    bad = frames[0].copy(); bad[13] ^= 1                         # BBHEADER bit error -> CRC-8 residue neither 0 nor 0xAB
    assert run(l, h, bad)[0] == -1
    assert run(l, h, np.ascontiguousarray(frames[0]), plp=3)[0] == -2        # not the selected PLP
    nosync = frames[0].copy()
    hdr = list(nosync[:56]) + t2_tx.bits_of(65535, 16)
    nosync[:80] = hdr + t2_tx.bits_of(t2_tx.crc8_d5(hdr) ^ 1, 8)
    assert run(l, h, nosync)[0] == -2                             # SYNCD = 65535: no packet starts here
    n, out, _ = run(l, h, np.ascontiguousarray(frames[0]))
    assert n > 0 and out[0] == 0x47
    l.t2gpu_bbdh_destroy(h)


# ---- against the reference's own class (tests/golden/t2fec_golden.npz "bbdh/...", made by tests/golden/make_t2_golden.py) ----
import os

import bbdh_cases


@pytest.fixture(scope="module")
def gold():
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t2fec_golden.npz")) as z:
        return {k[len("bbdh/bbdh/"):]: z[k] for k in z.files if k.startswith("bbdh/bbdh/")}


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("name", list(bbdh_cases.cases()))
def test_equals_the_reference_class(l, gold, name, packed):
    """Every TS byte bb_de_header::execute wrote for these BBFRAME sequences -- high-efficiency and normal mode (the latter with
    the reference's DFL slip), lost frames (both re-synchronisation branches), a frame of another PLP, a broken header CRC,
    SYNCD = 0xFFFF, a multiple-input-stream header -- and its message counts."""
    k_bch, frames, plps = bbdh_cases.cases()[name]
    h = l.t2gpu_bbdh_create(0)
    got, rcs, resync, errs = [], [], 0, 0
    for bits, plp in zip(frames, plps):
        n, out, err = run(l, h, np.ascontiguousarray(bits), plp, packed=packed)
        rcs.append(n)
        if n >= 0:
            got.append(out)
            resync += l.t2gpu_bbdh_resync_count(h)
            errs += 1 if err else 0
    l.t2gpu_bbdh_destroy(h)
    got = np.concatenate(got) if got else np.zeros(0, np.uint8)
    assert np.array_equal(got, gold["ts_" + name])
    msgs = [str(s) for s in gold["msg_" + name]]
    assert resync == sum("resynchronizing" in s for s in msgs)
    assert errs == sum(s == "TS error." for s in msgs)
    assert rcs.count(-1) == sum("CRC8" in s for s in msgs)


def test_wild_syncd_and_dfl_stay_inside_the_buffers(l):
    """SYNCD is 16 bits the reference trusts. A CRC-valid header with SYNCD = 60000 after a pending split packet asks
    for 7500 bytes from a 7032-bit frame: reads stay inside the frame (zeros beyond), writes inside out_cap (the frame is refused
    with -3 when they would not fit), and the next good frame decodes again. Run under guard bytes on both sides."""
    k_bch = bbdh_cases.K_BCH
    ts = t2_tx.ts_packets(60, 5)
    for maker, hem in ((bbdh_cases.hem_frames, True), (bbdh_cases.nm_frames, False)):
        fr = maker(ts, 4)
        wild = fr[1].copy()
        dfl = int("".join(str(b) for b in wild[32:48]), 2)
        wild[:80] = bbdh_cases._header(dfl, 60000, hem, upl=(0 if hem else 1504), sync=(0 if hem else 0x47))
        big = fr[1].copy()
        big[:80] = bbdh_cases._header(65528, int("".join(str(b) for b in big[56:72]), 2), hem, upl=(0 if hem else 1504), sync=(0 if hem else 0x47))
        h = l.t2gpu_bbdh_create(0)
        guard = 4096
        for frame in (fr[0], wild, fr[2], big, fr[3]):
            inbuf = np.full(guard + k_bch + guard, 0xA5, np.uint8)
            inbuf[guard:guard + k_bch] = frame
            cap = k_bch // 8 + 400
            outbuf = np.full(guard + cap + guard, 0x5A, np.uint8)
            err = ctypes.c_int(0)
            n = l.t2gpu_bbdh_execute(h, 0, k_bch, inbuf[guard:].ctypes.data, outbuf[guard:].ctypes.data, cap, ctypes.byref(err))
            assert n <= cap
            assert (outbuf[:guard] == 0x5A).all() and (outbuf[guard + cap:] == 0x5A).all()
            if n > 0:                                              # nothing of the guard pattern was read into the output
                assert not np.array_equal(outbuf[guard:guard + 8], np.full(8, 0xA5, np.uint8))
        l.t2gpu_bbdh_destroy(h)
