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


def run(l, h, frame, plp=0):
    out = np.zeros(frame.size // 8 + 400, np.uint8)
    err = ctypes.c_int(0)
    n = l.t2gpu_bbdh_execute(h, plp, frame.size, frame.ctypes.data, out.ctypes.data, out.size, ctypes.byref(err))
    return n, out[:max(n, 0)], err.value


@pytest.mark.parametrize("cid", [9, 8, 0, 6])
def test_hem_round_trip(l, cid):
    k_bch = t2_tx.K_BCH[cid]
    n_frames = 9
    ts = t2_tx.ts_packets(n_frames * (k_bch // (187 * 8) + 2), seed=cid)
    frames, used = t2_tx.bbframes_hem(ts, k_bch, n_frames)
    h = l.t2gpu_bbdh_create(0)
    got = []
    for f in range(n_frames):
        n, out, err = run(l, h, np.ascontiguousarray(frames[f]))
        assert n > 0 and err == 0 and l.t2gpu_bbdh_mode(h) == 1
        got.append(out)
    got = np.concatenate(got)
    whole = (got.size // 188) * 188
    assert whole // 188 >= used - 1
    assert np.array_equal(got[:whole], ts.reshape(-1)[:whole])
    l.t2gpu_bbdh_destroy(h)


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
