"""K-bch on the GPU through the C ABI (t2gpu_bch_decode / _dev) against oracle/bch_oracle.c: same corrected bits and the same
status for clean frames, up to t errors anywhere in the frame, and more than t errors. Opt-in stage: the
reference's bch_decoder::execute does not decode (bch_decoder.cpp:136)."""
import numpy as np
import pytest

import oracle_lib as ol
import t2_tx
from sdr_receiver_dvb_t2_amd.fec import bch_decoder

pytestmark = pytest.mark.gpu


def _codewords(cid, n, rng):
    _, _, kb, nb = ol.bch_params(cid)
    msg = rng.integers(0, 2, (n, kb), dtype=np.uint8)
    cw = np.zeros((n, nb), np.uint8)
    cw[:, :kb] = msg
    cw[:, kb:] = t2_tx.bch_parity(cid, msg)
    return cw


@pytest.mark.parametrize("cid", range(12))
def test_every_code_against_the_oracle(cid):
    m, t, kb, nb = ol.bch_params(cid)
    rng = np.random.default_rng(500 + cid)
    cw = _codewords(cid, 9, rng)
    bad = cw.copy()
    plan = [[], [0], [nb - 1], [0, nb - 1] + list(rng.choice(np.arange(1, nb - 1), t - 2, replace=False)),
            list(rng.choice(nb, t, replace=False)), list(rng.choice(nb, t + 1, replace=False)),
            list(rng.choice(nb, t + 5, replace=False)), list(range(kb - 2, kb + 4)), list(rng.choice(nb, 3 * t, replace=False))]
    for f, pos in enumerate(plan):
        bad[f, pos] ^= 1
    want, want_st = ol.ora_bch_decode(cid, bad)
    got, got_st = bch_decoder(cid // 6, cid % 6).correct(bad)
    assert list(got_st) == list(want_st)
    assert (got == want).all()
    assert list(got_st[:5]) == [0, 1, 1, t, t] and got_st[7] == 6 and (got[[0, 1, 2, 3, 4, 7]] == cw[[0, 1, 2, 3, 4, 7]]).all()
    assert (got[got_st < 0] == bad[got_st < 0]).all()                       # beyond t: reported, frame left as received


@pytest.mark.parametrize("cid", [4, 8, 9])
def test_a_buffer_of_frames_on_the_device(cid):
    """More frames than CUs, error counts 0..t mixed through the launch: every frame comes back as sent, status = its error count;
    a second pass over the corrected buffer reports clean frames (idempotence)."""
    import torch
    m, t, kb, nb = ol.bch_params(cid)
    rng = np.random.default_rng(900 + cid)
    n = 600
    cw = np.repeat(_codewords(cid, 24, rng), n // 24, axis=0)
    cw ^= np.roll(cw, 7, axis=0)                                            # linear code: sums of codewords are codewords
    errs = rng.integers(0, t + 1, n)
    bad = cw.copy()
    for f in range(n):
        bad[f, rng.choice(nb, errs[f], replace=False)] ^= 1
    dec = bch_decoder(cid // 6, cid % 6)
    d = torch.from_numpy(bad).cuda()
    st = dec.correct_dev(d)
    torch.cuda.synchronize()
    assert (st.cpu().numpy() == errs).all()
    assert (d.cpu().numpy() == cw).all()
    st2 = dec.correct_dev(d)
    assert not st2.cpu().numpy().any() and (d.cpu().numpy() == cw).all()
    # then the reference's stage proper: parity strip + descrambler on the corrected words
    out = dec.execute_dev(d).cpu().numpy()
    assert (out == ol.ora_bch_descramble(cid, cw)).all()


def test_more_frames_than_workgroups():
    """9000 short frames in one call (the launch caps its grid at 8192 workgroups and strides): sparse damage, everything repaired."""
    import torch
    cid = 1
    m, t, kb, nb = ol.bch_params(cid)
    rng = np.random.default_rng(77)
    n = 9000
    base = _codewords(cid, 8, rng)
    cw = base[rng.integers(0, 8, n)] ^ base[rng.integers(0, 8, n)]
    errs = np.where(rng.random(n) < 0.1, rng.integers(1, t + 1, n), 0)
    bad = cw.copy()
    for f in np.nonzero(errs)[0]:
        bad[f, rng.choice(nb, errs[f], replace=False)] ^= 1
    d = torch.from_numpy(bad).cuda()
    st = bch_decoder(0, 1).correct_dev(d)
    assert (st.cpu().numpy() == errs).all() and (d.cpu().numpy() == cw).all()


def test_bad_arguments_are_refused():
    import torch
    from sdr_receiver_dvb_t2_amd._lib import lib
    d = torch.zeros(7200 + 8, dtype=torch.uint8, device="cuda")
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    l = lib()
    assert l.t2gpu_bch_decode_dev(0, 0, d.data_ptr() + 1, 1, st.data_ptr(), None) == -1      # not 8-byte aligned
    assert l.t2gpu_bch_decode_dev(0, 7, d.data_ptr(), 1, st.data_ptr(), None) == -1          # no such code
    assert l.t2gpu_bch_decode_dev(0, 0, d.data_ptr(), 0, st.data_ptr(), None) == -1
    assert l.t2gpu_bch_decode_dev(0, 0, d.data_ptr(), 1, st.data_ptr(), None) == 1 and int(st[0]) == 0   # all-zero word: a codeword
