"""Time K-bch on a bench-sized buffer (3232 normal r=3/4 frames): clean frames, and frames with t errors each."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as ol, t2_tx
from sdr_receiver_dvb_t2_amd.fec import bch_decoder

cid, n = 9, 3232
m, t, kb, nb = ol.bch_params(cid)
rng = np.random.default_rng(1)
msg = rng.integers(0, 2, (16, kb), dtype=np.uint8)
cw = np.zeros((16, nb), np.uint8); cw[:, :kb] = msg; cw[:, kb:] = t2_tx.bch_parity(cid, msg)
cw = np.tile(cw, (n // 16, 1))
dec = bch_decoder(1, 3)
for label, nerr in (("clean", 0), ("t errors", t)):
    bad = cw.copy()
    for f in range(n):
        bad[f, rng.choice(nb, nerr, replace=False)] ^= 1
    src = torch.from_numpy(bad).cuda()
    d = src.clone(); dec.correct_dev(d); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        d.copy_(src); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); st = dec.correct_dev(d); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    assert (d.cpu().numpy() == cw).all() and (st.cpu().numpy() == nerr).all()
    print(label, "ms per %d frames:" % n, round(min(ts), 3), "->", round(n * nb / min(ts) / 1e6, 1), "GB/s of bit bytes")
