#!/usr/bin/env python3
"""One run of THE REFERENCE'S receiver with the GPU-bound slot bodies of integration/*_gpu.cpp, in a process of its own (the reference keeps
function-local statics): oracle/_ref/libref_t2rx_gpufec.so (llr_demapper, ldpc_decoder, bch_decoder on the GPU behind the reference's CPU
demodulator) or libref_t2rx_gpu.so (all six slots) on a closed-loop case of tests/ref_cases.py, behind the emulated tuner with the moves
the fixture recorded. Writes the TS packet CRCs, the BBFRAME count and the messages to --out (npz). Test infrastructure."""
import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_lib as ol  # noqa: E402
import ref_cases as rc  # noqa: E402
import t2_tx  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", required=True)
    ap.add_argument("--case", required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    with np.load(os.path.join(HERE, "golden", "t2rx_golden.npz")) as z:
        moves = z["rxoff/%s/moves" % a.case]
    c, m, bi, bq, sent = rc.rx_offset_case(a.case)
    i16, q16 = t2_tx.rx_offset_tuned(bi, bq, c["buf"], c["cfo_hz"], [(int(k), float(t)) for k, _, t in moves])
    tmp = tempfile.mkdtemp()
    r = ol.RefRx(os.path.join(tmp, "rx.ts"), sample_rate=rc.rx_offset_sample_rate(c), lib=a.lib)
    for w in (0, 1, 2, 4):
        r.keep(w, False)
    # rx_sdrplay::start's loop over the pre-tuned recording: a re-tune request is answered with the move the fixture lists for that buffer
    s = r.sig
    rf = [626.0e6]
    listed = {int(k): float(req) for k, req, _ in moves}
    asked = []

    def reset():
        s[7] = 0; s[1] = 0.0; s[0] = 1; s[6] = 0.0; s[4] = 0; s[3] = 1
        rf[0] = 626.0e6

    def set_rf(k):
        if not s[2]:
            s[2] = 1
        if s[0]:
            s[0] = 0; s[2] = 0
            if s[1] != 0.0:
                asked.append((k, float(s[1])))
                s[1] = listed.get(k, s[1])
            s[6] = s[1] / rf[0]
            rf[0] += s[1]
    reset()
    set_rf(0)
    buf = c["buf"]
    for k in range(len(i16) // buf):
        if s[7]:
            reset(); set_rf(k)
            continue
        set_rf(k)
        s[5] = 1
        r.execute(i16[k * buf:(k + 1) * buf], q16[k * buf:(k + 1) * buf])
    import time
    time.sleep(1.0)                                       # the stages' own threads finish what is queued
    bb = r.taps(3)
    msgs = [bytes(b).decode() for _, b in r.taps(5)]
    ts = r.ts()
    pk = ts[:ts.size // 188 * 188].reshape(-1, 188)
    np.savez(a.out, ts_len=np.int64(ts.size), ts_packet_crc=rc.crc_rows(pk), bbframes=np.int32(len(bb)), messages=np.array(msgs),
             asked=np.array(asked, np.float64).reshape(-1, 2))


if __name__ == "__main__":
    main()
