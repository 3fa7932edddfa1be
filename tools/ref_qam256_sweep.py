#!/usr/bin/env python3
"""Shows that the REFERENCE's own 256-QAM demapper never hands its LDPC stage a decodable batch (build container only: needs
oracle/_ref/libref_t2rx.so). For each SNR: 34 CFG-A FEC blocks (256-QAM rotated, 64800 r=3/4) through the reference's
time_deinterleaver -> llr_demapper -> ldpc_decoder objects; prints LDPC batches handed over, batches decoded, and the share of LLRs
with |L| >= 120 (the truncating int8 cast of llr_demapper.cpp:722-737 wraps the outer constellation points because the
hard-decision SNR estimate, :564-676, saturates). One process per SNR (the demapper keeps function-local statics).

    python tools/ref_qam256_sweep.py            # 19.5 ... 24 dB
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(snr):
    import numpy as np
    import oracle_lib as ol
    import ref_cases as rc
    rc.FEC_CASES["sweep"] = (3, 1, 3, 34, snr, 2)
    q, frames, ts, l1 = rc.fec_case("sweep")
    r = ol.RefFec(os.path.join(tempfile.mkdtemp(), "ref.ts"), 0)
    r.start(rc.FEC_L1_POST_SIZE, l1)
    r.frame(l1, np.concatenate([np.zeros(1840 + rc.FEC_L1_POST_SIZE, np.complex64), rc.dequantise(q, rc.GRID_CELL)]))
    llr, ld = r.taps(1), r.taps(2)
    big = float(np.mean(np.abs(np.concatenate([x[1] for x in llr]).astype(np.int32)) >= 120)) if llr else float("nan")
    print("SNR %.1f dB: LLR batches %d, decoded %d, |L|>=120: %.1f %%" % (snr, len(llr), len(ld), 100 * big))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(float(sys.argv[1]))
    else:
        for snr in (19.5, 20.0, 20.5, 21.0, 21.5, 22.0, 23.0, 24.0):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), str(snr)])
