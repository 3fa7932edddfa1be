#!/usr/bin/env python3
"""Generate tests/golden/ldpc_golden.npz from the REFERENCE's own LDPC decoder (oracle/_ref/libref_ldpc.so, i.e.
/root/reference/src/DVB_T2/LDPC/*.hh compiled unmodified by oracle/Makefile). Run in the build container:

    make -C oracle && python tests/golden/make_ldpc_golden.py

Each case stores the int8 input LLRs (the data), and what the reference produced: trials-left, packed hard bits and a
SHA-256 of all final a-posteriori LLRs. Cases cover all twelve codes the reference instantiates
(ldpc_decoder.cpp:85-111), a batch that needs many updates, and a batch the reference gives up on (-1).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

# (name, code id, frames, sigma, seed)
CASES = [
    ("S1_2", 0, 3, 0.88, 1), ("S3_5", 1, 2, 0.74, 2), ("S2_3", 2, 2, 0.66, 3), ("S3_4", 3, 2, 0.62, 4),
    ("S4_5", 4, 2, 0.56, 5), ("S5_6", 5, 2, 0.50, 6),
    ("N1_2", 6, 1, 0.86, 7), ("N3_5", 7, 1, 0.74, 8), ("N2_3", 8, 1, 0.66, 9), ("N3_4", 9, 2, 0.615, 10),
    ("N4_5", 10, 1, 0.555, 11), ("N5_6", 11, 1, 0.51, 12),
    ("S1_2_fail", 0, 2, 1.05, 13), ("N3_4_fail", 9, 1, 0.70, 14),
]


def main():
    assert ol.ref() is not None, "oracle/_ref/libref_ldpc.so missing (needs /root/reference)"
    out = {}
    for name, cid, frames, sigma, seed in CASES:
        info, llr = ol.make_llr(cid, frames, sigma, seed)
        t, bits, lo = ol.ref_decode(cid, llr)
        n, k, _, _ = ol.ldpc_params(cid)
        hard = (lo[:, :k] < 0).astype(np.uint8)
        out[name + "/cid"] = np.int32(cid)
        out[name + "/llr"] = llr
        out[name + "/trials_left"] = np.int32(t)
        out[name + "/hard"] = np.packbits(hard, axis=1)
        out[name + "/llr_sha256"] = np.frombuffer(hashlib.sha256(lo.tobytes()).digest(), dtype=np.uint8)
        out[name + "/info_ok"] = np.bool_(t >= 0 and np.array_equal(bits, info))
        print(name, "trials_left", t, "decoded==sent", bool(out[name + "/info_ok"]))
    np.savez_compressed(os.path.join(HERE, "ldpc_golden.npz"), **out)


if __name__ == "__main__":
    main()
