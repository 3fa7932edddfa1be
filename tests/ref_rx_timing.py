#!/usr/bin/env python3
"""cpu_baseline of bench.py, kind "reference": THE REFERENCE'S OWN RECEIVER (oracle/_ref/libref_t2rx.so = src/DVB_T2/*.cpp compiled with the
reference's flags, its stage objects on their own QThreads, the FFTW binary it ships) timed on this host from int16 I/Q to the transport
stream on the bench's own frames: rx_sdrplay::start's loop hands dvbt2_demodulator::execute buffers of 172 032 samples; the clock starts
when the receiver has acquired (P1, guard interval, L1-pre, L1-post: deint_start) and runs for about --seconds. Prints one JSON line.
Test / bench infrastructure: nothing of the product is involved."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--i", required=True)
    ap.add_argument("--q", required=True)
    ap.add_argument("--frame-samples", type=int, required=True)
    ap.add_argument("--seconds", type=float, default=15.0)
    ap.add_argument("--buf", type=int, default=172032)
    a = ap.parse_args()
    import oracle_lib as ol
    if not ol.RefRx.available():
        print(json.dumps({"error": "oracle/_ref/libref_t2rx.so does not load on this host"}))
        return
    ring_i, ring_q = np.fromfile(a.i, np.int16), np.fromfile(a.q, np.int16)        # whole frames: the ring closes on a frame boundary
    n_ring = ring_i.size
    ring_i, ring_q = np.concatenate([ring_i, ring_i[:a.buf]]), np.concatenate([ring_q, ring_q[:a.buf]])
    tmp = tempfile.mkdtemp()
    r = ol.RefRx(os.path.join(tmp, "rx.ts"))
    for w in range(6):
        r.keep(w, False)
    s = r.sig
    s[7] = 0; s[1] = 0.0; s[0] = 1; s[6] = 0.0; s[4] = 0; s[3] = 1                  # rx_sdrplay.cpp:135-156 reset()
    s[0] = 0; s[2] = 0                                                              # set_rf_frequency with a zero move
    pos, fed, t0, timed, acquired_after = 0, 0, None, 0, None
    t_begin = time.perf_counter()
    while True:
        s[2] = 1; s[5] = 1
        if s[0]:                                                                     # a re-tune request on a recording without offset: acknowledge
            s[0] = 0; s[6] = 0.0
        r.execute(ring_i[pos:pos + a.buf], ring_q[pos:pos + a.buf])
        pos = (pos + a.buf) % n_ring
        fed += a.buf
        if t0 is None:
            if r.state()["deint_start"]:
                t0 = time.perf_counter()
                acquired_after = fed
            elif time.perf_counter() - t_begin > 120:
                print(json.dumps({"error": "the reference did not acquire within 120 s"}))
                return
        else:
            timed += a.buf
            if time.perf_counter() - t0 >= a.seconds:
                break
    el = time.perf_counter() - t0
    st = r.state()
    print(json.dumps({"msamples_per_s": timed / el / 1e6, "seconds": el, "samples": timed, "frames": timed / a.frame_samples,
                      "acquired_after_frames": acquired_after / a.frame_samples, "threads": 6,
                      "guard_interval_size": int(st["guard_interval_size"]), "fft_size": int(st["fft_size"])}))


if __name__ == "__main__":
    main()
