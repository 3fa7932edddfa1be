#!/usr/bin/env python3
"""Where a one-frame call's time goes in the overlap mode with and without the library's host end: host wall time inside
execute_dev per call (median / sum) against the elapsed time of the run. Run on the GPU box. Arguments: frames-per-call [--ts|--ts-nol1]."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import bench
from sdr_receiver_dvb_t2_amd.receiver import t2_rx

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = sys.argv[2] if len(sys.argv) > 2 else ""
if os.environ.get("PROBE_STREAM_FIRST"):
    dummies = []
    for _ in range(int(os.environ.get("PROBE_DUMMY", "0"))):        # streams made (and used once) ahead of the caller's: they take the first hardware queues
        d = torch.cuda.Stream()
        with torch.cuda.stream(d):
            torch.zeros(16, device="cuda")
        dummies.append(d)
    torch.cuda.set_stream(torch.cuda.Stream())          # as bench.py: before anything touches the device
    torch.zeros(16, device="cuda")
w = bench.Workload(bench.CONFIGS[3])
ui, uq, _ = bench.make_frames(w, 2, 21.0, seed=20250614)
F = 48
di = torch.from_numpy(np.concatenate([ui] * (F // 2)).reshape(-1)).cuda()
dq = torch.from_numpy(np.concatenate([uq] * (F // 2)).reshape(-1)).cuda()
FS = w.frame_samples
if not os.environ.get("PROBE_NULL_STREAM") and not os.environ.get("PROBE_STREAM_FIRST"):
    torch.cuda.set_stream(torch.cuda.Stream())          # not the legacy null stream (PROBE_NULL_STREAM=1: on it)
if os.environ.get("PROBE_PRELUDE"):              # what bench.py has done before its sweep: a 48-frame handle with the host end, used and closed
    r0 = t2_rx(*w.mode, w.lps, *w.plp, w.nb, max_frames=F)
    r0.ts_enable(0, l1_check=True)
    for k in range(int(os.environ["PROBE_PRELUDE"])):
        r0.execute_dev(di, dq, F, first_call=(k == 0))
    torch.cuda.synchronize()
    r0.ts_read(wait_all=True)
    r0.close()
rx = t2_rx(*w.mode, w.lps, *w.plp, w.nb, max_frames=nf)
if mode.startswith("--ts"):
    rx.ts_enable(0, l1_check=(mode == "--ts"))
rx.execute_dev(di, dq, nf, first_call=True)
torch.cuda.synchronize()
if rx.carry:
    rx.flush_dev()
    torch.cuda.synchronize()
rx.set_overlap(True)
level = rx.results(nf)["level_detect"]
calls = int(os.environ.get("PROBE_CALLS", 96 // nf))
for _ in range(int(os.environ.get("PROBE_WARM", 8 // nf + 1))):
    rx.execute_dev(di, dq, nf, level_detect=level)
rx.wait(); torch.cuda.synchronize()
per = []
consumer = None
if os.environ.get("PROBE_CONSUMER"):             # bench.py's TS consumer: a Python thread that polls the host end while the calls are made
    import threading
    done = threading.Event()
    sink = np.empty(2 * 202 * 6000 * nf, np.uint8)
    nap = float(os.environ["PROBE_CONSUMER"])

    def consume():
        while not done.is_set():
            if rx.ts_read_into(sink) == 0:
                time.sleep(nap)
    consumer = threading.Thread(target=consume)
    consumer.start()
t0 = time.perf_counter()
for c in range(calls):
    a = 0 if os.environ.get("PROBE_FIXED") else (c * nf) % (F - nf + 1)
    t1 = time.perf_counter()
    rx.execute_dev(di[a * FS:], dq[a * FS:], nf, level_detect=level)
    per.append(time.perf_counter() - t1)
t2 = time.perf_counter()
if consumer:
    done.set(); consumer.join()
rx.wait(); torch.cuda.synchronize()
t3 = time.perf_counter()
if mode.startswith("--ts"):
    rx.ts_read(wait_all=True)
el = time.perf_counter() - t0
per = np.array(per) * 1e3
print("%s nf=%d: %.0f Msamples/s; elapsed %.2f ms, calls' host time sum %.2f ms (median %.3f, min %.3f, max %.3f), final wait %.2f ms"
      % (mode or "off", nf, calls * nf * FS / el / 1e6, el * 1e3, per.sum(), np.median(per), per.min(), per.max(), (t3 - t2) * 1e3))
print("  per call ms:", " ".join("%.2f" % x for x in per[:24]))
rx.close()
