#!/usr/bin/env python3
"""Throughput of the batch receiver at 1 / 2 / 4 T2 frames per call (CFG-A) with the decode overlapped (t2gpu_rx_set_overlap) for several
values of the collect policy's K (T2GPU_RX_COLLECT: SIMD batches that collect before a decode is launched = the decode's resident slots;
0 = a decode per call, round 5's form). Calls back to back, the host waits once behind the last. Run on the GPU box."""
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

w = bench.Workload(bench.CONFIGS[3])
ui, uq, _ = bench.make_frames(w, 2, 21.0, seed=20250614)
F = 48
di = torch.from_numpy(np.concatenate([ui] * (F // 2)).reshape(-1)).cuda()
dq = torch.from_numpy(np.concatenate([uq] * (F // 2)).reshape(-1)).cuda()
FS = w.frame_samples
if not os.environ.get("PROBE_NULL_STREAM"):
    torch.cuda.set_stream(torch.cuda.Stream())          # not the legacy null stream (PROBE_NULL_STREAM=1: on it)
TS = len(sys.argv) > 1 and sys.argv[1].startswith("--ts")
L1 = not (len(sys.argv) > 1 and sys.argv[1] == "--ts-nol1")             # with the library's host end (L1 parse, drop rule, de-framer) on
print("host end", ("on" if L1 else "on, without the per-frame L1 check") if TS else "off")
for nf in [int(x) for x in os.environ.get("SWEEP_NF", "1,2,4").split(",")]:
    row = []
    for K in [int(x) for x in os.environ.get("SWEEP_K", "0,12,13,14").split(",")]:
        os.environ["T2GPU_RX_COLLECT"] = str(K)
        rx = t2_rx(*w.mode, w.lps, *w.plp, w.nb, max_frames=nf)
        if TS:
            rx.ts_enable(0, l1_check=L1)
        rx.execute_dev(di, dq, nf, first_call=True)
        torch.cuda.synchronize()
        if rx.carry:
            rx.flush_dev()
            torch.cuda.synchronize()
        rx.set_overlap(True)
        level = rx.results(nf)["level_detect"]
        calls = max(6, 96 // nf)
        for _ in range(8 // nf + 1):
            rx.execute_dev(di, dq, nf, level_detect=level)
        rx.wait(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(calls):
            a = (c * nf) % (F - nf + 1)
            rx.execute_dev(di[a * FS:], dq[a * FS:], nf, level_detect=level)
        rx.wait(); torch.cuda.synchronize()
        if TS:
            rx.ts_read(wait_all=True)
        el = time.perf_counter() - t0
        row.append("K=%d %.0f" % (K, calls * nf * FS / el / 1e6))
        rx.close()
    print("%d frame(s) per call, Msamples/s: %s" % (nf, "  ".join(row)))
