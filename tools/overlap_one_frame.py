#!/usr/bin/env python3
"""One-frame calls of the batch receiver with t2gpu_rx_set_overlap (CFG-A, all-fail load as in the bench's headline leg), back to back:
    python tools/overlap_one_frame.py                       # ms per call, Msamples/s
    (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d $REPO/gpurun_out/ov1 -- python $REPO/tools/overlap_one_frame.py)
    python tools/ldpc_overlap.py gpurun_out/ov1             # sum / union of the LDPC launches: how much of them ran side by side
T2GPU_RX_PAIR=0 gives the one-decode-at-a-time form."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from sdr_receiver_dvb_t2_amd.receiver import t2_rx  # noqa: E402

w = bench.Workload(bench.CONFIGS[3])
ui, uq, _ = bench.make_frames(w, 2, 21.0, seed=3)
dev = torch.device("cuda:0")
d_i, d_q = torch.from_numpy(ui.reshape(-1)).to(dev), torch.from_numpy(uq.reshape(-1)).to(dev)
rx = t2_rx(*w.mode, w.lps, *w.plp, w.nb, max_frames=1, ldpc_trials=25, device=0)
FS = w.frame_samples
rx.execute_dev(d_i[:FS], d_q[:FS], 1, 0.03, first_call=True)
rx.flush_dev()
torch.cuda.synchronize()
rx.set_overlap(True)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 24
    for k in range(n):
        a = (k % 2) * FS
        rx.execute_dev(d_i[a:a + FS], d_q[a:a + FS], 1, 0.03, first_call=False)
    rx.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("overlapped one-frame calls (T2GPU_RX_PAIR=%s): %.3f ms per call, %.1f Msamples/s" % (os.environ.get("T2GPU_RX_PAIR", "1"), dt / n * 1e3, n * FS / dt / 1e6))
