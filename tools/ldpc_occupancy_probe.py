#!/usr/bin/env python3
"""LDPC-only timing of one code with all-noise input (every frame runs all sweeps), frames stopping individually (group 1):
how throughput moves with the workgroups per CU (T2GPU_LDPC_BLOCKS_PER_CU) for the register budget the library was built with."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdr_receiver_dvb_t2_amd as pkg
fec_type, cod, frames = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = 64800 if fec_type else 16200
rng = np.random.Generator(np.random.PCG64(3))
llr = rng.integers(-20, 21, size=(256, n), dtype=np.int8)
x = torch.from_numpy(np.tile(llr, ((frames + 255) // 256, 1))[:frames].copy()).cuda()
dec = pkg.ldpc_decoder(fec_type, cod, max_frames=frames, group=1)
dec.execute_dev(x); torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); bits, trials = dec.execute_dev(x); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("fec %d cod %d frames %d blocks/CU %s: %.3f ms (min of 3), %.1f kframes/s, avg sweeps %.1f" % (
    fec_type, cod, frames, os.environ.get("T2GPU_LDPC_BLOCKS_PER_CU", "default"), min(ts), frames / min(ts), float((25 - trials.float().clamp(min=0)).mean())))
