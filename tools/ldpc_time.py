import ctypes, sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import sdr_receiver_dvb_t2_amd as pkg
import oracle_lib as ol
frames = 7680
rng = np.random.default_rng(1)
llr = rng.integers(-20, 21, size=(256, 64800), dtype=np.int8)
llr = np.tile(llr, (frames // 256, 1))
x = torch.from_numpy(np.ascontiguousarray(llr)).cuda()
dec = pkg.ldpc_decoder(1, 3, max_frames=frames, group=32)
for it in range(4):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); bits, trials = dec.execute_dev(x); t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("iter", it, "host call %.3f ms, events %.3f ms, wall %.3f ms" % ((t1 - t0) * 1e3, e0.elapsed_time(e1), (t2 - t0) * 1e3), "trials", int(trials[0]))
