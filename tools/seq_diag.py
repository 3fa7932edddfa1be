import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import sdr_receiver_dvb_t2_amd as pkg
from test_fec_gpu import qam_cells
mod, nb = 3, 202
n = nb * 8100
dm = pkg.llr_demapper(mod, 1, 3, 1, max_cells=n)
frames = 48
cells = np.stack([qam_cells(mod, n, 22.0, seed=f, rotation=1) for f in range(2)])
x = torch.from_numpy(np.ascontiguousarray(np.tile(cells, (frames // 2, 1)).view(np.float32).reshape(frames, n, 2))).cuda()
sums = torch.zeros((frames, 8), dtype=torch.float32, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dm.stats_batch_dev(x, sums); e1.record(); torch.cuda.synchronize()
    print("stats_batch %.3f ms" % e0.elapsed_time(e1), sums[0].cpu().numpy(), sums[1].cpu().numpy())
