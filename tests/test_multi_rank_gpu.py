"""GPU tier: the N > 1 control flow on the one GPU a test box has. Two ranks share device 0 and talk gloo (RCCL refuses two ranks on
one device; the driver's scaling runs use one GPU per rank and RCCL) -- everything else is the code the 8-GPU run executes:
bench.py's launch contract, shard boundaries, barriers, max-over-ranks timing, and shard.ordered_receiver's gather of the ranks'
BBFRAMEs into ONE de-framer in frame order."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_device_prints_one_line(built):
    env = dict(os.environ, T2GPU_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "16", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "16 T2 frames/GPU/step" in d["config"]["workload"] and d["config"]["parallelism"].startswith("frame-shard x2")
    assert d["roofline"]["avg_launch_ms"] > 0 and d["host_end"]["counters"]["l1_pre_crc_errors"] == 0
    assert d["config_5"]["n_gpus"] == 2 and d["config_5"]["value"] > 0              # BASELINE configs[4] (r = 2/3) rides on the same line


WORKER = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
import ref_cases as rc, t2_tx
from sdr_receiver_dvb_t2_amd.receiver import t2_rx
from sdr_receiver_dvb_t2_amd.shard import ordered_receiver, frame_alignment
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
# 16K / 64-QAM / 16200 r = 1/2, 16 FEC blocks per T2 frame: alignment 2, eight frames -> four per rank
mode, lps, plp, nb, frames = (4, 1, 6, 4, 0, 8), 200, (2, 0, 0), 16, 8
rc.CARRY_CASES["shard"] = dict(mode=mode, lps=lps, plp=plp, nb=nb, frames=frames, snr=16.0, seed=31, s2=8, iq_snr=20.0)
i16, q16 = rc.carry_iq("shard")
k_bch = t2_tx.K_BCH[0]
d_i, d_q = torch.from_numpy(i16).cuda(), torch.from_numpy(q16).cuda()
rx = t2_rx(*mode, lps, *plp, 1, nb, max_frames=frames)
assert frame_alignment(nb, 32) == 2
def decode(lo, hi):
    rx.reset()
    n = rx.execute_dev(d_i[lo:hi].reshape(-1), d_q[lo:hi].reshape(-1), hi - lo, first_call=True)
    assert n == (hi - lo) * nb and rx.carry == 0                     # batch-aligned share: nothing left waiting
    return rx.fetch_packed(n)
orx = ordered_receiver(decode, nb, 32, 0, dist, packed_k_bch=k_bch)
got = orx.execute(frames)
if rank == 0:
    rx.reset()
    n = rx.execute_dev(d_i.reshape(-1), d_q.reshape(-1), frames, first_call=True)
    rows, trials = rx.fetch_packed(n)
    assert (trials >= 0).all()
    one = ordered_receiver(lambda lo, hi: (rows, trials), nb, 32, 0, None, packed_k_bch=k_bch)
    want = one.execute(frames)
    assert got is not None and got.size > 50000 and np.array_equal(got, want), (got.size, want.size)
    one.close()
else:
    assert got is None
orx.close(); rx.close()
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_ordered_receiver_two_real_ranks_on_one_device(built, tmp_path):
    """Two t2_rx ranks (both on device 0) each decode a batch-aligned half of eight T2 frames; rank 0 gathers the packed BBFRAMEs in frame
    order into one de-framer: the TS equals that of one sequential call over all eight frames."""
    script = tmp_path / "w.py"
    script.write_text(WORKER % (ROOT, ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
        assert "ok" in out
