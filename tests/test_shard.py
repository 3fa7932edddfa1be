"""CPU tier: frame sharding and the N>1 timing aggregation (world_size 2, gloo)."""
import os
import subprocess
import sys

import pytest

from sdr_receiver_dvb_t2_amd.shard import shard_frames

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("total,world,align", [(4096, 8, 32), (100, 3, 32), (0, 2, 32), (31, 4, 32), (65, 2, 1)])
def test_shards_partition_the_frames(total, world, align):
    spans = [shard_frames(total, world, r, align) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert a <= b == c <= d
    for lo, hi in spans:
        assert lo % align == 0 or lo == total


WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from sdr_receiver_dvb_t2_amd.shard import shard_frames, aggregate_timing
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lo, hi = shard_frames(1000, world, rank, 32)
secs, units = aggregate_timing(1.0 + rank, hi - lo, dist, torch.device("cpu"))
assert secs == float(world) and units == 1000.0, (secs, units)
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_two_rank_aggregation_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out
        assert "ok" in out
