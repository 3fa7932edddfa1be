"""GPU tier: THE REFERENCE ITSELF running on libt2gpu.so. oracle/_ref/libref_t2rx_gpufec.so and libref_t2rx_gpu.so are the reference's own
compiled objects (src/DVB_T2/*.cpp, Qt signal flow, stage threads, buffers -- oracle/Makefile) in which the bodies of the slots named in
INTEGRATION.md are the ones of integration/*_gpu.cpp: calls into the C ABI of include/t2gpu.h. What the reference then writes to its TS
file must be what the unmodified reference wrote on the same samples (fixture rxoff/rx16k, tests/golden/t2rx_golden.npz), packet for
packet. This is the drop-in boundary exercised from the reference's side."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
from test_ref_pins import _load, sub

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _have(lib):
    return os.path.exists(os.path.join(ol.ROOT, "oracle", "_ref", "lib%s.so" % lib)) and os.path.exists("/opt/conda/lib/libQt5Core.so.5")


def run_binding(lib, case, tmp_path, env_extra=None):
    out = str(tmp_path / (lib + ".npz"))
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(HERE, "binding_case.py"), "--lib", lib, "--case", case, "--out", out], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    with np.load(out) as z:
        return {k: z[k] for k in z.files}, p.stderr


@pytest.mark.parametrize("lib,case", [("ref_t2rx_gpufec", "rx16k"), ("ref_t2rx_gpu", "rx16k"), ("ref_t2rx_gpu", "rx32k")])
def test_the_reference_runs_on_the_library(built, tmp_path, lib, case):
    if not _have(lib):
        pytest.skip("oracle/_ref/lib%s.so (built where /root/reference and the Qt SDK are) is not here" % lib)
    g = sub(_load("t2rx_golden.npz"), "rxoff", case)
    got, err = run_binding(lib, case, tmp_path)
    want = g["ts_packet_crc"]
    mine = got["ts_packet_crc"]
    print("%s on %s: %d BBFRAMEs (reference %d), %d TS packets (reference %d), re-tune requests %s" %
          (lib, case, int(got["bbframes"]), int(g["bbframes"]), len(mine), len(want), got["asked"].tolist()))
    assert [int(k) for k, _ in got["asked"]] == [int(k) for k, _, _ in g["moves"]]
    assert np.abs(got["asked"][:, 1] - g["moves"][:, 1]).max() < 1.0
    assert int(got["bbframes"]) == int(g["bbframes"])
    assert len(mine) == len(want) and np.array_equal(mine, want), (len(mine), len(want), int((mine[:min(len(mine), len(want))] != want[:min(len(mine), len(want))]).sum()))
    assert int(got["ts_len"]) == int(g["ts_len"])
