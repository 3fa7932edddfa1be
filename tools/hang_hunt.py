#!/usr/bin/env python3
"""Run the reset-in-the-middle case of tests/test_host_mirror_gpu.py (stage_mirror_test rx, three forms of the loop) N times with a short
timeout; on a timeout dump every thread's stack with rocgdb. Run on the GPU box: python tools/hang_hunt.py [N]."""
import os
import pathlib
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_host_mirror_gpu as t

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tmp = pathlib.Path(tempfile.mkdtemp())
exe = str(tmp / "stage_mirror_test")
pkg = os.path.join(ROOT, "sdr_receiver_dvb_t2_amd")
subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "stage_mirror_test.cpp"),
                       "-L" + pkg, "-lt2gpu", "-Wl,-rpath," + pkg, "-o", exe])
m, buf, marks = t._unconfigured_stream(tmp, 16, 291, 0.0, spoil_frame=9)
env0 = dict(os.environ)
env0["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env0.get("LD_LIBRARY_PATH", "")
env0["STAGE_EXIT_TRACE"] = "1"
env0["T2GPU_EXIT_TRACE"] = "1"
variants = [{"STAGE_DEVICE_LOOP": "0"}, {"STAGE_DEVICE_LOOP": "1"}, {"STAGE_DEVICE_LOOP": "1", "STAGE_PIN": "1"}]
if os.environ.get("HUNT_ONLY_PIN"):
    variants = variants[2:]
hangs = 0
for k in range(n):
    for v in variants:
        env = dict(env0); env.update(v)
        p = subprocess.Popen([exe, "rx", str(tmp / "i.s16"), str(tmp / "q.s16"), str(tmp / "o.ts"), str(buf), "0", str(tmp / "log.txt")],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        try:
            p.communicate(timeout=40)
            if p.returncode != 0:
                print("run", k, v, "rc", p.returncode, flush=True)
        except subprocess.TimeoutExpired:
            hangs += 1
            print("HANG in run", k, v, flush=True)
            g = subprocess.run(["timeout", "60", "/opt/rocm/bin/rocgdb", "-p", str(p.pid), "-batch", "-ex", "thread apply all bt 14"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
            print(g.stdout[-12000:], flush=True)
            p.kill()
            out, err = p.communicate()
            print("stderr of the hung process:\n" + "\n".join(err.decode(errors="replace").splitlines()[-14:]), flush=True)
            print("log tail:\n" + "\n".join(open(tmp / "log.txt").read().splitlines()[-6:]), flush=True)
            if hangs >= 4:
                sys.exit(1)
print("runs", n, "x 3, hangs", hangs)
