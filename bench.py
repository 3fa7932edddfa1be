#!/usr/bin/env python3
"""bench.py -- throughput of the DVB-T2 demod -> TS hot path on MI355X (contract: task brief / DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}] [--frames F] [--trials T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload (BASELINE.json config 3, "CFG-A"; --config 5 = the same with r = 2/3, --config 4 = 16K / 64-QAM / 16200 r = 1/2,
--config 2 = FFT + equalise + de-interleave + demap only): F complete T2 frames per GPU per step -- 8 MHz, 32K extended, GI 1/128, PP7,
1 P2 + 59 data symbols, real L1-pre / L1-post signalling in every P2 symbol, one PLP, rotated 256-QAM, LDPC 64800 r=3/4, 202 FEC
blocks per frame -- synthetic, built by the transmitter model in tests/t2_tx.py (P1 + cyclic prefixes + AWGN), resident in HBM as the
int16 I/Q samples a tuner delivers at 64/7 Msps (the dvbt2_demodulator::execute boundary) before the clock starts. One step = front
end (dc / IQ imbalance / NCO / Farrow x2 / 64-tap decimator) -> P1 detection at every frame start -> guard-interval correlation of
every symbol -> FFT with the guard dropped -> P2/data equaliser + frequency de-interleave -> time/cell de-interleave -> demap -> LDPC
(reference SIMD-batch rule: batches of 32 formed across frames, 25 trials) -> BB descramble + bit packing -> [host worker of the library,
overlapped with the next step: L1-pre / L1-post parse + CRC-32 of every frame, batch drop rule, BBFRAME de-framing -> TS] for all F
frames, tracking loops open (zeros and the nominal resample). F is a multiple of the frame alignment (16 for CFG-A: 16 x 202 FEC blocks
= 101 batches of 32), so no step ends inside a SIMD batch. The clock stops when the TS bytes of the last step are on the host.
Rank 0 prints ONE JSON line: `value` = input IQ samples per second; `roofline` is for the dominant kernel (LDPC) from HIP events around
its launches inside the timed region, `roofline.kernels` has every stage (algorithmic HBM bytes of SURVEY.md 8d / measured stage
time); `cpu_baseline` times the same chain on ALL host cores (one process per core: oracle restatement + the reference's own LDPC where
oracle/_ref is loadable) on one frame each, and names the CPU.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLOCK_HZ = 2.4e9                            # MI355X_MICROARCH.md; GRBM_GUI_ACTIVE of the LDPC launch / its duration = 2.39 GHz
# Hardware counters of the dominant kernel come from the committed PMC passes of THIS workload (tools/pmc_passes.sh -> tools/pmc_summary.py ->
# profiles/ldpc_counters.json: per-launch averages of ldpc_decode2_kernel, separate --pmc runs as MI355X_MICROARCH.md prescribes) and are scaled
# by frames x sweeps; the launch duration they are divided by is measured live in this run.
COUNTERS_FILE = os.path.join(ROOT, "profiles", "ldpc_counters.json")
LDPC_LINKS = {(1, 3): 226799, (1, 2): 215999, (0, 0): 48599}
LDPC_KERNEL_SOURCES = ("ldpc_kernel2.hip", "ldpc_cn3.h", "ldpc_kernel.h")     # what the PMC passes profiled: the counters file names their hash


def ldpc_kernel_hash():
    import hashlib
    h = hashlib.sha256()
    for fn in LDPC_KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "sdr_receiver_dvb_t2_amd", "csrc", fn), "rb").read())
    return h.hexdigest()[:16]       # edges per frame of the benchmarked codes (t2gpu_ldpc_graph_stats)

# BASELINE.json configs that run on one GPU. mode = (fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data),
# plp = (modulation, fec_type, code_rate, rotation); frames = T2 frames per GPU per step: a multiple of the frame alignment (no step
# ends inside a SIMD batch) that fills whole rounds of the decoder's 16 resident batch slots: 48 x 202 = 9696 FEC frames = 303 batches
# = 18.9 rounds (38 frames in round 2 were 239 batches + a 28-frame tail the reference would never form)
CONFIGS = {
    2: dict(name="config 2 (CFG-A, FFT + equalise + de-interleave + demap only)", mode=(5, 1, 6, 4, 0, 59), lps=350, plp=(3, 1, 3, 1), frames=48, s2=10, snr=21.0,
            metric="IQ Msamples/s through FFT + equaliser + demap (32K, 256-QAM)"),
    3: dict(name="config 3 (CFG-A)", mode=(5, 1, 6, 4, 0, 59), lps=350, plp=(3, 1, 3, 1), frames=48, s2=10, snr=21.0,
            metric="IQ Msamples/s demod->TS (32K, 256-QAM, LDPC 64800 r=3/4)"),
    4: dict(name="config 4 (CFG-B)", mode=(4, 1, 6, 4, 0, 40), lps=200, plp=(2, 0, 0, 1), frames=64, s2=8, snr=12.0,
            metric="IQ Msamples/s demod->TS (16K, 64-QAM, LDPC 16200 r=1/2)"),
    5: dict(name="config 5 (CFG-C)", mode=(5, 1, 6, 4, 0, 59), lps=350, plp=(3, 1, 2, 1), frames=48, s2=10, snr=21.0,
            metric="IQ Msamples/s demod->TS (32K, 256-QAM, LDPC 64800 r=2/3)"),
}
# not a BASELINE config: the 32K mode of the headline with a constellation the reference's own arithmetic DECODES (its wrapping int8 cast
# loses every 256-QAM batch, llr_demapper.cpp:722-737) -- demod -> TS at 32K timed with a real transport stream and without the clamping
# extension. 151 FEC blocks per frame: the frame alignment is 32 T2 frames = 151 SIMD batches.
CONFIGS[6] = dict(name="32K / 64-QAM / 64800 r=2/3 (CFG-A's OFDM mode; decodes in reference arithmetic)", mode=(5, 1, 6, 4, 0, 59), lps=350,
                  plp=(2, 1, 2, 1), frames=32, s2=10, snr=20.0, metric="IQ Msamples/s demod->TS (32K, 64-QAM, LDPC 64800 r=2/3)")
K_LDPC = {0: (7200, 9720, 10800, 11880, 12600, 13320), 1: (32400, 38880, 43200, 48600, 51840, 54000)}
K_BCH = {0: (7032, 9552, 10632, 11712, 12432, 13152), 1: (32208, 38688, 43040, 48408, 51648, 53840)}


class Workload(object):
    """Geometry of a config from the PRODUCT's own mode tables (t2gpu_ofdm_mode_info; host only, no GPU needed)."""

    def __init__(self, cfg):
        import ctypes
        import sdr_receiver_dvb_t2_amd as pkg
        self.cfg, self.mode, self.lps, self.plp = cfg, cfg["mode"], cfg["lps"], cfg["plp"]
        info = (ctypes.c_int * 12)()
        assert pkg.lib().t2gpu_ofdm_mode_info(*self.mode, info) == 0
        (self.fft_size, self.k_total, _, _, _, self.c_p2, self.c_data, self.n_fc, _, self.l_fc, self.len_frame, self.guard) = (int(v) for v in info)
        self.n_data = self.mode[5]
        self.sym = self.fft_size + self.guard
        self.frame_samples = 2048 + self.len_frame * self.sym
        self.fec_size = 64800 if self.plp[1] else 16200
        self.bpc = 2 * (self.plp[0] + 1)
        self.cpf = self.fec_size // self.bpc
        self.cid = 6 * self.plp[1] + self.plp[2]
        self.frame_cells = (self.c_p2 - 1840 - self.lps) + (self.n_data - self.l_fc) * self.c_data + self.l_fc * self.n_fc
        self.nb = self.frame_cells // self.cpf                                    # FEC blocks of the one PLP that fills the frame
        self.k_ldpc, self.k_bch = K_LDPC[self.plp[1]][self.plp[2]], K_BCH[self.plp[1]][self.plp[2]]

    def stage_bytes(self, F):
        """ALGORITHMIC HBM bytes of every stage for F frames (each datum once in, once out; tables excluded)."""
        syms = F * self.len_frame
        cells = F * self.nb * self.cpf
        return {
            "front": 12 * F * self.frame_samples,                                    # int16 I/Q in, complex64 out
            "guard_corr": syms * 2 * self.guard * 8,
            "fft": syms * 16 * self.fft_size,
            "equalise": F * 8 * (self.k_total * self.len_frame + self.c_p2 + (self.n_data - self.l_fc) * self.c_data + self.l_fc * self.n_fc),
            "ti": 16 * cells,
            "demap": cells * (8 + self.bpc),                                         # cells in once, one LLR byte per bit out (8d: 16 B per 256-QAM cell)
            "ldpc": F * self.nb * (self.fec_size + self.k_ldpc),
            "descramble": F * self.nb * (self.k_ldpc + self.k_bch // 8),             # bit-bytes in, packed bytes out (8d)
        }


def make_frames(w, n_unique, snr_db, seed):
    """n_unique synthetic frames of workload w as int16 I/Q at the tuner interface: (I [n][frame_samples], Q, sent TS packets)."""
    import oracle_lib as ol
    import t2_tx
    m = ol.ora_mode(*w.mode)
    assert t2_tx.plp_blocks_per_frame(m, w.lps, w.cpf) == w.nb
    per = w.nb * (w.k_bch // 1496 + 1)
    frames, sent = [], []
    for f in range(n_unique):
        ts = t2_tx.ts_packets(per, seed + f)
        stream, _, _ = t2_tx.build_plp_frame_cells(w.cid, w.plp[0], w.plp[1], w.plp[2], ts, w.nb)
        l1 = t2_tx.l1_cells(w.mode, w.lps, w.plp[0], w.plp[1], w.plp[2], w.nb, frame_idx=f)
        frames.append(t2_tx.build_frame(m, stream, w.lps, seed + 100 + f, snr_db=None, phase=0.0, l1_cells=l1))
        sent.append(ts)
    i16, q16, flen = t2_tx.iq_stream(frames, w.guard, w.cfg["s2"], snr_db, seed)
    assert flen == w.frame_samples
    return i16.reshape(n_unique, flen), q16.reshape(n_unique, flen), sent


CPU_STAGES = ("front", "farrow", "decim", "p1", "guard_corr", "fft", "eq", "ti", "demap", "ldpc", "descramble")


def _cpu_worker(args):
    """One frame (int16 I/Q) through the CPU chain on one core, over and over for `seconds`, with a timer around every stage. What the
    reference computes once at init (carrier maps, pilot references, de-interleaver tables: pilot_generator, address_freq_deinterleaver)
    is computed once here too, outside the clock. Stages: oracle front loop (dc / IQ / NCO), Farrow and 64-tap decimator (the
    reference's own classes from oracle/_ref/libref_dsp.so where that loads, else the C restatement), P1 detector, guard correlation,
    FFT (oracle/fft_oracle.c), equaliser + frequency de-interleaver, time / cell de-interleaver, demapper (C restatement, -O3 -mavx2),
    LDPC (the reference's own headers compiled, oracle/_ref/libref_ldpc.so, where that loads), BB descrambler.
    Returns (passes, elapsed, kinds, per-stage seconds)."""
    cfg_id, i16, q16, seconds, full = args
    import ctypes
    import numpy as np
    import oracle_lib as ol
    w = Workload(CONFIGS[cfg_id])
    m = ol.ora_mode(*w.mode)
    L = ol.oracle()
    ldpc, lkind = (ol.ref_decode, "reference LDPC headers compiled") if ol.ref() is not None else (ol.ora_decode, "C restatement")
    ref_dsp = ol.ref_dsp() is not None
    dkind = "reference filter_decimator / interpolator_farrow compiled" if ref_dsp else "C restatement"
    # init-time tables (the reference: pilot_generator + address_freq_deinterleaver in dvbt2_demodulator::init)
    carriers = [ol.ora_symbol_carriers(m, l) for l in range(m.len_frame)]
    fdi = {k: ol.ora_freq_deint(m, k) for k in (0, 1, 2)}
    kind_of = [0 if l < m.n_p2 else (2 if (m.l_fc and l == m.len_frame - 1) else 1) for l in range(m.len_frame)]
    ncell = [m.c_p2, m.c_data, m.n_fc]
    eq = L.ora_symbol_equalise
    eq.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6
    fft = ol.OraFft(m.fft_size)
    ti = ol.OraTi(w.cpf, w.nb)
    x = np.concatenate((i16, i16[:4096])), np.concatenate((q16, q16[:4096]))         # a little of the next frame: filter delay
    sync = np.zeros(2, np.float32)
    # the stage buffers exist before the clock starts, as the reference's do (allocated in its constructors)
    b_derot, b_up, b_dec = (np.zeros(len(x[0]) * k + 16, np.complex64) for k in (1, 2, 1))
    cells, tib = np.zeros(sum(ncell[k] for k in kind_of), np.complex64), np.zeros(w.nb * w.cpf, np.complex64)
    acc = dict.fromkeys(CPU_STAGES, 0.0)
    clock = time.perf_counter
    t0 = clock()
    reps = 0
    while True:
        fo, fa, de, p1 = ol.OraFront(0), ol.OraFarrow(ref=ref_dsp), ol.OraDecim(ref=ref_dsp), ol.OraP1()
        t = clock(); derot, theta = fo.execute(x[0], x[1], [len(x[0])], [0.0], [0.0], out=b_derot); acc["front"] += clock() - t
        t = clock(); up = fa(derot, 0.5, out=b_up); acc["farrow"] += clock() - t
        t = clock(); stream = de(up, out=b_dec); acc["decim"] += clock() - t
        t = clock()
        level = float(np.mean(np.abs(stream.real)) * np.mean(np.abs(stream.imag)))
        r = p1.execute(stream[:3072], 0, True, level)
        acc["p1"] += clock() - t
        first = r["consume"] - r["idx_buffer_sym"] if r["detected"] else 2048 + 17
        at = 0
        for l in range(m.len_frame):
            s0 = first + l * w.sym
            t = clock(); ol.ora_cp_frequency_est(stream[s0:s0 + w.sym], m.fft_size, w.guard); acc["guard_corr"] += clock() - t
            t = clock(); spec = fft(stream[s0 + w.guard:s0 + w.sym]); acc["fft"] += clock() - t
            t = clock()
            k = kind_of[l]
            h = fdi[k][1] if l % 2 == 0 else fdi[k][0]
            out = cells[at:at + ncell[k]]
            n = eq(ctypes.addressof(m), k, spec.ctypes.data, carriers[l][0].ctypes.data, carriers[l][1].ctypes.data, h.ctypes.data,
                   out.ctypes.data, sync.ctypes.data)
            assert n == ncell[k]
            acc["eq"] += clock() - t
            at += n
        plp = np.concatenate((cells[1840 + w.lps:ncell[0]], cells[ncell[0]:]))[:w.nb * w.cpf]   # the PLP's cells: behind L1-pre / L1-post in P2
        t = clock()
        ti.begin(w.nb)
        ti.push(plp, tib)
        acc["ti"] += clock() - t
        t = clock(); llr, _, _ = ol.ora_demap(w.plp[0], w.plp[1], w.plp[2], w.plp[3], tib); acc["demap"] += clock() - t
        if full:
            for b0 in range(0, (w.nb // 32) * 32, 32):
                t = clock(); tr, bits, _ = ldpc(w.cid, llr[b0:b0 + 32]); acc["ldpc"] += clock() - t
                if tr >= 0:
                    t = clock(); ol.ora_bch_descramble(w.cid, bits); acc["descramble"] += clock() - t
        reps += 1
        el = clock() - t0
        if el >= seconds:
            break
    return reps, el, (lkind, dkind), acc


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_chain_baseline(cfg_id, w, i16, q16, full):
    """The chain of _cpu_worker (i) on ONE core with the machine otherwise idle, ~8 s -- the reference's effective execution (its stage
    threads are serialised by their handshake) -- and (ii) on every host core at once (one process per core, each
    on the same frame), ~12 s. `value` is (ii); the per-stage split is (i)'s."""
    import multiprocessing as mp
    cores = max(1, min(len(os.sched_getaffinity(0)), 128))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(1) as pool:
        one = pool.map(_cpu_worker, [(cfg_id, i16, q16, 8.0, full)])[0]
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(cfg_id, i16, q16, 12.0, full)] * cores)
    wall = time.perf_counter() - t0
    rate = sum(r[0] * w.frame_samples / r[1] for r in res) / 1e6
    tot = sum(one[3].values())
    stages = {k: {"ms_per_frame": round(one[3][k] / one[0] * 1e3, 2), "share": round(one[3][k] / tot, 3)} for k in CPU_STAGES if one[3][k] > 0}
    batches = (w.nb // 32) if full else 0
    return {"value": round(rate, 3), "unit": "Msamples/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "single_core": {"value": round(one[0] * w.frame_samples / one[1] / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                            "ms_per_frame": round(one[1] / one[0] * 1e3, 1), "stages": stages,
                            "ldpc_codewords_per_s": round(32 * batches * one[0] / one[3]["ldpc"], 1) if full and one[3]["ldpc"] > 0 else None,
                            "sample": "one process, ~8 s, machine otherwise idle"},
            "best_core_of_all": round(max(r[0] * w.frame_samples / r[1] for r in res) / 1e6, 3),
            "build": "oracle/liboracle.so: gcc -O3 -mavx2 -ffp-contract=off (source-order float arithmetic); LDPC = %s; Farrow + decimator = %s"
                     % one[2],
            "sample": "%d processes x ~12 s after one process x ~8 s (%.1f s wall in all), each passing one %s frame from int16 I/Q over and over "
                      "(front end, P1, %d symbols%s); init-time tables outside the clock as in the reference"
                      % (cores, wall, w.cfg["name"], w.len_frame,
                         ", %d of %d FEC blocks in whole SIMD batches of 32 through the LDPC, 25 trials each" % ((w.nb // 32) * 32, w.nb) if full
                         else ", up to the LLRs")}


def cpu_reference_baseline(w, ui, uq, seconds=15.0):
    """kind "reference": the reference's own receiver (oracle/_ref/libref_t2rx.so: src/DVB_T2/*.cpp compiled with the reference's flags, stage
    objects on their own QThreads, FFTW as shipped) from int16 I/Q to the TS on this box's host cores, on the bench's own frames, in a process
    of its own (tests/ref_rx_timing.py). None when the build does not load here."""
    import subprocess
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        np.concatenate([ui] * 8).reshape(-1).tofile(os.path.join(d, "i.s16"))          # 16 frames: the ring closes on a frame boundary
        np.concatenate([uq] * 8).reshape(-1).tofile(os.path.join(d, "q.s16"))
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_rx_timing.py"), "--i", os.path.join(d, "i.s16"), "--q", os.path.join(d, "q.s16"),
                                "--frame-samples", str(w.frame_samples), "--seconds", str(seconds)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            return None
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return None
    r = json.loads(lines[-1])
    if "error" in r:
        return None
    return {"value": round(r["msamples_per_s"], 3), "unit": "Msamples/s", "cores": r["threads"], "kind": "reference", "cpu": cpu_model(),
            "build": "oracle/_ref/libref_t2rx.so: /root/reference/src/DVB_T2/*.cpp + LDPC/*.hh compiled unmodified with the reference's flags (g++ -Ofast "
                     "-mavx2, Qt 5.9.7, the FFTW binary the reference ships); dvbt2_demodulator -> time_deinterleaver -> llr_demapper -> ldpc_decoder -> "
                     "bch_decoder -> bb_de_header each on its own QThread as the reference runs them",
            "sample": "%.1f s = %.1f %s frames from int16 I/Q through dvbt2_demodulator::execute in 172 032-sample buffers (rx_sdrplay.h:64) to the TS file, "
                      "clock started once the receiver had acquired (after %.1f frames); every 256-QAM SIMD batch runs its 25 trials and is dropped, as on the GPU"
                      % (r["seconds"], r["frames"], w.cfg["name"], r["acquired_after_frames"])}


DROP_IN_EXE = os.path.join(ROOT, "sdr_receiver_dvb_t2_amd", "bin", "t2gpu_rx_file")
DROP_IN_BUF = 172032          # samples per execute() call: norm_blocks x 384 of the reference's SDRplay thread (rx_sdrplay.h:64, rx_sdrplay.cpp:199-261)


def drop_in_leg(w, ui, uq, device, frames=132, warm_frames=12, sent=None, saturate=False, snr_db=None):
    """The slot-shaped path: int16 I/Q in device-buffer-sized calls through t2::dvbt2_demodulator::execute (t2gpu_demod_execute: closed
    tracking loops, the reference's own acquisition from P1 / guard search / L1-pre / L1-post) and the stage classes of
    include/t2gpu_stages.hpp wired as the reference wires its objects (time_deinterleaver -> llr_demapper -> ldpc_decoder -> bch_decoder ->
    bb_de_header), in a plain C++ process (examples/t2gpu_rx_file.cpp, built by csrc/Makefile). The first warm_frames frames'
    worth of buffers (acquisition) run before the program's clock starts; 60 frames (100 of config 4) are timed, more than half a second,
    so that two rounds' figures can be compared (round 4 timed 15 frames, 0.18 s). sent: the TS packets the frames carry -- every packet that
    comes out is then looked up among them. saturate: the clamped-LLR extension (the reference's wrapping cast loses every 256-QAM
    SIMD batch, here as in the batch legs)."""
    import subprocess
    import tempfile
    import numpy as np
    if not os.path.exists(DROP_IN_EXE):
        return {"error": "sdr_receiver_dvb_t2_amd/bin/t2gpu_rx_file not built"}
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        reps = (frames + ui.shape[0] - 1) // ui.shape[0]
        np.concatenate([ui] * reps)[:frames].reshape(-1).tofile(os.path.join(d, "i.s16"))
        np.concatenate([uq] * reps)[:frames].reshape(-1).tofile(os.path.join(d, "q.s16"))
        warm = (warm_frames * w.frame_samples + DROP_IN_BUF - 1) // DROP_IN_BUF
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
        t0 = time.perf_counter()
        wrap = os.environ.get("T2GPU_DROPIN_WRAPPER", "").split()          # e.g. "rocprofv3 --kernel-trace --stats -d gpurun_out/x --"
        ts_path = os.path.join(d, "out.ts")
        p = subprocess.run(wrap + [DROP_IN_EXE, os.path.join(d, "i.s16"), os.path.join(d, "q.s16"), "--out", ts_path, "--buf", str(DROP_IN_BUF),
                            "--warm", str(warm), "--json", "1", "--device", str(device), "--saturate", "1" if saturate else "0"] + os.environ.get("T2GPU_DROPIN_ARGS", "").split(),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
        wall = time.perf_counter() - t0
        ts = np.fromfile(ts_path, np.uint8) if p.returncode == 0 and os.path.exists(ts_path) else np.zeros(0, np.uint8)
    if p.returncode != 0:
        return {"error": p.stderr[-400:]}
    if os.environ.get("T2GPU_DEMOD_PROF") or os.environ.get("T2GPU_RX_PROF"):
        sys.stderr.write("".join(l + "\n" for l in p.stderr.splitlines() if l.startswith("  ") or l.startswith("t2gpu_demod profile")))
    r = json.loads(p.stdout.strip().splitlines()[-1])
    dropped = p.stderr.count("LDPC decoder could not recover the codeword!")
    out = {"value": round(r["msamples_per_s"], 1), "unit": "Msamples/s", "real_time_factor": round(r["msamples_per_s"] / (64.0 / 7.0), 1),
           "samples_per_call": DROP_IN_BUF, "calls_timed": r["buffers"], "t2_frames_timed": r["t2_frames"], "seconds": round(r["seconds"], 4),
           "bbframes": r["bbframes"], "ts_bytes": r["ts_bytes"], "simd_batches_dropped_by_ldpc": dropped, "resets": r["resets"],
           "acquired": bool(r["deint_start"]), "process_wall_s": round(wall, 2), "llr_cast": "clamped (extension)" if saturate else "reference (wraps)",
           "device_loop": "--device-loop 0" not in os.environ.get("T2GPU_DROPIN_ARGS", ""),
           "ldpc_batches_of_a_burst_in_one_launch": "--ldpc-merge 0" not in os.environ.get("T2GPU_DROPIN_ARGS", "")}
    if sent is not None:
        # whole packets of the program's output (from the start of the file: the first BBFRAME's SYNCD puts it on a packet boundary) among those sent
        pk = ts[:ts.size // 188 * 188].reshape(-1, 188)
        pk = pk[pk[:, 0] == 0x47] if pk.size else pk
        want = np.unique(packet_hashes(np.concatenate(sent)))
        hit = int(np.isin(packet_hashes(pk), want).sum()) if pk.shape[0] else 0
        out["ts_packets"] = int(pk.shape[0])
        out["ts_packets_that_were_sent"] = hit
        if r["bbframes"] == 0 and dropped > 0:
            out["ts_matches_sent"] = None
            out["note"] = "no transport stream: every SIMD batch was dropped by the LDPC stage (the reference's wrapping int8 cast on 256-QAM, as in the headline leg)"
        else:
            out["ts_matches_sent"] = bool(pk.shape[0] > 0 and hit >= pk.shape[0] - 2 * frames)     # a cut packet at a dropped batch / the stream's first
    out["entry"] = ("t2::dvbt2_demodulator::execute(len, i, q, signal) per buffer = t2gpu_demod_execute, host buffers between all stage "
                    "classes as the reference's slots carry them; loops closed, nothing configured (mode from P1 / L1)")
    out["workload"] = "%s, %d frames of int16 I/Q at %.1f dB, the first %d frames' buffers untimed (acquisition)" % (
        w.cfg["name"], frames, w.cfg["snr"] if snr_db is None else snr_db, warm_frames)
    return out


def packet_hashes(packets):
    """64-bit mixing hash of every 188-byte row (to count transport-stream packets that are among those sent)."""
    import numpy as np
    wts = (np.arange(1, 189, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    out = np.empty(packets.shape[0], np.uint64)
    for a in range(0, packets.shape[0], 65536):
        out[a:a + 65536] = (packets[a:a + 65536].astype(np.uint64) * wts[None, :]).sum(axis=1, dtype=np.uint64)
    return out


def ldpc_counters(cfg_id, frames, sweeps, launch_s, links_per_frame, occ):
    """north_star's LDPC figures: PMC counters of the committed passes (per launch of the profiled workload) scaled to this run's
    frames x sweeps, over this run's measured launch time."""
    try:
        c = json.load(open(COUNTERS_FILE))
    except (OSError, ValueError):
        return None
    if c.get("config") != cfg_id:
        return None
    # counters of ANOTHER build of the kernel are not this run's: the file names the kernel and the hash of its sources
    # (tools/pmc_summary.py writes both); a file without them, or with others, is refused and the line says so
    if c.get("kernel") != "ldpc_decode2_kernel" or c.get("kernel_sources_sha16") != ldpc_kernel_hash():
        return {"counters_stale": "profiles/ldpc_counters.json was taken from another build of the kernel (%s, sources %s; this build: %s): re-run tools/pmc_passes.sh"
                                  % (c.get("kernel"), c.get("kernel_sources_sha16"), ldpc_kernel_hash())}
    scale = frames * sweeps / float(c["frames"] * c["sweeps"])
    valu = c["SQ_INSTS_VALU"] * scale                                   # wave instructions
    cus, waves_cu = occ["cus"], occ["workgroups_per_cu"] * occ["waves_per_workgroup"]
    cu_cycles = launch_s * CLOCK_HZ * cus
    out = {
        "traffic": round((2 * c["FETCH_SIZE_KiB"] + c["WRITE_SIZE_KiB"]) * 1024 * scale),
        "occupancy_waves_per_cu": waves_cu, "occupancy_max_waves_per_cu": 32,
        "workgroups_per_cu": occ["workgroups_per_cu"], "lds_bytes_per_workgroup": occ["lds_bytes_per_workgroup"],
        "lds_busy_frac": round(c["SQ_LDS_IDX_ACTIVE"] * scale / cu_cycles, 3),
        "lds_bank_conflict_frac_of_lds_busy": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3),
        "wait_any_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
        "valu_lane_insts_per_edge_update": round(valu * 64 / (links_per_frame * frames * sweeps), 2),
        "valu_wave_insts_per_simd_cycle": round(valu / (cu_cycles * 4), 3),
        "lds_algorithmic_TBs": round(4.0 * links_per_frame * frames * sweeps / launch_s / 1e12, 2),
        "counters_from": "profiles/ldpc_counters.json (%s), scaled by frames x sweeps" % c.get("source", "?"),
    }
    vr = c.get("valu_cycles_per_wave_inst")                              # tools/ubench/valu_rate: measured issue cost of the op classes
    if vr:
        out["valu_cycles_per_wave_inst_measured"] = vr
        mix = c.get("valu_mix_cycles")                                   # op-mix weighted cycles per instruction of the kernel's ISA listing
        if mix:
            out["valu_issue_frac"] = round(valu * mix / (cu_cycles * 4), 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json config (3 = the metric's own)")
    ap.add_argument("--frames", type=int, default=0, help="T2 frames per GPU per step (0: the config's default)")
    ap.add_argument("--snr", type=float, default=None, help="AWGN SNR in dB of the synthetic input (default: BASELINE.md section 3 -- 21 dB, 12 dB for config 4)")
    ap.add_argument("--trials", type=int, default=25, help="LDPC trial limit (the reference's TRIALS = 25, ldpc_decoder.h)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the informative legs: 50-trial point, clamped-LLR variant, config 5 (profiling runs)")
    ap.add_argument("--no-clamped-variant", action="store_true", help="(kept for old command lines) same as --no-extra-legs")
    ap.add_argument("--only-frames-sweep", action="store_true", help="of the informative legs run the frames-per-call sweep alone (no 50-trial / clamped / drop-in / other-config legs)")
    ap.add_argument("--only-drop-in", action="store_true", help="run the drop_in leg alone (the slot-shaped path) and print its JSON")
    ap.add_argument("--saturate", action="store_true", help="with --only-drop-in: the clamped-LLR extension instead of the reference's wrapping cast")
    ap.add_argument("--no-ts-end", action="store_true", help="leave the library's host end (L1 parse + de-framing worker) off")
    args = ap.parse_args()
    if args.no_clamped_variant:
        args.no_extra_legs = True

    import numpy as np
    import torch
    import torch.distributed as dist
    import sdr_receiver_dvb_t2_amd as pkg
    from sdr_receiver_dvb_t2_amd.shard import shard_frames, aggregate_timing, frame_alignment

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # dry run of the N > 1 control flow on a one-GPU box: T2GPU_BENCH_ONE_DEVICE=1 puts every rank on device 0; the driver's runs use
    # one GPU per rank. The path has no cross-GPU dependency (north_star: "no RCCL"): what the ranks exchange is the
    # barrier around the timed region and two scalars (max seconds, sum of frames) -- over gloo, on CPU tensors, whatever the launcher;
    # nothing of RCCL is initialised.
    one_device = os.environ.get("T2GPU_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # Every library call below goes on a stream of this process's own, not the legacy NULL stream (whose every launch checks all the
    # process's blocking streams). The receiver's overlap mode does not depend on it: there a call runs on the handle's own streams.
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # one node: the loopback interface (the container's hostname may not resolve)
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from sdr_receiver_dvb_t2_amd.receiver import t2_rx

    if args.only_drop_in:
        w = Workload(CONFIGS[args.config])
        ui, uq, sent = make_frames(w, 2, args.snr if args.snr is not None else CONFIGS[args.config]["snr"], seed=20250614)
        print(json.dumps(drop_in_leg(w, ui, uq, local_rank, frames=args.frames or 132, sent=sent, saturate=args.saturate, snr_db=args.snr)))
        return

    def run_config(cfg_id, steps, warmup, extras, check_ts=False):
        """One bench line's worth of measurement for a config; returns the dict to print (rank 0) or None. check_ts: keep the TS bytes of
        the timed steps and count the packets that are among those sent."""
        cfg = CONFIGS[cfg_id]
        w = Workload(cfg)
        full = cfg_id != 2
        frames_per_gpu = args.frames or cfg["frames"]
        align = frame_alignment(w.nb, 32)
        if full and frames_per_gpu % align:
            frames_per_gpu = max(align, frames_per_gpu // align * align)              # no step ends inside a SIMD batch
        # weak scaling: every GPU demodulates frames_per_gpu whole T2 frames per step (frames are independent: no collective)
        lo, hi = shard_frames(frames_per_gpu * world, world, rank, align=1)
        F = hi - lo
        snr_db = args.snr if args.snr is not None else cfg["snr"]
        ui, uq, sent = make_frames(w, 2, snr_db, seed=20250614 + 10 * rank)
        nb, FS = w.nb, w.frame_samples
        d_i = torch.from_numpy(np.concatenate([ui] * ((F + 1) // 2))[:F].reshape(-1)).to(dev)     # int16 [F * frame_samples]
        d_q = torch.from_numpy(np.concatenate([uq] * ((F + 1) // 2))[:F].reshape(-1)).to(dev)

        # The receiver is the library's batch object (t2gpu_rx_*, csrc/t2gpu_rx.cpp): buffers, stage sequencing, launches and the host
        # end (L1 parse + de-framing worker thread) are C++ behind the C ABI; this script hands over two device pointers per step and reads
        # TS bytes and counters back. torch is the allocator of the input buffers and the process-group plumbing, nothing else.
        def make_rx(saturate, frames, trials):
            rx = t2_rx(*w.mode, w.lps, *w.plp, nb, max_frames=frames, ldpc_trials=trials, saturate_llr=saturate, device=local_rank)
            if full and not args.no_ts_end:
                rx.ts_enable(0, l1_check=True)
            return rx

        if full:
            def step(rx, level, nf):
                return rx.execute_dev(d_i, d_q, nf, level)
        else:
            def step(rx, level, nf):                   # config 2: the frames' stream and positions stay in the handle from the set-up call
                return rx.fft_eq_demap_dev(nf)

        def timed_leg(rx, steps, warmup, level, keep_ts=False, nf=None, drain=True):
            """W untimed + K timed steps bracketed by barrier + synchronize; the clock stops when the last step's TS bytes are on the host.
            Returns (seconds, per-stage ms sums, LDPC ms list, TS bytes of the timed steps, the bytes themselves if keep_ts)."""
            ts_on = full and not args.no_ts_end
            nf = nf or F
            for _ in range(max(warmup, 1)):
                step(rx, level, nf)
            torch.cuda.synchronize(dev)
            if ts_on:
                rx.ts_read(wait_all=True)                                              # warm-up output is not counted
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            acc, ldpc, ts_bytes = {}, [], 0
            if ts_on:                                  # the consumer's buffer: allocated and touched before the clock starts
                cap = (steps + 1) * nf * nb * (w.k_bch // 8 + 64) if keep_ts else (nf + 1) * nb * (w.k_bch // 8 + 64)
                sink = np.empty(cap, np.uint8)
                sink.fill(0)
            # The TS consumer is a thread of its own, as the reference's UDP / file sink is: it takes what the library's worker has
            # finished while the main thread is in the next call (ctypes releases the GIL for the call, the copy runs in C).
            done, got = threading.Event(), [0]

            def consume():
                while not done.is_set():
                    n = rx.ts_read_into(sink[got[0]:] if keep_ts else sink)
                    got[0] += n
                    if n == 0:
                        time.sleep(0.0002)
                while True:                            # the last step's frames de-framed: demod -> TS is complete
                    n = rx.ts_read_into(sink[got[0]:] if keep_ts else sink, wait_all=True)
                    got[0] += n
                    if n == 0:
                        break
            consumer = threading.Thread(target=consume) if ts_on else None
            t0 = time.perf_counter()
            if consumer:
                consumer.start()
            call_s = [] if os.environ.get("T2GPU_BENCH_CALL_TIMES") else None
            for it in range(steps):
                tc = time.perf_counter()
                step(rx, level, nf)
                if call_s is not None:
                    call_s.append(time.perf_counter() - tc)
                if not drain and it + 1 < steps:       # calls follow each other without the host waiting for the device in between
                    continue
                for k, v in rx.stage_ms().items():     # HIP events between the stages on the call's stream; also drains the step
                    if v >= 0:
                        acc[k] = acc.get(k, 0.0) + v * (1 if drain else steps)
                if full:
                    ldpc.extend([rx.last_ldpc_ms()] * (1 if drain else steps))
            if consumer:
                done.set()
                consumer.join()
                ts_bytes = got[0]
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            if call_s:                                 # T2GPU_BENCH_CALL_TIMES=1: host time inside each call of the leg
                print("  calls of %d frame(s), host ms: %s ... sum %.2f of %.2f" % (nf, " ".join("%.2f" % (x * 1e3) for x in call_s[:24]), sum(call_s) * 1e3,
                                                                               (time.perf_counter() - t0) * 1e3), file=sys.stderr)
            return time.perf_counter() - t0, acc, ldpc, ts_bytes, (sink[:ts_bytes] if ts_on and keep_ts else None)

        def ts_check(ts, tsb, secs, n_steps, counters=None):
            """TS bytes of n_steps timed steps against the packets that were sent (the bytes start wherever the warm-up's last packet
            ended: the packet phase is found from the sync bytes). SIMD batches the LDPC gave up on are dropped as the reference drops
            them (counters: the host end's own count): their packets are missing, and the packet cut at either edge of such a gap does
            not match; everything else must be a packet that was sent."""
            off = next((o for o in range(188) if ts.size > o + 188 * 64 and (ts[o:o + 188 * 64:188] == 0x47).all()), 0)
            pk = ts[off:off + (ts.size - off) // 188 * 188].reshape(-1, 188)
            good = np.isin(packet_hashes(pk), np.concatenate([packet_hashes(x) for x in sent]))
            per_frame = (nb * ((w.k_bch - 80) // 8)) // 187 - 1                     # whole packets one T2 frame's BBFRAMEs carry
            drop_frac = (counters["fec_frames_dropped_ldpc"] / max(1, counters["fec_frames"])) if counters else 0.0
            gaps = int(round(drop_frac * n_steps * F * nb / 32.0)) + 1 if drop_frac > 0 else 0   # dropped SIMD batches inside the timed steps
            want = int(n_steps * F * per_frame * (1.0 - drop_frac)) - 4 * gaps
            return {"ts_bytes": int(tsb), "ts_bytes_per_s": round(tsb / secs, 1), "ts_mbit_per_s": round(tsb * 8 / secs / 1e6, 1),
                    "ts_packets": int(pk.shape[0]), "ts_packets_that_were_sent": int(good.sum()),
                    "fec_frames_dropped_ldpc_frac": round(drop_frac, 4),
                    "ts_matches_sent": bool(good.sum() >= want and pk.shape[0] - good.sum() <= n_steps * F * 2 + 2 * gaps)}

        rx = make_rx(False, F, args.trials)            # reference semantics: truncating int8 cast in the demapper
        assert rx.frame_len == FS
        count = rx.execute_dev(d_i, d_q, F, first_call=True)                               # thresholds from the level estimate
        assert not full or (count == F * nb and rx.carry == 0)
        level = rx.results(F)["level_detect"]
        occ = rx.ldpc_occupancy()
        # One step = one call = the whole chain over one buffer of F frames (front end .. descrambler + packing), drained by the host once
        # per call because the P1 decisions are host data; the host end (copies on their own stream, worker thread) overlaps the next step.
        elapsed, stage_acc, ldpc_ms, ts_bytes, ts_kept = timed_leg(rx, steps, warmup, level, keep_ts=check_ts)
        ref_trials = rx.fetch_packed(count)[1] if full else None
        counters = rx.ts_counters() if full and not args.no_ts_end else None
        max_s, units = aggregate_timing(elapsed, F * steps, dist if world > 1 else None, None)
        rx.close()

        extra = {}
        if rank == 0 and world == 1 and full and extras:     # N = 1 only: timed_leg's barriers are collective
            if not args.only_frames_sweep:
                # (i) the BASELINE config's "50 iters" point beside the reference's own TRIALS = 25
                r50 = make_rx(False, F, 50)
                r50.execute_dev(d_i, d_q, F, first_call=True)
                e50, _, l50, _, _ = timed_leg(r50, 2, 1, level)
                r50.close()
                extra["trials_50"] = {"msamples_per_s": round(2 * F * FS / e50 / 1e6, 1), "ldpc_ms": round(sum(l50) / len(l50), 3),
                                      "note": "same workload with the LDPC trial limit at 50 (BASELINE.json config text); the reference itself stops at 25"}
                # (ii) clamped LLRs (extension) -> the same frames decode; demod -> TS with every byte checked
                c3 = make_rx(True, F, args.trials)
                c3.execute_dev(d_i, d_q, F, first_call=True)
                e3, _, _, tsb, ts = timed_leg(c3, 3, 2, level, keep_ts=True)
                t3 = c3.fetch_packed(F * nb)[1]
                k3 = c3.ts_counters() if not args.no_ts_end else None
                c3.close()
                var = {"msamples_per_s": round(3 * F * FS / e3 / 1e6, 1), "avg_ldpc_updates": round(float((args.trials - t3).mean()), 2),
                       "note": "extension (t2gpu_demap_configure saturate=1): not the reference's arithmetic"}
                if ts is not None:
                    var.update(ts_check(ts, tsb, e3, 3, k3))
                    var.update({"bytes_d2h_per_fec_frame": w.k_bch // 8, "host_end_counters": k3})
                extra["clamped_llr_variant"] = var
            # (iii) throughput against T2 frames per call of the batch receiver (the headline's 48 fill 18.9 rounds of the decoder's
            # resident batch slots; one frame = 202 FEC frames = 6.3 SIMD batches, formed across calls exactly as the reference forms them)
            # Calls are made back to back (the host waits for the device once, behind the last); `overlapped` = the same with the decode of a
            # call on the handle's own stream (t2gpu_rx_set_overlap), beside the next call's front end .. demapper.
            sweep = []
            for nf in (1, 2, 4, 8, 16, F):
                if nf > F:
                    continue
                row = {"frames_per_call": nf}
                for mode in ("plain", "overlapped"):
                    rs = make_rx(False, nf, args.trials)
                    rs.execute_dev(d_i, d_q, nf, first_call=True)
                    torch.cuda.synchronize(dev)
                    if mode == "overlapped":
                        if rs.carry:
                            rs.flush_dev()
                            torch.cuda.synchronize(dev)
                        rs.set_overlap(True)
                    # a burst of ~0.2 s per row (rounds 4-6a: at most 32 calls, where the last decode behind the last one-frame call was
                    # 8 % of the burst); the warm-up long enough for the collecting handle to have a decode resident when the clock starts
                    k = max(4, min(160, 192 // nf))
                    es, accs, ls, _, _ = timed_leg(rs, k, 2 if nf >= 8 else 6, level, nf=nf, drain=False)
                    rs.close()
                    if mode == "plain":
                        row.update({"msamples_per_s": round(k * nf * FS / es / 1e6, 1), "ms_per_call": round(es / k * 1e3, 3),
                                    "ldpc_ms_per_call": round(sum(ls) / len(ls), 3), "calls_timed": k,
                                    "stage_ms_last_call": {n: round(v / k, 4) for n, v in accs.items() if v > 0}})
                    else:
                        row.update({"overlapped_msamples_per_s": round(k * nf * FS / es / 1e6, 1), "overlapped_ms_per_call": round(es / k * 1e3, 3)})
                sweep.append(row)
            top = sweep[-1]["msamples_per_s"]
            for r in sweep:
                r["of_full_batch_rate"] = round(r["msamples_per_s"] / top, 3)
                r["overlapped_of_full_batch_rate"] = round(r["overlapped_msamples_per_s"] / top, 3)
            extra["frames_sweep"] = sweep
            if not args.only_frames_sweep:
                # (iv) the same input through the reference's own call shape (slot by slot, device-buffer-sized calls, loops closed)
                # -- with the reference's cast (all 256-QAM batches dropped by the LDPC stage, as in the headline leg), then with clamped LLRs so
                # that the transport stream comes out and is checked; config 4 (64-QAM: the cast does not wrap) through the same program
                # (each leg is a process of 2 - 3 s whose rate moves by several per cent with whatever else the box does in that moment: it is run
                # three times and the MEDIAN run reported, all three rates listed in "runs")
                def best_of_two(*a, **kw):
                    rs = [drop_in_leg(*a, **kw) for _ in range(3)]
                    bad = [r for r in rs if "error" in r]
                    if bad:
                        return bad[0]
                    rs.sort(key=lambda r: r["value"])
                    rs[1]["runs"] = [r["value"] for r in rs]
                    rs[1]["reported"] = "median of 3 runs"
                    return rs[1]
                d_in = best_of_two(w, ui, uq, local_rank, sent=sent)
                d_in["clamped_llr_variant"] = {k: v for k, v in best_of_two(w, ui, uq, local_rank, sent=sent, saturate=True).items() if k not in ("entry", "workload", "unit")}
                if cfg_id != 4:
                    w4 = Workload(CONFIGS[4])
                    ui4, uq4, sent4 = make_frames(w4, 2, CONFIGS[4]["snr"], seed=20250614)
                    d_in["config_4"] = {k: v for k, v in best_of_two(w4, ui4, uq4, local_rank, frames=240, warm_frames=20, sent=sent4).items() if k != "entry"}
                    # ... and the 32K mode with a constellation the reference's arithmetic decodes: the slot-shaped path at 32K with its TS checked
                    w6 = Workload(CONFIGS[6])
                    ui6, uq6, sent6 = make_frames(w6, 2, CONFIGS[6]["snr"], seed=20250614)
                    d_in["config_32k_64qam"] = {k: v for k, v in best_of_two(w6, ui6, uq6, local_rank, sent=sent6).items() if k != "entry"}
                extra["drop_in"] = d_in

        if rank != 0:
            return None
        msps = units * FS / max_s / 1e6
        sb = w.stage_bytes(F)
        kernels = []
        for k in ("front", "guard_corr", "fft", "equalise", "ti", "demap", "ldpc", "descramble"):
            if k in stage_acc and stage_acc[k] > 0:
                ms = stage_acc[k] / steps
                gbs = sb[k] / (ms * 1e-3) / 1e9
                kernels.append({"stage": k, "ms": round(ms, 4), "algorithmic_bytes": int(sb[k]), "achieved_GBs": round(gbs, 1),
                                "frac": round(gbs / HBM_PEAK_GBS, 4)})
        if "p1" in stage_acc:
            kernels.append({"stage": "p1 (incl. the step's one host round trip)", "ms": round(stage_acc["p1"] / steps, 4)})
        if full:
            avg_ldpc_s = (sum(ldpc_ms) / len(ldpc_ms)) / 1e3
            achieved = sb["ldpc"] / avg_ldpc_s / 1e9
            dom = {"kernel": "ldpc_decode2_kernel (two FEC frames per workgroup, packed 16-bit halves)", "avg_launch_ms": round(avg_ldpc_s * 1e3, 3),
                   "share_of_step": round(avg_ldpc_s / (max_s / steps), 3), "traffic": None,
                   "note": "HBM algorithmic bytes over the launch time, as the contract defines it; the kernel itself is bound by vector issue and "
                           "LDS latency inside one resident workgroup per CU (DESIGN.md K-ldpc): the LDS / occupancy / issue figures north_star asks "
                           "for are the keys below"}
            sweeps = args.trials if ref_trials is None or (ref_trials < 0).all() else None
            if sweeps is not None:                         # every batch ran all its sweeps: the counters scale exactly
                pmc = ldpc_counters(cfg_id, count, sweeps, avg_ldpc_s, LDPC_LINKS.get((w.plp[1], w.plp[2]), 0), occ)
                if pmc:
                    dom.update(pmc)
        else:
            top = max((k for k in kernels if "frac" in k), key=lambda k: k["ms"])
            achieved = top["achieved_GBs"]
            dom = {"kernel": top["stage"], "avg_launch_ms": top["ms"], "traffic": None}
        dropped = int((ref_trials < 0).sum()) if ref_trials is not None else 0
        out = {
            "metric": cfg["metric"],
            "value": round(msps, 1), "unit": "Msamples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(max_s / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16 in, f32 (front end, OFDM, demap) + int8 (LDPC)", "data": "synthetic",
            "config": {"workload": "%s: %d T2 frames/GPU/step = %d symbols of %dK, %d FEC frames (%d bits, code rate id %d, %d-QAM) = %d SIMD batches of 32 "
                                   "formed across frames as the reference forms them, %s; tracking loops open (the per-symbol synchronisation sums are formed, "
                                   "nobody reads them); one call of the library's batch receiver per step; reference arithmetic incl. the wrapping int8 LLR cast"
                                   "%s; %s (%d samples per frame)"
                                   % (cfg["name"], F, F * w.len_frame, w.fft_size // 1024, F * nb, w.fec_size, w.plp[2], 1 << w.bpc, (F * nb) // 32,
                                      ("from int16 I/Q at the dvbt2_demodulator::execute boundary; stages on GPU: front end (dc, IQ imbalance, NCO, "
                                       "Farrow x2, 64-tap decimator), P1 detect, guard correlation, FFT, P2+data equaliser/freq-deint, TI/cell-deint, "
                                       "demap, LDPC (group 32, max %d trials), BB descramble + bit packing (t2gpu_rx_execute_dev)" % args.trials) if full else
                                      "from the decimated stream in HBM; stages: FFT (guard dropped), P2+data equaliser/freq-deint, TI/cell-deint, demap "
                                      "(t2gpu_rx_fft_eq_demap_dev)",
                                      (", so %d of %d SIMD batches run all trials and are dropped as the reference would" % (dropped, len(ref_trials)))
                                      if ref_trials is not None and dropped else "",
                                      "the host end (per-frame L1-pre/L1-post parse + CRC-32, batch drop rule, BBFRAME de-framing -> TS) runs on the library's worker "
                                      "thread inside the timed region, overlapped with the next step" if full and not args.no_ts_end else
                                      "L1 parsing and TS de-framing (host code) are not inside the timed region", FS),
                       "ldpc_codewords_per_s": round(count / avg_ldpc_s, 1) if full else None,
                       "parallelism": "frame-shard x%d, no collective" % world},
            "roofline": dict({"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 5)}, **dom, kernels=kernels),
        }
        if counters is not None:
            out["host_end"] = {"ts_bytes": ts_bytes, "ts_bytes_per_s": round(ts_bytes / max_s, 1), "bytes_d2h_per_fec_frame": w.k_bch // 8, "counters": counters}
            if ts_kept is not None and ts_kept.size:
                out["host_end"].update(ts_check(ts_kept, ts_bytes, max_s, steps, counters))
        out["snr_db"] = snr_db
        out["stage_ms_sum"] = round(sum(v for k, v in stage_acc.items() if v > 0) / steps, 3)
        out.update(extra)
        out["_cpu_args"] = (cfg_id, w, ui[0], uq[0], full)
        out["_frames"] = (ui, uq)
        return out

    out = run_config(args.config, args.steps, args.warmup, extras=not args.no_extra_legs, check_ts=args.config == 4)
    if rank == 0:
        cpu_args = out.pop("_cpu_args")
        ui_all, uq_all = out.pop("_frames")
    if args.config == 3 and not args.no_extra_legs and not args.only_frames_sweep:
        # BASELINE.json configs[4] (r = 2/3) as an extra key of the same line: with N > 1 the scaling run then reports both codes
        c5 = run_config(5, 2, 1, extras=False)
        if rank == 0:
            c5.pop("_cpu_args"); c5.pop("_frames")
            out["config_5"] = {"metric": c5["metric"], "value": c5["value"], "unit": c5["unit"], "ms_per_step": c5["ms_per_step"],
                               "ldpc_codewords_per_s": c5["config"]["ldpc_codewords_per_s"], "ldpc_ms": c5["roofline"]["avg_launch_ms"],
                               "n_gpus": world, "workload": c5["config"]["workload"][:120] + " ..."}
        if world == 1:
            # BASELINE.json configs[3] (16K, 64-QAM, 16200 r = 1/2 at 12 dB): the leg that DECODES in the reference's own arithmetic --
            # its TS leaves the library's host end inside the clock and every packet is compared with the packets sent
            c4 = run_config(4, 5, 2, extras=False, check_ts=True)
            c4.pop("_cpu_args"); c4.pop("_frames")
            he = c4.get("host_end", {})
            out["config_4"] = {"metric": c4["metric"], "value": c4["value"], "unit": c4["unit"], "ms_per_step": c4["ms_per_step"],
                               "stage_ms_sum": c4["stage_ms_sum"], "snr_db": c4["snr_db"],
                               "ldpc_codewords_per_s": c4["config"]["ldpc_codewords_per_s"], "ldpc_ms": c4["roofline"]["avg_launch_ms"],
                               "ts_bytes": he.get("ts_bytes"), "ts_mbit_per_s": he.get("ts_mbit_per_s"), "ts_packets": he.get("ts_packets"),
                               "ts_packets_that_were_sent": he.get("ts_packets_that_were_sent"), "ts_matches_sent": he.get("ts_matches_sent"),
                               "fec_frames": he.get("counters", {}).get("fec_frames"),
                               "fec_frames_dropped_ldpc": he.get("counters", {}).get("fec_frames_dropped_ldpc"),
                               "fec_frames_dropped_l1": he.get("counters", {}).get("fec_frames_dropped_l1"),
                               "workload": c4["config"]["workload"][:140] + " ..."}
        if world == 1:
            # 32K with a constellation the reference's own arithmetic decodes (64-QAM r = 2/3 at 20 dB): demod -> TS at 32K with the transport
            # stream leaving the host end inside the clock, every packet compared with the packets sent, no extension involved
            c6 = run_config(6, 3, 1, extras=False, check_ts=True)
            c6.pop("_cpu_args"); c6.pop("_frames")
            he = c6.get("host_end", {})
            out["config_32k_64qam"] = {"metric": c6["metric"], "value": c6["value"], "unit": c6["unit"], "ms_per_step": c6["ms_per_step"],
                                       "stage_ms_sum": c6["stage_ms_sum"], "snr_db": c6["snr_db"], "ldpc_codewords_per_s": c6["config"]["ldpc_codewords_per_s"],
                                       "ldpc_ms": c6["roofline"]["avg_launch_ms"], "ts_bytes": he.get("ts_bytes"), "ts_mbit_per_s": he.get("ts_mbit_per_s"),
                                       "ts_packets": he.get("ts_packets"), "ts_packets_that_were_sent": he.get("ts_packets_that_were_sent"),
                                       "ts_matches_sent": he.get("ts_matches_sent"), "fec_frames": he.get("counters", {}).get("fec_frames"),
                                       "fec_frames_dropped_ldpc": he.get("counters", {}).get("fec_frames_dropped_ldpc"),
                                       "workload": c6["config"]["workload"][:160] + " ..."}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            port = cpu_chain_baseline(*cpu_args)
            ref = cpu_reference_baseline(cpu_args[1], ui_all, uq_all) if cpu_args[4] else None
            if ref:                                      # the reference itself, timed here; the oracle's per-stage split and all-core figure beside it
                ref["port"] = port
                out["cpu_baseline"] = ref
            else:
                out["cpu_baseline"] = port
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
