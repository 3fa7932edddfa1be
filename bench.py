#!/usr/bin/env python3
"""bench.py -- throughput of the DVB-T2 demod -> TS hot path on MI355X (contract: task brief / DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}] [--frames F] [--trials T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload (BASELINE.json config 3, "CFG-A"; --config 5 = the same with r = 2/3, --config 4 = 16K / 64-QAM / 16200 r = 1/2,
--config 2 = FFT + equalise + de-interleave + demap only): F complete T2 frames per GPU per step -- 8 MHz, 32K extended, GI 1/128, PP7,
1 P2 + 59 data symbols, one PLP, rotated 256-QAM, LDPC 64800 r=3/4, 202 FEC blocks per frame -- synthetic, built by the
transmitter model in tests/t2_tx.py (P1 + cyclic prefixes + AWGN), resident in HBM as the int16 I/Q samples a tuner
delivers at 64/7 Msps (the dvbt2_demodulator::execute boundary) before the clock starts. One step = front end (dc / IQ
imbalance / NCO / Farrow x2 / 64-tap decimator) -> P1 detection at every frame start -> guard-interval correlation of
every symbol -> FFT with the guard dropped -> P2/data equaliser + frequency de-interleave -> time/cell de-interleave ->
demap -> LDPC (reference SIMD-batch rule, 25 trials) -> BB descramble for all F frames, tracking loops open (zeros and
the nominal resample). Rank 0 prints ONE JSON line: `value` = input IQ samples per second; `roofline` is for the dominant
kernel (LDPC) from HIP events around its launches inside the timed region, `roofline.kernels` has every stage (algorithmic HBM
bytes / measured stage time); `cpu_baseline` times the same chain on ALL host cores (one process per core: oracle restatement +
the reference's own LDPC where oracle/_ref is loadable) on one frame each, and names the CPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# memory-side traffic per FEC frame and sweep from the committed PMC passes (profiles/r02_rx_pmc.txt, tools/pmc_passes.sh: this
# bench's launch of 7676 frames x 25 sweeps of ldpc_decode2_kernel<12,12,4>): 2 x FETCH_SIZE (gfx950 half-count correction,
# MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes. It is the per-link message bytes of the two-frame kernel (one byte per link and
# frame, 16 B per lane and layer, read and written once per sweep) streaming through L2 / Infinity Cache.
LDPC_TRAFFIC_BYTES_PER_FRAME_SWEEP = (2 * 2.4553e7 + 5.0817e7) * 1024 / 7676 / 25
# SQ_INSTS_VALU of the same launch; a wave instruction occupies a SIMD for 4 cycles, 4 SIMDs x 256 CUs, 2.4 GHz (GRBM_GUI_ACTIVE of the
# launch / its duration = 2.39 GHz per XCD)
LDPC_VALU_WAVE_INSTS_PER_FRAME_SWEEP = 1.6021e10 / 7676 / 25
VALU_ISSUE_SLOTS_PER_S = 4 * 256 * 2.4e9 / 4

# BASELINE.json configs that run on one GPU. mode = (fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data),
# plp = (modulation, fec_type, code_rate, rotation); frames = T2 frames per GPU per step (chosen so that the FEC frames fill whole
# rounds of the decoder's 16 resident SIMD-batch slots: 38 x 202 = 7676 frames = 240 batches = 15 rounds; with 16 frames the 101
# batches took 7 rounds for 6.3 rounds of work)
CONFIGS = {
    2: dict(name="config 2 (CFG-A, FFT + equalise + de-interleave + demap only)", mode=(5, 1, 6, 4, 0, 59), lps=350, plp=(3, 1, 3, 1), frames=38, s2=10,
            metric="IQ Msamples/s through FFT + equaliser + demap (32K, 256-QAM)"),
    3: dict(name="config 3 (CFG-A)", mode=(5, 1, 6, 4, 0, 59), lps=350, plp=(3, 1, 3, 1), frames=38, s2=10,
            metric="IQ Msamples/s demod->TS (32K, 256-QAM, LDPC 64800 r=3/4)"),
    4: dict(name="config 4 (CFG-B)", mode=(4, 1, 6, 4, 0, 40), lps=200, plp=(2, 0, 0, 1), frames=40, s2=8,
            metric="IQ Msamples/s demod->TS (16K, 64-QAM, LDPC 16200 r=1/2)"),
    5: dict(name="config 5 (CFG-C)", mode=(5, 1, 6, 4, 0, 59), lps=350, plp=(3, 1, 2, 1), frames=38, s2=10,
            metric="IQ Msamples/s demod->TS (32K, 256-QAM, LDPC 64800 r=2/3)"),
}
K_LDPC = {0: (7200, 9720, 10800, 11880, 12600, 13320), 1: (32400, 38880, 43200, 48600, 51840, 54000)}
K_BCH = {0: (7032, 9552, 10632, 11712, 12432, 13152), 1: (32208, 38688, 43040, 48408, 51648, 53840)}


class Workload(object):
    def __init__(self, cfg):
        import oracle_lib as ol
        import t2_tx
        self.cfg, self.mode, self.lps, self.plp = cfg, cfg["mode"], cfg["lps"], cfg["plp"]
        self.m = ol.ora_mode(*self.mode)
        m = self.m
        self.guard = {0: m.fft_size // 32, 1: m.fft_size // 16, 2: m.fft_size // 8, 3: m.fft_size // 4, 4: m.fft_size // 128}[self.mode[3]]
        self.sym = m.fft_size + self.guard
        self.frame_samples = 2048 + m.len_frame * self.sym
        self.fec_size = 64800 if self.plp[1] else 16200
        self.bpc = 2 * (self.plp[0] + 1)
        self.cpf = self.fec_size // self.bpc
        self.cid = ol.code_id(self.plp[1], self.plp[2])
        self.nb = t2_tx.plp_blocks_per_frame(m, self.lps, self.cpf)
        self.frame_cells = (m.c_p2 - 1840 - self.lps) + (m.n_data - m.l_fc) * m.c_data + m.l_fc * m.n_fc
        self.k_ldpc, self.k_bch = K_LDPC[self.plp[1]][self.plp[2]], K_BCH[self.plp[1]][self.plp[2]]

    def stage_bytes(self, F):
        """ALGORITHMIC HBM bytes of every stage for F frames (each datum once in, once out; tables excluded)."""
        m = self.m
        syms = F * m.len_frame
        cells = F * self.nb * self.cpf
        return {
            "front": 12 * F * self.frame_samples,                                    # int16 I/Q in, complex64 out
            "guard_corr": syms * 2 * self.guard * 8,
            "fft": syms * 16 * m.fft_size,
            "equalise": F * 8 * (m.k_total * m.len_frame + m.c_p2 + (m.n_data - m.l_fc) * m.c_data + m.l_fc * m.n_fc),
            "ti": 16 * cells,
            "demap": cells * (8 + 8 + self.bpc),                                     # statistics pass, LLR pass in, LLRs out
            "ldpc": F * self.nb * (self.fec_size + self.k_ldpc),
            "descramble": F * self.nb * (self.k_ldpc + self.k_bch),
        }


def make_frames(w, n_unique, snr_db, seed):
    """n_unique synthetic frames of workload w as int16 I/Q at the tuner interface: (I [n][frame_samples], Q, sent TS packets)."""
    import t2_tx
    per = w.nb * (w.k_bch // 1496 + 1)
    frames, sent = [], []
    for f in range(n_unique):
        ts = t2_tx.ts_packets(per, seed + f)
        stream, _, _ = t2_tx.build_plp_frame_cells(w.cid, w.plp[0], w.plp[1], w.plp[2], ts, w.nb)
        frames.append(t2_tx.build_frame(w.m, stream, w.lps, seed + 100 + f, snr_db=None, phase=0.0))
        sent.append(ts)
    i16, q16, flen = t2_tx.iq_stream(frames, w.guard, w.cfg["s2"], snr_db, seed)
    assert flen == w.frame_samples
    return i16.reshape(n_unique, flen), q16.reshape(n_unique, flen), sent


def _cpu_worker(args):
    """One frame (int16 I/Q) through the CPU chain on one core, over and over for `seconds`: oracle front end (dc / IQ / NCO,
    Farrow, decimator), P1 detector, guard correlation, numpy FFT, oracle equaliser / de-interleavers / demapper (C restatement), the
    reference's own LDPC build when loadable (else the C restatement), oracle descrambler. Returns (passes, elapsed, LDPC kind)."""
    cfg_id, i16, q16, seconds, full = args
    import numpy as np
    import oracle_lib as ol
    w = Workload(CONFIGS[cfg_id])
    m = w.m
    ldpc, kind = (ol.ref_decode, "reference LDPC + port") if ol.ref() is not None else (ol.ora_decode, "port")
    ti = ol.OraTi(w.cpf, w.nb)
    t0 = time.perf_counter()
    reps = 0
    while True:
        fo, fa, de, p1 = ol.OraFront(0), ol.OraFarrow(), ol.OraDecim(), ol.OraP1()
        x = np.concatenate((i16, i16[:4096])), np.concatenate((q16, q16[:4096]))     # a little of the next frame: filter delay
        derot, theta = fo.execute(x[0], x[1], [len(x[0])], [0.0], [0.0])
        stream = de(fa(derot, 0.5))
        level = float(np.mean(np.abs(stream.real)) * np.mean(np.abs(stream.imag)))
        r = p1.execute(stream[:3072], 0, True, level)
        first = r["consume"] - r["idx_buffer_sym"] if r["detected"] else 2048 + 17
        cells = []
        for l in range(m.len_frame):
            s0 = first + l * w.sym
            ol.ora_cp_frequency_est(stream[s0:s0 + w.sym], m.fft_size, w.guard)
            spec = np.fft.fftshift(np.fft.fft(stream[s0 + w.guard:s0 + w.sym])).astype(np.complex64)
            out, _, _ = ol.ora_data_symbol(m, l, spec)
            cells.append(out[1840 + w.lps:] if l == 0 else out)
        cells = np.concatenate(cells)[:w.nb * w.cpf]
        ti.begin(w.nb)
        tib = np.zeros(w.nb * w.cpf, np.complex64)
        ti.push(cells, tib)
        llr, _, _ = ol.ora_demap(w.plp[0], w.plp[1], w.plp[2], w.plp[3], tib)
        if full:
            for b0 in range(0, (w.nb // 32) * 32, 32):
                t, bits, _ = ldpc(w.cid, llr[b0:b0 + 32])
                if t >= 0:
                    ol.ora_bch_descramble(w.cid, bits)
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return reps, el, kind


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_chain_baseline(cfg_id, w, i16, q16, full):
    """The chain of _cpu_worker on every host core at once (one process per core, each on the same frame), ~12 s."""
    import multiprocessing as mp
    cores = max(1, min(len(os.sched_getaffinity(0)), 128))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(cfg_id, i16, q16, 12.0, full)] * cores)
    wall = time.perf_counter() - t0
    rate = sum(r[0] * w.frame_samples / r[1] for r in res) / 1e6
    one = max(r[0] * w.frame_samples / r[1] for r in res) / 1e6
    return {"value": round(rate, 3), "unit": "Msamples/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "best_single_core": round(one, 3),
            "sample": "%d processes x ~12 s (%.1f s wall), each passing one %s frame from int16 I/Q over and over (front end, P1, %d symbols"
                      "%s); stages: oracle C restatement + numpy FFT, LDPC = %s"
                      % (cores, wall, w.cfg["name"], w.m.len_frame,
                         ", %d of %d FEC blocks in whole SIMD batches of 32 through the LDPC, 25 trials each" % ((w.nb // 32) * 32, w.nb) if full
                         else ", up to the LLRs", res[0][2])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json config (3 = the metric's own)")
    ap.add_argument("--frames", type=int, default=0, help="T2 frames per GPU per step (0: the config's default)")
    ap.add_argument("--snr", type=float, default=22.0)
    ap.add_argument("--trials", type=int, default=25, help="LDPC trial limit (the reference's TRIALS = 25, ldpc_decoder.h)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the informative legs: 50-trial point, clamped-LLR variant (profiling runs)")
    ap.add_argument("--no-clamped-variant", action="store_true", help="(kept for old command lines) same as --no-extra-legs")
    args = ap.parse_args()
    if args.no_clamped_variant:
        args.no_extra_legs = True

    import numpy as np
    import torch
    import torch.distributed as dist
    import sdr_receiver_dvb_t2_amd as pkg
    from sdr_receiver_dvb_t2_amd.shard import shard_frames, aggregate_timing

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # dry run of the N > 1 control flow on a one-GPU box: T2GPU_BENCH_ONE_DEVICE=1 puts every rank on device 0 and uses gloo (RCCL
    # refuses two ranks on one device); the driver's runs use one GPU per rank and RCCL
    one_device = os.environ.get("T2GPU_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = CONFIGS[args.config]
    w = Workload(cfg)
    full = args.config != 2
    frames_per_gpu = args.frames or cfg["frames"]
    # weak scaling: every GPU demodulates frames_per_gpu whole T2 frames per step (frames are independent: no collective)
    lo, hi = shard_frames(frames_per_gpu * world, world, rank, align=1)
    F = hi - lo
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    from sdr_receiver_dvb_t2_amd.chain import ts_from_bits
    ui, uq, sent = make_frames(w, 2, args.snr if args.config != 4 else 16.0, seed=20250614 + 10 * rank)
    nb, FS = w.nb, w.frame_samples
    d_i = torch.from_numpy(np.concatenate([ui] * ((F + 1) // 2))[:F].reshape(-1)).to(dev)     # int16 [F * frame_samples]
    d_q = torch.from_numpy(np.concatenate([uq] * ((F + 1) // 2))[:F].reshape(-1)).to(dev)

    # The receiver is the library's batch object (t2gpu_rx_*, csrc/t2gpu_rx.cpp): buffers, stage sequencing and launches are C++
    # behind the C ABI; this script hands over two device pointers per step and reads results back. torch is the allocator of the
    # input buffers and the process-group plumbing, nothing else.
    def make_rx(saturate, frames, trials):
        return t2_rx(*w.mode, w.lps, *w.plp, nb, max_frames=frames, ldpc_trials=trials, saturate_llr=saturate, device=local_rank)

    def timed_leg(rx, steps, warmup, level):
        """W untimed + K timed steps bracketed by barrier + synchronize; returns (seconds, per-stage ms sums, LDPC ms list)."""
        for _ in range(max(warmup, 1)):
            step(rx, level)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        acc, ldpc = {}, []
        t0 = time.perf_counter()
        for _ in range(steps):
            step(rx, level)
            for k, v in rx.stage_ms().items():     # HIP events between the stages on the call's stream; also drains the step
                if v >= 0:
                    acc[k] = acc.get(k, 0.0) + v
            if full:
                ldpc.append(rx.last_ldpc_ms())
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t0, acc, ldpc

    if full:
        def step(rx, level):
            return rx.execute_dev(d_i, d_q, F, level)
    else:
        def step(rx, level):                       # config 2: the frames' stream and positions stay in the handle from the set-up call
            return rx.fft_eq_demap_dev(F)

    rx = make_rx(False, F, args.trials)            # reference semantics: truncating int8 cast in the demapper
    assert rx.frame_len == FS
    count = rx.execute_dev(d_i, d_q, F, first_call=True)                               # thresholds from the level estimate
    level = rx.results(F)["level_detect"]
    # One step = one call = the whole chain over one buffer of F frames (front end .. descrambler), drained by the host once per
    # call because the P1 decisions are host data. Running stages of neighbouring buffers beside the decoder on other streams
    # measured slower, so there is no overlap to lose.
    elapsed, stage_acc, ldpc_ms = timed_leg(rx, args.steps, args.warmup, level)
    ref_trials = rx.fetch(count)[1] if full else None
    max_s, units = aggregate_timing(elapsed, F * args.steps, dist if world > 1 else None, None if one_device else dev)
    rx.close()

    extra = {}
    if rank == 0 and world == 1 and full and not args.no_extra_legs:     # N = 1 only: timed_leg's barriers are collective
        # (i) the BASELINE config's "50 iters" point beside the reference's own TRIALS = 25
        r50 = make_rx(False, F, 50)
        r50.execute_dev(d_i, d_q, F, first_call=True)
        e50, _, l50 = timed_leg(r50, 2, 1, level)
        r50.close()
        extra["trials_50"] = {"msamples_per_s": round(2 * F * FS / e50 / 1e6, 1), "ldpc_ms": round(sum(l50) / len(l50), 3),
                              "note": "same workload with the LDPC trial limit at 50 (BASELINE.json config text); the reference itself stops at 25"}
        # (ii) clamped LLRs (extension) -> the same frames decode; checks the TS bytes
        c2 = make_rx(True, 2, args.trials)
        n2 = c2.execute_dev(d_i[:2 * FS], d_q[:2 * FS], 2, first_call=True)
        b2, t2h = c2.fetch(n2)
        got = ts_from_bits(b2, t2h)
        c2.close()
        want = sent[0].reshape(-1)
        npk = (nb * ((w.k_bch - 80) // 8)) // 187 - 1
        ok = bool((t2h >= 0).all()) and bool(np.array_equal(got[:npk * 188], want[:npk * 188]))
        c3 = make_rx(True, F, args.trials)
        c3.execute_dev(d_i, d_q, F, first_call=True)
        e3, _, _ = timed_leg(c3, 3, 2, level)
        c3.close()
        extra["clamped_llr_variant"] = {"msamples_per_s": round(3 * F * FS / e3 / 1e6, 1), "ts_matches_sent": ok,
                                        "avg_ldpc_updates": round(float((args.trials - t2h).mean()), 2),
                                        "note": "extension (t2gpu_demap_configure saturate=1): not the reference's arithmetic"}

    if rank == 0:
        msps = units * FS / max_s / 1e6
        sb = w.stage_bytes(F)
        kernels = []
        for k in ("front", "guard_corr", "fft", "equalise", "ti", "demap", "ldpc", "descramble"):
            if k in stage_acc and stage_acc[k] > 0:
                ms = stage_acc[k] / args.steps
                gbs = sb[k] / (ms * 1e-3) / 1e9
                kernels.append({"stage": k, "ms": round(ms, 4), "algorithmic_bytes": int(sb[k]), "achieved_GBs": round(gbs, 1),
                                "frac": round(gbs / HBM_PEAK_GBS, 4)})
        if "p1" in stage_acc:
            kernels.append({"stage": "p1 (incl. the step's one host round trip)", "ms": round(stage_acc["p1"] / args.steps, 4)})
        if full:
            ldpc_frames = count
            avg_ldpc_s = (sum(ldpc_ms) / len(ldpc_ms)) / 1e3
            achieved = sb["ldpc"] / avg_ldpc_s / 1e9
            dom = {"kernel": "ldpc_decode2_kernel (two FEC frames per workgroup, packed 16-bit halves)", "avg_launch_ms": round(avg_ldpc_s * 1e3, 3),
                   "traffic": round(LDPC_TRAFFIC_BYTES_PER_FRAME_SWEEP * ldpc_frames * args.trials) if args.config == 3 else None,
                   "share_of_step": round(avg_ldpc_s / (max_s / args.steps), 3),
                   "valu_issue_frac": round(LDPC_VALU_WAVE_INSTS_PER_FRAME_SWEEP * ldpc_frames * args.trials / avg_ldpc_s / VALU_ISSUE_SLOTS_PER_S, 3)
                   if args.config == 3 else None,
                   "note": "the LDPC is VALU/LDS-bound by construction (DESIGN.md): HBM sees each LLR once and each bit once; traffic = "
                           "PMC-measured bytes per frame-sweep (profiles/) x frames x sweeps (the check-node records, streaming through L2 / "
                           "Infinity Cache); valu_issue_frac = PMC-measured vector instructions / measured launch time / the chip's vector issue rate"}
        else:
            top = max((k for k in kernels if "frac" in k), key=lambda k: k["ms"])
            achieved = top["achieved_GBs"]
            dom = {"kernel": top["stage"], "avg_launch_ms": top["ms"], "traffic": None}
        dropped = int((ref_trials < 0).sum()) if ref_trials is not None else 0
        out = {
            "metric": cfg["metric"],
            "value": round(msps, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(max_s / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16 in, f32 (front end, OFDM, demap) + int8 (LDPC)", "data": "synthetic",
            "config": {"workload": "%s: %d T2 frames/GPU/step = %d symbols of %dK, %d FEC frames (%d bits, code rate id %d, %d-QAM), %s; "
                                   "tracking loops open (the per-symbol synchronisation sums are formed, nobody reads them); one call of the "
                                   "library's batch receiver per step, drained per call; reference arithmetic incl. the wrapping int8 LLR cast"
                                   "%s; L1 parsing and TS de-framing (host code) are not inside the timed region (%d samples per frame)"
                                   % (cfg["name"], F, F * w.m.len_frame, w.m.fft_size // 1024, F * nb, w.fec_size, w.plp[2], 1 << w.bpc,
                                      ("from int16 I/Q at the dvbt2_demodulator::execute boundary; stages on GPU: front end (dc, IQ imbalance, NCO, "
                                       "Farrow x2, 64-tap decimator), P1 detect, guard correlation, FFT, P2+data equaliser/freq-deint, TI/cell-deint, "
                                       "demap, LDPC (group 32, max %d trials), BB descramble (t2gpu_rx_execute_dev)" % args.trials) if full else
                                      "from the decimated stream in HBM; stages: FFT (guard dropped), P2+data equaliser/freq-deint, TI/cell-deint, demap "
                                      "(t2gpu_rx_fft_eq_demap_dev)",
                                      (", so %d of %d SIMD batches run all trials and are dropped as the reference would" % (dropped, len(ref_trials)))
                                      if ref_trials is not None and dropped else "", FS),
                       "ldpc_codewords_per_s": round(count / avg_ldpc_s, 1) if full else None,
                       "parallelism": "frame-shard x%d, no collective" % world},
            "roofline": dict({"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 5)}, **dom, kernels=kernels),
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_chain_baseline(args.config, w, ui[0], uq[0], full)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
