#!/usr/bin/env python3
"""bench.py -- throughput of the DVB-T2 demod -> TS hot path on MI355X (contract: task brief / DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json config 3, "CFG-A"): F complete T2 frames per GPU per step -- 8 MHz, 32K extended, GI 1/128, PP7,
1 P2 + 59 data symbols, one PLP, rotated 256-QAM, LDPC 64800 r=3/4, 202 FEC blocks per frame -- synthetic, built by the
transmitter model in tests/t2_tx.py (P1 + cyclic prefixes + AWGN), resident in HBM as the int16 I/Q samples a tuner
delivers at 64/7 Msps (the dvbt2_demodulator::execute boundary) before the clock starts. One step = front end (dc / IQ
imbalance / NCO / Farrow x2 / 64-tap decimator) -> P1 detection at every frame start -> guard-interval correlation of
every symbol -> FFT with the guard dropped -> P2/data equaliser + frequency de-interleave -> time/cell de-interleave ->
demap -> LDPC (reference SIMD-batch rule, 25 trials) -> BB descramble for all F frames, tracking loops open (zeros and
the nominal resample). Rank 0 prints ONE JSON line: `value` = input IQ samples per second; `roofline` is for the dominant
kernel (LDPC) from HIP events around its launches inside the timed region; `cpu_baseline` times the same chain on one
host core (oracle restatement + the reference's own LDPC where oracle/_ref is loadable) on one frame.
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
LDPC_HBM_BYTES_PER_FRAME = 64800 + 48600    # LLR in + one-bit-per-byte hard decisions out
# memory-side traffic per FEC frame and sweep from the committed PMC passes (profiles/r01_rx_pmc.txt, tools/pmc_passes.sh: this
# bench's launch of 3232 frames x 25 sweeps): 2 x FETCH_SIZE (gfx950 half-count correction, MI355X_MICROARCH.md) + WRITE_SIZE,
# KiB -> bytes
LDPC_TRAFFIC_BYTES_PER_FRAME_SWEEP = (2 * 6.078e6 + 1.0823e7) * 1024 / 3232 / 25
# the bound that does apply to this kernel: vector-ALU issue. SQ_INSTS_VALU per frame-sweep from the same PMC pass (1.1908e10 wave
# instructions per launch of 3232 frames x 25 sweeps), 4 cycles of a SIMD each, 4 SIMDs x 256 CUs at the 2.4 GHz peak engine clock
LDPC_VALU_WAVE_INSTS_PER_FRAME_SWEEP = 1.1908e10 / 3232 / 25
VALU_ISSUE_SLOTS_PER_S = 4 * 256 * 2.4e9 / 4
MODE = (5, 1, 6, 4, 0, 59)                  # FFTSIZE_32K, extended, PP7, GI 1/128, no PAPR, 59 data symbols
L1_POST_SIZE = 350
PLP = (3, 1, 3, 1)                          # 256-QAM, normal FEC frame, r = 3/4, rotated
FRAME_SAMPLES = 2048 + 60 * (32768 + 256)   # P1 + 60 symbols with GI 1/128 = 1 983 488


def make_frames(n_unique, snr_db, seed):
    """n_unique synthetic CFG-A frames as int16 I/Q at the tuner interface: (I [n][FRAME_SAMPLES], Q, sent TS packets, blocks
    per frame)."""
    import numpy as np
    import oracle_lib as ol
    import t2_tx
    m = ol.ora_mode(*MODE)
    cid = ol.code_id(PLP[1], PLP[2])
    nb = t2_tx.plp_blocks_per_frame(m, L1_POST_SIZE, 8100)
    per = nb * (t2_tx.K_BCH[cid] // 1496 + 1)
    frames, sent = [], []
    for f in range(n_unique):
        ts = t2_tx.ts_packets(per, seed + f)
        stream, _, _ = t2_tx.build_plp_frame_cells(cid, PLP[0], PLP[1], PLP[2], ts, nb)
        frames.append(t2_tx.build_frame(m, stream, L1_POST_SIZE, seed + 100 + f, snr_db=None, phase=0.0))
        sent.append(ts)
    i16, q16, flen = t2_tx.iq_stream(frames, 256, 10, snr_db, seed)                  # S2 = 1010: 32K, not mixed
    assert flen == FRAME_SAMPLES
    return i16.reshape(n_unique, flen), q16.reshape(n_unique, flen), sent, nb


def cpu_chain_baseline(i16, q16):
    """One CFG-A frame (int16 I/Q) through the CPU chain on one core: oracle front end (dc / IQ / NCO, Farrow, decimator), P1
    detector, guard correlation, numpy FFT, oracle equaliser / de-interleavers / demapper (C restatement), the reference's own
    LDPC build when loadable (else the C restatement), oracle descrambler."""
    import numpy as np
    import oracle_lib as ol
    m = ol.ora_mode(*MODE)
    cid = ol.code_id(PLP[1], PLP[2])
    nb = (m.c_p2 - 1840 - L1_POST_SIZE + 59 * m.c_data) // 8100
    ldpc, kind = (ol.ref_decode, "reference LDPC + port") if ol.ref() is not None else (ol.ora_decode, "port")
    ti = ol.OraTi(8100, nb)
    sym = 32768 + 256
    t0 = time.perf_counter()
    reps = 0
    while True:                                   # the same frame over and over until ~12 s of single-core work are done
        fo, fa, de, p1 = ol.OraFront(0), ol.OraFarrow(), ol.OraDecim(), ol.OraP1()
        x = np.concatenate((i16, i16[:4096])), np.concatenate((q16, q16[:4096]))     # a little of the next frame: filter delay
        derot, theta = fo.execute(x[0], x[1], [len(x[0])], [0.0], [0.0])
        stream = de(fa(derot, 0.5))
        level = float(np.mean(np.abs(stream.real)) * np.mean(np.abs(stream.imag)))
        r = p1.execute(stream[:3072], 0, True, level)
        first = r["consume"] - r["idx_buffer_sym"] if r["detected"] else 2048 + 17
        cells = []
        for l in range(60):
            s0 = first + l * sym
            ol.ora_cp_frequency_est(stream[s0:s0 + sym], 32768, 256)
            spec = np.fft.fftshift(np.fft.fft(stream[s0 + 256:s0 + sym])).astype(np.complex64)
            out, _, _ = ol.ora_data_symbol(m, l, spec)
            cells.append(out[1840 + L1_POST_SIZE:] if l == 0 else out)
        cells = np.concatenate(cells)[:nb * 8100]
        ti.begin(nb)
        tib = np.zeros(nb * 8100, np.complex64)
        ti.push(cells, tib)
        llr, _, _ = ol.ora_demap(PLP[0], PLP[1], PLP[2], PLP[3], tib)
        for b0 in range(0, (nb // 32) * 32, 32):
            t, bits, _ = ldpc(cid, llr[b0:b0 + 32])
            if t >= 0:
                ol.ora_bch_descramble(cid, bits)
        reps += 1
        el = time.perf_counter() - t0
        if el >= 12.0:
            break
    return {"value": round(reps * FRAME_SAMPLES / el / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d passes over one CFG-A frame from int16 I/Q (front end, P1, 60 symbols, %d of %d FEC blocks in whole SIMD "
                      "batches of 32 through the LDPC, 25 trials each: the wrapped 256-QAM LLRs never converge), %.1f s; stages: "
                      "oracle C restatement + numpy FFT, LDPC = %s" % (reps, (nb // 32) * 32, nb, el, kind)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=16, help="T2 frames per GPU per step")
    ap.add_argument("--snr", type=float, default=22.0)
    ap.add_argument("--trials", type=int, default=25)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clamped-variant", action="store_true", help="skip the informative second leg (profiling runs)")
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # weak scaling: every GPU demodulates args.frames whole T2 frames per step (frames are independent: no collective)
    lo, hi = shard_frames(args.frames * world, world, rank, align=1)
    F = hi - lo
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    from sdr_receiver_dvb_t2_amd.chain import ts_from_bits
    ui, uq, sent, nb = make_frames(2, args.snr, seed=20250614 + 10 * rank)
    d_i = torch.from_numpy(np.concatenate([ui] * ((F + 1) // 2))[:F].reshape(-1)).to(dev)     # int16 [F * FRAME_SAMPLES]
    d_q = torch.from_numpy(np.concatenate([uq] * ((F + 1) // 2))[:F].reshape(-1)).to(dev)

    # The receiver is the library's batch object (t2gpu_rx_*, csrc/t2gpu_rx.cpp): buffers, stage sequencing and launches are C++
    # behind the C ABI; this script hands over two device pointers per step and reads results back. torch is the allocator of the
    # input buffers and the process-group plumbing, nothing else.
    def make_rx(saturate, frames):
        return t2_rx(*MODE, L1_POST_SIZE, *PLP, nb, max_frames=frames, ldpc_trials=args.trials, saturate_llr=saturate, device=local_rank)

    rx = make_rx(False, F)                    # reference semantics: truncating int8 cast in the demapper
    assert rx.frame_len == FRAME_SAMPLES
    count = rx.execute_dev(d_i, d_q, F, first_call=True)                               # thresholds from the level estimate
    level = rx.results(F)["level_detect"]
    # One step = one call = the whole chain over one buffer of F frames (front end .. descrambler), drained by the host once per
    # call because the P1 decisions are host data. Running stages of neighbouring buffers beside the decoder on other streams
    # measured slower, so there is no overlap to lose.
    for _ in range(max(args.warmup, 1)):
        count = rx.execute_dev(d_i, d_q, F, level)
    ref_trials = rx.fetch(count)[1]
    torch.cuda.synchronize(dev)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ldpc_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rx.execute_dev(d_i, d_q, F, level)
        ldpc_ms.append(rx.last_ldpc_ms())          # HIP events around the decoder's launch, on its stream; also drains the step
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    ldpc_frames = count
    max_s, units = aggregate_timing(t1 - t0, F * args.steps, dist if world > 1 else None, dev)
    rx.close()

    # informative second leg (rank 0, N = 1): clamped LLRs (extension) -> the same frames decode; checks the TS bytes
    extra = {}
    if rank == 0 and not args.no_clamped_variant:
        c2 = make_rx(True, 2)
        n2 = c2.execute_dev(d_i[:2 * FRAME_SAMPLES], d_q[:2 * FRAME_SAMPLES], 2, first_call=True)
        b2, t2h = c2.fetch(n2)
        got = ts_from_bits(b2, t2h)
        c2.close()
        want = sent[0].reshape(-1)
        npk = (nb * ((48408 - 80) // 8)) // 187 - 1
        ok = bool((t2h >= 0).all()) and bool(np.array_equal(got[:npk * 188], want[:npk * 188]))
        c3 = make_rx(True, F)
        c3.execute_dev(d_i, d_q, F, first_call=True)
        for _ in range(3):
            c3.execute_dev(d_i, d_q, F, level)
        c3.last_ldpc_ms()
        tc0 = time.perf_counter()
        for _ in range(4):
            c3.execute_dev(d_i, d_q, F, level)
        c3.last_ldpc_ms()
        torch.cuda.synchronize(dev)
        tc1 = time.perf_counter()
        c3.close()
        extra = {"clamped_llr_variant": {"msamples_per_s": round(4 * F * FRAME_SAMPLES / (tc1 - tc0) / 1e6, 1),
                                          "ts_matches_sent": ok, "avg_ldpc_updates": round(float((args.trials - t2h).mean()), 2),
                                          "note": "extension (t2gpu_demap_configure saturate=1): not the reference's arithmetic"}}

    if rank == 0:
        msps = units * FRAME_SAMPLES / max_s / 1e6
        avg_ldpc_s = (sum(ldpc_ms) / len(ldpc_ms)) / 1e3
        achieved = LDPC_HBM_BYTES_PER_FRAME * ldpc_frames / avg_ldpc_s / 1e9
        dropped = int((ref_trials < 0).sum()) if ref_trials is not None else -1
        out = {
            "metric": "IQ Msamples/s demod->TS (32K, 256-QAM, LDPC 64800 r=3/4)",
            "value": round(msps, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(max_s / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16 in, f32 (front end, OFDM, demap) + int8 (LDPC)", "data": "synthetic",
            "config": {"workload": "config 3 (CFG-A): %d T2 frames/GPU/step = %d symbols of 32K, %d FEC frames, from int16 I/Q at the "
                                   "dvbt2_demodulator::execute boundary; stages on GPU: front end (dc, IQ imbalance, NCO, Farrow x2, "
                                   "64-tap decimator), P1 detect, guard correlation, FFT, P2+data equaliser/freq-deint, TI/cell-deint, "
                                   "demap, LDPC (group 32, max %d trials), BB descramble; tracking loops open; one call of the library's batch receiver "
                                   "(t2gpu_rx_execute_dev) per step, drained per call (overlapping other kernels with the decoder measured slower); "
                                   "reference arithmetic incl. the wrapping int8 LLR cast, so %d of %d SIMD batches run all trials and "
                                   "are dropped as the reference would; L1 parsing and TS de-framing (host code) are not inside the "
                                   "timed region (%d samples per frame)"
                                   % (F, F * 60, F * nb, args.trials, dropped, len(ref_trials) if ref_trials is not None else 0,
                                      FRAME_SAMPLES),
                       "ldpc_codewords_per_s": round(ldpc_frames / avg_ldpc_s, 1),
                       "parallelism": "frame-shard x%d, no collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": round(LDPC_TRAFFIC_BYTES_PER_FRAME_SWEEP * ldpc_frames * args.trials),
                         "kernel": "ldpc_decode_kernel<12,12,4>", "avg_launch_ms": round(avg_ldpc_s * 1e3, 3),
                         "share_of_step": round(avg_ldpc_s / (max_s / args.steps), 3),
                         "valu_issue_frac": round(LDPC_VALU_WAVE_INSTS_PER_FRAME_SWEEP * ldpc_frames * args.trials / avg_ldpc_s / VALU_ISSUE_SLOTS_PER_S, 3),
                         "note": "the LDPC is VALU/LDS-bound by construction (DESIGN.md): HBM sees each LLR once and each bit once; "
                                 "traffic = PMC-measured bytes per frame-sweep (profiles/r01_rx_pmc.txt) x frames x sweeps (the check-node records, "
                                 "streaming through L2 / Infinity Cache); valu_issue_frac = PMC-measured vector instructions of this workload / "
                                 "the measured launch time / the chip's vector issue rate: the bound this kernel actually runs against"},
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_chain_baseline(ui[0], uq[0])
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
