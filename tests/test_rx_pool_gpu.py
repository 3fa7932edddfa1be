"""GPU tier: the C-ABI multi-device receiver (include/t2gpu.h, t2gpu_rx_pool_*; csrc/t2gpu_rx_pool.cpp). Pool members on device 0 (a test
box has one GPU; on a node they are its eight) each decode a batch-aligned contiguous share of a buffer of T2 frames on a thread of their
own, nothing crosses between them, and ONE de-framer sees the packed BBFRAMEs in frame order: the transport stream is, byte for byte, what
a single t2gpu_rx writes for the same frames. Called through the C ABI with ctypes, no torch.distributed anywhere."""
import ctypes

import numpy as np
import pytest

import ref_cases as rc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shard_stream(built):
    # 16K / 64-QAM / 16200 r = 1/2, 16 FEC blocks per T2 frame: alignment 2 T2 frames; 12 frames
    mode, lps, plp, nb, frames = (4, 1, 6, 4, 0, 8), 200, (2, 0, 0), 16, 12
    rc.CARRY_CASES["pool"] = dict(mode=mode, lps=lps, plp=plp, nb=nb, frames=frames, snr=16.0, seed=41, s2=8, iq_snr=20.0)
    i16, q16 = rc.carry_iq("pool")
    return mode, lps, plp, nb, frames, np.ascontiguousarray(i16.reshape(-1)), np.ascontiguousarray(q16.reshape(-1))


def single_call_ts(mode, lps, plp, nb, frames, i16, q16):
    import torch
    from sdr_receiver_dvb_t2_amd.receiver import t2_rx
    rx = t2_rx(*mode, lps, *plp, 1, nb, max_frames=frames)
    rx.ts_enable(0, l1_check=False)
    n = rx.execute_dev(torch.from_numpy(i16).cuda(), torch.from_numpy(q16).cuda(), frames, first_call=True)
    assert n == frames * nb
    ts = rx.ts_read(wait_all=True).copy()
    c = rx.ts_counters()
    rx.close()
    return ts, c


@pytest.mark.parametrize("members", [1, 2, 3])
def test_pool_members_give_the_single_receivers_transport_stream(shard_stream, members):
    from sdr_receiver_dvb_t2_amd._lib import lib
    from sdr_receiver_dvb_t2_amd.receiver import rx_config, rx_geometry
    mode, lps, plp, nb, frames, i16, q16 = shard_stream
    want, wc = single_call_ts(mode, lps, plp, nb, frames, i16, q16)
    assert want.size > 100000 and wc["fec_frames_dropped_ldpc"] == 0
    l = lib()
    cfg = rx_config(0, 0.0, *mode, lps, *plp, 1, nb, frames, 32, 25, 0)
    devices = (ctypes.c_int * members)(*([0] * members))
    pool = l.t2gpu_rx_pool_create(ctypes.byref(cfg), devices, members, 0)
    assert pool, l.t2gpu_last_error().decode()
    try:
        assert l.t2gpu_rx_pool_frame_alignment(pool) == 2
        geo = rx_geometry()
        assert l.t2gpu_rx_pool_info(pool, ctypes.byref(geo)) == 0 and geo.fec_frames_per_t2_frame == nb
        # the shares: contiguous, disjoint, aligned, covering the buffer
        lo, hi = ctypes.c_int(), ctypes.c_int()
        edges = []
        for k in range(members):
            assert l.t2gpu_rx_pool_share(pool, frames, k, ctypes.byref(lo), ctypes.byref(hi)) == 0
            edges.append((lo.value, hi.value))
            assert lo.value % 2 == 0 and hi.value % 2 == 0
        assert edges[0][0] == 0 and edges[-1][1] == frames and all(a[1] == b[0] for a, b in zip(edges[:-1], edges[1:]))
        # a buffer that is not a whole number of alignments is refused, not mis-batched
        assert l.t2gpu_rx_pool_execute(pool, i16.ctypes.data, q16.ctypes.data, 3) == -3
        # two calls of six frames each = the stream in two buffers (6 is a multiple of the alignment)
        out = np.zeros(want.size + 4096, np.uint8)
        got = 0
        per = frames // 2 * geo.frame_len
        for call in range(2):
            n = l.t2gpu_rx_pool_execute(pool, i16[call * per:].ctypes.data, q16[call * per:].ctypes.data, frames // 2)
            assert n == frames // 2 * nb, (n, l.t2gpu_last_error().decode())
            got += l.t2gpu_rx_pool_ts_read(pool, out[got:].ctypes.data, out.size - got)
        counts = (ctypes.c_int64 * 6)()
        secs = ctypes.c_double()
        assert l.t2gpu_rx_pool_counters(pool, counts, ctypes.byref(secs)) == 0
        assert counts[0] == frames * nb and counts[1] == 0 and counts[2] == 0
        assert got == want.size and np.array_equal(out[:got], want), (got, want.size)
    finally:
        l.t2gpu_rx_pool_destroy(pool)


def test_pool_refuses_a_device_that_is_not_there(built):
    from sdr_receiver_dvb_t2_amd._lib import lib
    from sdr_receiver_dvb_t2_amd.receiver import rx_config
    l = lib()
    cfg = rx_config(0, 0.0, 4, 1, 6, 4, 0, 8, 200, 2, 0, 0, 1, 16, 4, 32, 25, 0)
    devices = (ctypes.c_int * 2)(0, 99)
    assert not l.t2gpu_rx_pool_create(ctypes.byref(cfg), devices, 2, 0)
    assert b"device" in l.t2gpu_last_error()
