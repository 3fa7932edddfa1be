"""Frame-batch sharding across the GPUs of one node: every FEC frame / SIMD batch is independent
given the mode tables, so ranks take disjoint, contiguous, batch-aligned ranges and never exchange data on the data
path. The only collectives are the bench's barrier and the max-over-ranks of the timed region -- CPU tensors over gloo; RCCL is not
used anywhere. The same sharding inside ONE process, over the devices of a node, is the C ABI's t2gpu_rx_pool_* (csrc/t2gpu_rx_pool.cpp)."""


def shard_frames(total_frames, world_size, rank, align=32):
    """Contiguous range [lo, hi) of `total_frames` for `rank`; boundaries are multiples of `align` (a reference SIMD
    batch never straddles two GPUs). Ranges of all ranks are disjoint and cover [0, total_frames)."""
    assert 0 <= rank < world_size and total_frames >= 0 and align >= 1
    units = (total_frames + align - 1) // align
    base, rem = divmod(units, world_size)
    lo_u = rank * base + min(rank, rem)
    hi_u = lo_u + base + (1 if rank < rem else 0)
    return min(lo_u * align, total_frames), min(hi_u * align, total_frames)


def aggregate_timing(local_seconds, local_units, dist=None, device=None):
    """(max seconds over ranks, sum of units over ranks). With `dist` None (single process) returns the inputs."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_seconds, local_units
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def frame_alignment(fec_frames_per_t2_frame, group=32):
    """Smallest number of T2 frames whose FEC frames are a whole number of reference SIMD batches: shard boundaries at multiples of
    it leave every batch of `group` FEC frames exactly as a single sequential receiver would form it (llr_demapper.cpp:742-760 fills a
    batch across T2-frame boundaries), so drops and stop decisions are the same whichever GPU decodes a batch."""
    from math import gcd
    return group // gcd(group, fec_frames_per_t2_frame)


class ordered_receiver(object):
    """A stream of T2 frames over N devices with the transport stream coming out of ONE de-framer in frame order.

    Every rank decodes a contiguous, batch-aligned range of the frames on its own GPU -- no collective on the data path. What crosses
    ranks is the result: the descrambled BBFRAMEs (packed to bits) and the per-batch LDPC verdicts are gathered to rank 0 in rank
    order = frame order, where the sequential epilogue runs as in the reference: bb_de_header carries a split TS packet (and in normal
    mode the running CRC-8) from one BBFRAME into the next (bb_de_header.cpp:166-322), so it must see the frames of all ranks as one
    stream. `decode(lo, hi)` -> (uint8 [n_fec][k_bch] one bit per byte, int32 trials per SIMD batch) for T2 frames [lo, hi) -- or, with
    packed_k_bch = k_bch, the rows already packed MSB first, [n_fec][k_bch / 8], as t2_rx.fetch_packed delivers them; in production it
    is a t2_rx on the rank's GPU, in the CPU tests a stub."""

    def __init__(self, decode, fec_frames_per_t2_frame, group=32, need_plp=0, dist=None, packed_k_bch=None):
        self.decode, self.per_frame, self.group, self.need_plp, self.dist = decode, fec_frames_per_t2_frame, group, need_plp, dist
        self.packed_k_bch = packed_k_bch
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self._bbdh = None
        self._out = None

    def _cpu_group(self):
        if getattr(self, "_grp", None) is None:
            self._grp = self.dist.group.WORLD if self.dist.get_backend() == "gloo" else self.dist.new_group(backend="gloo")
        return self._grp

    def close(self):
        if self._bbdh is not None:
            from ._lib import lib
            lib().t2gpu_bbdh_destroy(self._bbdh)
            self._bbdh = None

    def execute(self, total_frames):
        """Decode `total_frames` T2 frames (this rank its share); returns the TS bytes on rank 0, None elsewhere. The shares travel as
        two tensors per rank (packed rows, batch verdicts; `dist.gather`, fixed shapes every rank derives from the shard table, the
        short shares padded) and rank 0 de-frames each share with ONE call into the library (t2gpu_bbdh_execute_packed_rows: the
        LDPC drop rule and bb_de_header row after row in C). last_counts / last_merge_seconds: what that loop did and how long it took."""
        import time
        import numpy as np
        align = frame_alignment(self.per_frame, self.group)
        shares = [shard_frames(total_frames, self.world, r, align) for r in range(self.world)]
        lo, hi = shares[self.rank]
        k_bch = self.packed_k_bch
        if hi > lo:
            bits, trials = self.decode(lo, hi)
            bits = np.ascontiguousarray(bits, np.uint8)
            trials = np.ascontiguousarray(trials, np.int32)
            if k_bch:
                rows = bits
            else:
                k_bch = bits.shape[1]
                rows = np.packbits(bits, axis=1)
        else:
            rows, trials = np.zeros((0, 0), np.uint8), np.zeros(0, np.int32)
        n_rows = [(b - a) * self.per_frame for a, b in shares]
        if self.world > 1:
            import torch
            # every rank needs the row width to shape its tensors: the ranks that decoded something know it
            kb = torch.tensor([k_bch or 0], dtype=torch.int64)
            # The rows are RESULTS on their way to the one de-framer, host memory to host memory: CPU tensors over gloo whatever the
            # default backend is (SURVEY.md 8e / north_star: no RCCL on this path; a process group that only has nccl gets a gloo group
            # of the same ranks beside it, made once).
            grp = self._cpu_group()
            self.dist.all_reduce(kb, op=self.dist.ReduceOp.MAX, group=grp)
            k_bch = int(kb.item())
            row_bytes = (k_bch + 7) // 8
            max_rows = max(n_rows)
            max_batches = (max_rows + self.group - 1) // self.group
            t_rows = torch.zeros((max_rows, row_bytes), dtype=torch.uint8)
            t_tr = torch.zeros(max_batches, dtype=torch.int32)
            if rows.shape[0]:
                t_rows[:rows.shape[0]] = torch.from_numpy(rows)
                t_tr[:trials.shape[0]] = torch.from_numpy(trials)
            g_rows = [torch.empty_like(t_rows) for _ in range(self.world)] if self.rank == 0 else None
            g_tr = [torch.empty_like(t_tr) for _ in range(self.world)] if self.rank == 0 else None
            self.dist.gather(t_rows, g_rows, dst=0, group=grp)       # rank order = frame order; k_bch / 8 bytes per FEC frame
            self.dist.gather(t_tr, g_tr, dst=0, group=grp)
            if self.rank != 0:
                return None
            parts = [(g_rows[r].numpy()[:n_rows[r]], g_tr[r].numpy()) for r in range(self.world)]
        else:
            parts = [(rows, trials)]
        from ._lib import lib
        l = lib()
        if self._bbdh is None:
            self._bbdh = l.t2gpu_bbdh_create(self.need_plp)          # ONE de-framer for the whole stream: its packet state crosses ranks
        total_rows = sum(p[0].shape[0] for p in parts)
        need = total_rows * ((k_bch or 0) // 8) + (k_bch or 0) // 8 + 376 * (total_rows + 1)
        if self._out is None or self._out.size < need:               # the TS buffer lives across calls; its pages are touched here, once,
            self._out = np.empty(need, np.uint8)                     # not inside the de-framing loop (first touch costs more than the copy)
            self._out.fill(0)
        out = self._out
        counts, acc = np.zeros(6, np.int64), np.zeros(6, np.int64)
        used = 0
        t0 = time.perf_counter()
        for prow, ptr in parts:
            if prow.shape[0] == 0:
                continue
            prow, ptr = np.ascontiguousarray(prow), np.ascontiguousarray(ptr, np.int32)
            n = l.t2gpu_bbdh_execute_packed_rows(self._bbdh, self.need_plp, k_bch, prow.ctypes.data, prow.shape[0], prow.strides[0],
                                                 ptr.ctypes.data, self.group, out[used:].ctypes.data, out.size - used, counts.ctypes.data)
            if n < 0:
                raise RuntimeError("t2gpu_bbdh_execute_packed_rows: %s" % l.t2gpu_last_error().decode())
            used += n
            acc += counts
        self.last_merge_seconds = time.perf_counter() - t0
        self.last_counts = dict(zip(("rows", "dropped_ldpc", "bbheader_crc_errors", "skipped", "ts_packet_errors", "resync"), (int(v) for v in acc)))
        return out[:used].copy()
