"""Test-side bindings of the CPU checker: oracle/liboracle.so (our C restatement), oracle/_ref/libref_ldpc.so
(the reference's own LDPC headers compiled by oracle/Makefile, absent when never built) and the schedule emulator
tests/emu (host replay of the kernel's schedule). TEST INFRASTRUCTURE: nothing in the product imports this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_i8p = ctypes.POINTER(ctypes.c_int8)
_u8p = ctypes.POINTER(ctypes.c_uint8)

_cache = {}


def _make_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def oracle():
    if "ora" not in _cache:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            _make_oracle()
        _cache["ora"] = ctypes.CDLL(path)
    return _cache["ora"]


def ref():
    """The compiled reference LDPC (or None when oracle/_ref was never built on this machine)."""
    if "ref" not in _cache:
        path = os.path.join(ROOT, "oracle", "_ref", "libref_ldpc.so")
        try:
            _cache["ref"] = ctypes.CDLL(path) if os.path.exists(path) else None
        except OSError:
            _cache["ref"] = None
    return _cache["ref"]


def emu():
    if "emu" not in _cache:
        bdir = os.path.join(ROOT, "tests", "_build")
        os.makedirs(bdir, exist_ok=True)
        path = os.path.join(bdir, "libemu.so")
        srcs = [os.path.join(ROOT, "tests", "emu", "ldpc_emu.cpp"),
                os.path.join(ROOT, "sdr_receiver_dvb_t2_amd", "csrc", "ldpc_graph.cpp")]
        deps = srcs + [os.path.join(ROOT, "sdr_receiver_dvb_t2_amd", "csrc", "ldpc_cn.h")]
        if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", path] + srcs)
        _cache["emu"] = ctypes.CDLL(path)
    return _cache["emu"]


def code_id(fec_type, code_rate):
    return fec_type * 6 + code_rate


def ldpc_params(cid):
    n, k, q, lt = (ctypes.c_int() for _ in range(4))
    assert oracle().ora_ldpc_params(cid, ctypes.byref(n), ctypes.byref(k), ctypes.byref(q), ctypes.byref(lt)) == 0
    return n.value, k.value, q.value, lt.value


def ldpc_encode(cid, info):
    n, k, _, _ = ldpc_params(cid)
    info = np.ascontiguousarray(info, dtype=np.uint8).reshape(-1, k)
    cw = np.zeros((info.shape[0], n), dtype=np.uint8)
    for b in range(info.shape[0]):
        oracle().ora_ldpc_encode(cid, info[b].ctypes.data_as(_u8p), cw[b].ctypes.data_as(_u8p))
    return cw


def make_llr(cid, frames, sigma, seed, scale=8.0, info=None):
    """Random codewords (or the codewords of the given information bits) through a BPSK/AWGN channel, quantised to the int8 LLR
    format the demapper produces (positive = bit 0). Returns (info bits, int8 LLRs)."""
    n, k, _, _ = ldpc_params(cid)
    rng = np.random.Generator(np.random.PCG64(seed))
    rnd = rng.integers(0, 2, size=(frames, k), dtype=np.uint8)
    info = rnd if info is None else np.ascontiguousarray(info, dtype=np.uint8)
    cw = ldpc_encode(cid, info)
    y = (1.0 - 2.0 * cw) + sigma * rng.standard_normal(cw.shape)
    llr = np.clip(np.rint(y * scale), -127, 127).astype(np.int8)
    return info, llr


def _decode(fn, cid, llr, trials):
    n, k, _, _ = ldpc_params(cid)
    llr = np.ascontiguousarray(llr, dtype=np.int8).reshape(-1, n)
    blocks = llr.shape[0]
    bits = np.full((blocks, k), 255, dtype=np.uint8)
    lo = np.zeros((blocks, n), dtype=np.int8)
    r = fn(cid, llr.ctypes.data_as(_i8p), blocks, trials, bits.ctypes.data_as(_u8p), lo.ctypes.data_as(_i8p))
    return r, bits, lo


def ora_decode(cid, llr, trials=25):
    """One reference-style batch (all frames stop together). Returns (trials_left, bits[blocks][k], llr_out)."""
    return _decode(oracle().ora_ldpc_decode, cid, llr, trials)


def ref_decode(cid, llr, trials=25):
    return _decode(ref().ref_ldpc_decode, cid, llr, trials)


def emu_decode(cid, llr, trials=25, order_mode=0):
    """Host replay of the kernel schedule; order_mode picks the thread order inside every barrier epoch
    (0 ascending, 1 descending, >= 2 seeded shuffles)."""
    n, k, _, _ = ldpc_params(cid)
    llr = np.ascontiguousarray(llr, dtype=np.int8).reshape(-1, n)
    blocks = llr.shape[0]
    bits = np.full((blocks, k), 255, dtype=np.uint8)
    lo = np.zeros((blocks, n), dtype=np.int8)
    r = emu().emu_ldpc_decode(cid, llr.ctypes.data_as(_i8p), blocks, trials, bits.ctypes.data_as(_u8p),
                              lo.ctypes.data_as(_i8p), order_mode)
    return r, bits, lo


def ora_decode_batched(cid, llr, group=32, trials=25):
    """Decode [frames][n] in reference batches of `group`; returns (trials_left[batches], bits, llr_out)."""
    n, k, _, _ = ldpc_params(cid)
    llr = np.ascontiguousarray(llr, dtype=np.int8).reshape(-1, n)
    frames = llr.shape[0]
    tl, bits, lo = [], np.zeros((frames, k), np.uint8), np.zeros((frames, n), np.int8)
    for b0 in range(0, frames, group):
        r, b, l = ora_decode(cid, llr[b0:b0 + group], trials)
        tl.append(r)
        lo[b0:b0 + group] = l
        bits[b0:b0 + group] = (l[:, :k] < 0)
    return np.array(tl, np.int32), bits, lo


# ------------------------------------------------------------------------------------------------ FEC-side stages
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int)


def ora_bb_prbs(n):
    out = np.zeros(n, np.uint8)
    oracle().ora_bb_prbs(out.ctypes.data_as(_u8p), n)
    return out


def ora_bch_descramble(cid, bits):
    n, k, _, _ = ldpc_params(cid)
    bits = np.ascontiguousarray(bits, np.uint8).reshape(-1, k)
    kb = [7032, 9552, 10632, 11712, 12432, 13152, 32208, 38688, 43040, 48408, 51648, 53840][cid]
    out = np.zeros((bits.shape[0], kb), np.uint8)
    assert oracle().ora_bch_descramble(cid, bits.ctypes.data_as(_u8p), bits.shape[0], out.ctypes.data_as(_u8p)) == kb
    return out


def bch_params(cid):
    """(m, t, k_bch, n_bch) of LDPC code cid: parity bits = n_bch - k_bch = m * t (bch_decoder.cpp:79-134 holds the pairs)."""
    _, k, _, _ = ldpc_params(cid)
    kb = [7032, 9552, 10632, 11712, 12432, 13152, 32208, 38688, 43040, 48408, 51648, 53840][cid]
    m = 14 if cid < 6 else 16
    return m, (k - kb) // m, kb, k


def ora_bch_minpoly(m, j):
    o = oracle()
    o.ora_bch_minpoly.restype = ctypes.c_uint32
    return int(o.ora_bch_minpoly(m, j))


def ora_bch_generator(m, t):
    g = np.zeros(m * t + 1, np.uint8)
    deg = oracle().ora_bch_generator(m, t, g.ctypes.data_as(_u8p))
    return g[:deg + 1]


def ora_bch_encode(cid, msg):
    """[f][k_bch] -> [f][n_bch] systematic codewords (bitwise LFSR)."""
    m, t, kb, nb = bch_params(cid)
    out = np.zeros((msg.shape[0], nb), np.uint8)
    out[:, :kb] = msg
    for row in out:
        assert oracle().ora_bch_encode(m, t, row.ctypes.data_as(_u8p), kb, nb) == 0
    return out


def ora_bch_decode(cid, words):
    """[f][n_bch] received words -> (corrected copy, status per frame: corrected bits, or -1 = more than t errors)."""
    m, t, _, nb = bch_params(cid)
    out = np.ascontiguousarray(words, dtype=np.uint8).copy()
    st = np.zeros(out.shape[0], np.int32)
    for i, row in enumerate(out):
        st[i] = oracle().ora_bch_decode(m, t, row.ctypes.data_as(_u8p), nb)
    return out, st


def ora_bitdeint_address(mod, fec_type, code_rate):
    size = 64800 if fec_type == 1 else 16200
    a = np.zeros(size, np.int32)
    assert oracle().ora_bitdeint_address(mod, fec_type, code_rate, a.ctypes.data_as(_ip)) == size
    return a


def ora_demap(mod, fec_type, code_rate, rotation, cells, precision_override=0.0):
    """cells complex64 [n]; returns (llr [frames][fec_size], sums[3], derotated cells)."""
    c = np.ascontiguousarray(cells, np.complex64).copy()
    size = 64800 if fec_type == 1 else 16200
    cpf = size // (2 * (mod + 1))
    frames = c.size // cpf
    out = np.zeros((frames, size), np.int8)
    sums = np.zeros(3, np.float32)
    fn = oracle().ora_demap
    fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float]
    r = fn(mod, fec_type, code_rate, rotation, c.ctypes.data, c.size, out.ctypes.data, sums.ctypes.data, precision_override)
    assert r == frames
    return out, sums, c


def ora_cell_perm(num_blocks, cells_per_fec):
    p = np.zeros(num_blocks * cells_per_fec, np.int32)
    oracle().ora_cell_perm(num_blocks, cells_per_fec, p.ctypes.data_as(_ip))
    return p


def ora_ti_frame_walk(plps, dyns, num_cells, plp_state=0, max_out=4096):
    """plps: [(mod, fec_type, time_il_length, time_il_type)], dyns: [(id, start, num_blocks)] -> ([(plp, first, blocks, size)], state)"""
    o = oracle()
    P = np.ascontiguousarray(np.array(plps, np.int32).reshape(len(plps), 4))
    D = np.ascontiguousarray(np.array(dyns, np.int32).reshape(len(plps), 3))
    out = np.zeros((max_out, 4), np.int32)
    st = ctypes.c_int(plp_state)
    o.ora_ti_frame_walk.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_int]
    rc = o.ora_ti_frame_walk(len(plps), P.ctypes.data, D.ctypes.data, num_cells, ctypes.byref(st), out.ctypes.data, max_out)
    if rc < 0:
        return rc, st.value
    return [tuple(int(v) for v in out[k]) for k in range(rc)], st.value


class OraTi(object):
    def __init__(self, cells_per_fec, num_blocks_max):
        o = oracle()
        o.ora_ti_create.restype = ctypes.c_void_p
        o.ora_ti_begin.argtypes = [ctypes.c_void_p, ctypes.c_int]
        o.ora_ti_push.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        o.ora_ti_destroy.argtypes = [ctypes.c_void_p]
        self._o, self._h = o, o.ora_ti_create(cells_per_fec, num_blocks_max)

    def begin(self, num_blocks):
        self._o.ora_ti_begin(self._h, num_blocks)

    def push(self, cells, out):
        cells = np.ascontiguousarray(cells, np.complex64)
        return self._o.ora_ti_push(self._h, cells.ctypes.data, cells.size, out.ctypes.data)

    def __del__(self):
        self._o.ora_ti_destroy(self._h)


# ------------------------------------------------------------------------------------------------ OFDM side
class OraMode(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("fft_mode", "carrier_mode", "pilot_pattern", "guard_interval_mode", "papr_mode", "n_data",
                                            "is32k", "fft_size", "k_total", "k_ext", "k_offset", "l_nulls", "n_p2", "c_p2", "c_data",
                                            "n_fc", "c_fc", "l_fc", "len_frame", "dx", "dy")] + \
               [(n, ctypes.c_float) for n in ("amp_sp", "amp_cp", "amp_p2")]


def ora_mode(fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data):
    m = OraMode(fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data)
    rc = oracle().ora_mode_init(ctypes.byref(m))
    return m if rc == 0 else None


def ora_symbol_carriers(m, idx_symbol):
    mp = np.zeros(m.k_total, np.int32)
    rf = np.zeros(m.k_total, np.float32)
    oracle().ora_symbol_carriers(ctypes.byref(m), idx_symbol, mp.ctypes.data_as(_ip), rf.ctypes.data_as(_fp))
    return mp, rf


def ora_freq_deint(m, kind):
    cells = [m.c_p2, m.c_data, m.n_fc][kind]
    he, ho = np.zeros(cells, np.int32), np.zeros(cells, np.int32)
    assert oracle().ora_freq_deint(ctypes.byref(m), kind, he.ctypes.data_as(_ip), ho.ctypes.data_as(_ip)) == cells
    return he, ho


def ora_data_symbol(m, idx_symbol, ofdm_cell):
    """data_symbol::execute on one fft-shifted symbol (complex64[fft_size]) -> (cells complex64[c_data], phase_offset, sro)."""
    mp, rf = ora_symbol_carriers(m, idx_symbol)
    kind = 0 if idx_symbol < m.n_p2 else (2 if (m.l_fc and idx_symbol == m.len_frame - 1) else 1)
    he, ho = ora_freq_deint(m, kind)
    h = ho if idx_symbol % 2 == 0 else he
    x = np.ascontiguousarray(ofdm_cell, np.complex64)
    ncell = [m.c_p2, m.c_data, m.n_fc][kind]
    out = np.zeros(ncell, np.complex64)
    sync = np.zeros(2, np.float32)
    fn = oracle().ora_data_symbol
    fn.argtypes = [ctypes.c_void_p] * 7
    n = fn(ctypes.addressof(m), x.ctypes.data, mp.ctypes.data, rf.ctypes.data, h.ctypes.data, out.ctypes.data, sync.ctypes.data)
    assert n == ncell, (n, ncell)
    return out, float(sync[0]), float(sync[1])


# ---- sample-rate front end (oracle/front_oracle.c) and the reference's Qt-free DSP classes (oracle/_ref/libref_dsp.so) ----
def ref_dsp(strict=False):
    """The reference's Qt-free DSP headers compiled unmodified: with the reference's own flags (-Ofast), or strict=True
    without fast-math so every float operation happens in the order the reference source writes it."""
    key = "ref_dsp_strict" if strict else "ref_dsp"
    if key not in _cache:
        path = os.path.join(ROOT, "oracle", "_ref", "lib%s.so" % key)
        try:
            _cache[key] = ctypes.CDLL(path) if os.path.exists(path) else None
        except OSError:
            _cache[key] = None
    return _cache[key]


def _c64(x):
    return np.ascontiguousarray(x, np.complex64)


class OraDecim(object):
    """filter_decimator restated; ref=True drives the reference class itself."""

    def __init__(self, ref=False, strict=False):
        self.lib = ref_dsp(strict) if ref else oracle()
        self.new, self.free, self.run = ((self.lib.ref_decim_new, self.lib.ref_decim_free, self.lib.ref_decim_execute) if ref else
                                         (self.lib.ora_decim_create, self.lib.ora_decim_destroy, self.lib.ora_decim_execute))
        self.new.restype = ctypes.c_void_p
        self.free.argtypes = [ctypes.c_void_p]
        self.run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.h = self.new()

    def __call__(self, x):
        x = _c64(x)
        out = np.zeros(len(x) // 2 + 1, np.complex64)
        n = self.run(self.h, len(x), x.ctypes.data, out.ctypes.data)
        return out[:n].copy()

    def __del__(self):
        if getattr(self, "h", None):
            self.free(self.h)
            self.h = None


class OraFarrow(object):
    """interpolator_farrow<complex,float> restated; ref=True drives the reference template itself."""

    def __init__(self, ref=False, strict=False):
        self.ref = ref
        self.lib = ref_dsp(strict) if ref else oracle()
        self.new, self.free, self.run = ((self.lib.ref_farrow_new, self.lib.ref_farrow_free, self.lib.ref_farrow_execute) if ref else
                                         (self.lib.ora_farrow_create, self.lib.ora_farrow_destroy, self.lib.ora_farrow_execute))
        self.new.restype = ctypes.c_void_p
        self.free.argtypes = [ctypes.c_void_p]
        self.run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p] + ([] if ref else [ctypes.c_void_p])
        self.h = self.new()

    def __call__(self, x, resample, want_phases=False):
        x = _c64(x)
        cap = int(len(x) / max(resample, 0.05)) + 8
        out = np.zeros(cap, np.complex64)
        if self.ref:
            n = self.run(self.h, len(x), x.ctypes.data, resample, out.ctypes.data)
            return out[:n].copy()
        ph = np.zeros(cap, np.float32)
        n = self.run(self.h, len(x), x.ctypes.data, resample, out.ctypes.data, ph.ctypes.data)
        return (out[:n].copy(), ph[:n].copy()) if want_phases else out[:n].copy()

    def phase(self):
        """x1 after the last call (restated class only)."""
        fn = self.lib.ora_farrow_phase
        fn.restype = ctypes.c_float
        fn.argtypes = [ctypes.c_void_p]
        return np.float32(fn(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            self.free(self.h)
            self.h = None


class OraPi(ctypes.Structure):
    _fields_ = [("k_p", ctypes.c_float), ("k_i", ctypes.c_float), ("old_integral", ctypes.c_float)]


class OraFront(object):
    """dvbt2_demodulator::execute front loop (dc / iq imbalance / NCO), chunk by chunk."""

    def __init__(self, id_device=0):
        o = oracle()
        o.ora_front_create.restype = ctypes.c_void_p
        o.ora_front_destroy.argtypes = [ctypes.c_void_p]
        o.ora_front_chunk.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
        o.ora_front_finish.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        o.ora_front_get_state.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.o = o
        self.h = o.ora_front_create(id_device)
        self.stride = 2 if id_device == 1 else 1

    def execute(self, i_in, q_in, chunk_len, phase_est_filtered, frequency_est_filtered):
        """One execute() call: chunks in order with their loop values. Returns the derotated stream."""
        i_in = np.ascontiguousarray(i_in, np.int16)
        q_in = np.ascontiguousarray(q_in, np.int16)
        total = int(np.sum(chunk_len))
        out = np.zeros(total, np.complex64)
        theta = np.zeros(3, np.float32)
        pos = 0
        for n, pe, fe in zip(chunk_len, phase_est_filtered, frequency_est_filtered):
            self.o.ora_front_chunk(self.h, int(n), i_in.ctypes.data, q_in.ctypes.data, pos, float(pe), float(fe),
                                   theta.ctypes.data, out[pos:].ctypes.data)
            pos += int(n)
        self.o.ora_front_finish(self.h, total, theta.ctypes.data)
        return out, theta

    def set_iq(self, c1, c2):
        self.o.ora_front_set_iq.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
        self.o.ora_front_set_iq(self.h, float(c1), float(c2))

    def state(self):
        v = np.zeros(8, np.float32)
        self.o.ora_front_get_state(self.h, v.ctypes.data)
        return dict(dc_re=v[0], dc_im=v[1], c1=v[2], c2=v[3], phase_nco=v[4], frequency_nco=v[5], level_detect=v[6])

    def __del__(self):
        if getattr(self, "h", None):
            self.o.ora_front_destroy(self.h)
            self.h = None


def ora_cp_frequency_est(sym, fft_size, guard):
    sym = _c64(sym)
    s2 = np.zeros(2, np.float32)
    fn = oracle().ora_cp_frequency_est
    fn.restype = ctypes.c_float
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return float(fn(sym.ctypes.data, fft_size, guard, s2.ctypes.data)), complex(s2[0], s2[1])


class OraSync(object):
    """Tracking updates of symbol_acquisition (PI loop filters + the bang-bang sample-rate tracker)."""

    def __init__(self, sample_rate):
        o = oracle()
        o.ora_sync_create.restype = ctypes.c_void_p
        o.ora_sync_create.argtypes = [ctypes.c_float]
        o.ora_sync_destroy.argtypes = [ctypes.c_void_p]
        o.ora_sync_frequency.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_int]
        o.ora_sync_symbol.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
        o.ora_sync_resample.restype = ctypes.c_double
        o.ora_sync_resample.argtypes = [ctypes.c_void_p]
        o.ora_sync_get.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.o = o
        self.h = o.ora_sync_create(sample_rate)

    def frequency(self, frequency_est, fft_size):
        self.o.ora_sync_frequency(self.h, frequency_est, fft_size)

    def symbol(self, phase_est, sample_rate_est):
        self.o.ora_sync_symbol(self.h, phase_est, sample_rate_est)

    def resample(self):
        return self.o.ora_sync_resample(self.h)

    def get(self):
        v = np.zeros(4, np.float64)
        self.o.ora_sync_get(self.h, v.ctypes.data)
        return dict(phase_est_filtered=np.float32(v[0]), frequency_est_filtered=np.float32(v[1]),
                    sample_rate_est_filtered=v[2], resample=v[3])

    def __del__(self):
        if getattr(self, "h", None):
            self.o.ora_sync_destroy(self.h)
            self.h = None


class OraP1(object):
    """p1_symbol restated (oracle/p1_oracle.c)."""

    def __init__(self):
        o = oracle()
        o.ora_p1_create.restype = ctypes.c_void_p
        o.ora_p1_destroy.argtypes = [ctypes.c_void_p]
        o.ora_p1_execute.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.o = o
        self.h = o.ora_p1_create()
        self.buffer_sym = np.zeros(4096, np.complex64)

    def execute(self, x, consume=0, gain_changed=False, level_detect=0.0, reset=False, want_trace=False):
        x = _c64(x)
        c = ctypes.c_int(consume)
        res = (ctypes.c_int * 9)()
        coarse = ctypes.c_double()
        tr = np.full(len(x), np.nan, np.float32)
        fft = np.zeros(1024, np.complex64)
        d = self.o.ora_p1_execute(self.h, int(gain_changed), float(level_detect), len(x), x.ctypes.data, ctypes.byref(c),
                                  self.buffer_sym.ctypes.data, res, ctypes.byref(coarse), int(reset), tr.ctypes.data, fft.ctypes.data)
        out = dict(detected=bool(d), consume=c.value, idx_buffer_sym=res[0], p1_decoded=res[1], preamble=res[2], fft_mode=res[3],
                   s1=res[4], s2=res[5], shift=res[6], coarse_freq_offset=coarse.value)
        if want_trace:
            out["trace"], out["p1_fft"] = tr, fft
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.o.ora_p1_destroy(self.h)
            self.h = None
