"""CPU tier: pin the LDPC oracle (oracle/ldpc_oracle.c) and the kernel's schedule (tests/emu replaying
csrc/ldpc_graph.cpp + csrc/ldpc_cn.h) against the reference.

Pins: tests/golden/ldpc_golden.npz was produced by the reference's own LDPC/*.hh (tests/golden/make_ldpc_golden.py);
when oracle/_ref/libref_ldpc.so exists on this machine the comparison is also made live against it."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as ol

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ldpc_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})


def _check_against_golden(name, decode):
    cid = int(GOLD[name + "/cid"])
    llr = GOLD[name + "/llr"]
    n, k, _, _ = ol.ldpc_params(cid)
    res = decode(cid, llr)
    t, bits, lo = res[0], res[1], res[2]
    assert t == int(GOLD[name + "/trials_left"])
    assert hashlib.sha256(lo.tobytes()).digest() == GOLD[name + "/llr_sha256"].tobytes()
    hard = (lo[:, :k] < 0).astype(np.uint8)
    assert np.array_equal(np.packbits(hard, axis=1), GOLD[name + "/hard"])
    if t >= 0:
        assert np.array_equal(bits, hard)
    return res


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    _check_against_golden(name, ol.ora_decode)


@pytest.mark.parametrize("order_mode", [0, 1, 2, 3])
@pytest.mark.parametrize("name", CASES)
def test_kernel_schedule_matches_reference_golden(name, order_mode):
    """The kernel's phase schedule (PLAIN / PAIR chain walk / GENERIC levels) + compressed check-node records give
    the reference's LLRs whatever order the threads run in between two barriers."""
    _check_against_golden(name, lambda cid, llr: ol.emu_decode(cid, llr, order_mode=order_mode))


@pytest.mark.skipif(ol.ref() is None, reason="oracle/_ref not built here (reference tree absent)")
@pytest.mark.parametrize("cid,sigma,frames", [(9, 0.61, 32), (0, 0.93, 32), (8, 0.66, 8), (6, 0.88, 4)])
def test_oracle_matches_reference_live(cid, sigma, frames):
    info, llr = ol.make_llr(cid, frames, sigma, seed=77 + cid)
    t1, b1, l1 = ol.ora_decode(cid, llr)
    t2, b2, l2 = ol.ref_decode(cid, llr)
    assert t1 == t2
    assert np.array_equal(l1, l2)
    if t1 >= 0:
        assert np.array_equal(b1, b2)


def test_encoder_produces_codewords():
    """Noise-free codewords satisfy every check immediately (trials-left == max) for all twelve codes."""
    for cid in range(12):
        n, k, _, _ = ol.ldpc_params(cid)
        rng = np.random.Generator(np.random.PCG64(cid))
        info = rng.integers(0, 2, size=(2, k), dtype=np.uint8)
        cw = ol.ldpc_encode(cid, info)
        llr = (40 * (1 - 2 * cw.astype(np.int32))).astype(np.int8)
        t, bits, _ = ol.ora_decode(cid, llr)
        assert t == 25 and np.array_equal(bits, info)


def test_zero_llr_is_bad():
    """A zero LLR fails the parity check even when all signs agree (layered_decoder.hh:65-82 via vsign)."""
    cid = 0
    n, k, _, _ = ol.ldpc_params(cid)
    cw = ol.ldpc_encode(cid, np.zeros((1, k), np.uint8))
    llr = (40 * (1 - 2 * cw.astype(np.int32))).astype(np.int8)
    llr[0, 5] = 0
    t, bits, lo = ol.ora_decode(cid, llr)
    assert t == 24 and lo[0, 5] > 0


@pytest.mark.parametrize("code_id", range(12))
def test_band_walk_description_of_generic_layers(code_id):
    """ldpc_graph.cpp marks GENERIC layers the two-frame kernel may walk in bands (band = D, slots 0 / 1 = the pair handed down in a
    register, band_prefetch). tests/emu checks the three properties the walk relies on by brute force over nodes and bits; N 3/4 (id 9)
    must have its layers 5 (D = 31) and 42 (D = 11) among them -- they are a fifth of a sweep otherwise."""
    import ctypes
    emu = ol.emu()
    n = ctypes.c_int(0)
    assert emu.emu_ldpc_band_check(code_id, ctypes.byref(n)) == 0
    if code_id == 9:
        assert n.value == 2


@pytest.mark.parametrize("code_id", range(12))
def test_layers_left_open_share_nothing(code_id):
    """ldpc_graph.cpp marks layers the two-frame kernel does not close with a workgroup barrier (no_close). tests/emu re-derives the
    condition by brute force: whatever runs without a barrier in between is PLAIN / PAIR, shares no information-bit group and holds
    at most one PAIR layer. N 3/4 must have some (11 of its 44 inner transitions are eligible)."""
    import ctypes
    emu = ol.emu()
    n = ctypes.c_int(0)
    assert emu.emu_ldpc_open_check(code_id, ctypes.byref(n)) == 0
    if code_id == 9:
        assert n.value >= 8
