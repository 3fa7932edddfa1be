"""CPU tier: the permutation tables the product builds (csrc/fec_tables.cpp, own formulation of EN 302 755 6.1.3 / 6.4)
against the reference-shaped oracle restatement (oracle/fec_oracle.c), for every mode the reference instantiates."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol


@pytest.fixture(scope="module")
def l(built):
    import sdr_receiver_dvb_t2_amd as pkg
    return pkg.lib()


@pytest.mark.parametrize("mod", [1, 2, 3])
@pytest.mark.parametrize("fec_type,code_rate", [(0, 0), (0, 3), (1, 0), (1, 1), (1, 2), (1, 3), (1, 5)])
def test_bit_deinterleaver_address(l, mod, fec_type, code_rate):
    size = 64800 if fec_type == 1 else 16200
    got = np.zeros(size, np.uint16)
    assert l.t2gpu_table_bitdeint(mod, fec_type, code_rate, got.ctypes.data) == size
    want = ol.ora_bitdeint_address(mod, fec_type, code_rate)
    assert np.array_equal(got.astype(np.int32), want)
    assert np.array_equal(np.sort(want), np.arange(size))        # a permutation


@pytest.mark.parametrize("cells,blocks", [(8100, 7), (2025, 3), (2700, 5), (4050, 2), (10800, 2), (16200, 2), (32400, 1)])
def test_cell_deinterleaver_permutation(l, cells, blocks):
    got = np.zeros(cells * blocks, np.int32)
    assert l.t2gpu_table_cell_deint(blocks, cells, got.ctypes.data) == cells * blocks
    want = ol.ora_cell_perm(blocks, cells)
    assert np.array_equal(got, want)
    for r in range(blocks):                                        # each FEC block permuted within itself
        assert np.array_equal(np.sort(want[r * cells:(r + 1) * cells]), np.arange(r * cells, (r + 1) * cells))


def test_bb_scrambler_sequence(l):
    got = np.zeros(54000, np.uint8)
    assert l.t2gpu_table_bb_prbs(got.ctypes.data, 54000) == 54000
    assert np.array_equal(got, ol.ora_bb_prbs(54000))
    # 1 + x^14 + x^15 is maximal length: period 2^15 - 1, balanced up to one bit
    assert np.array_equal(got[:54000 - 32767], got[32767:54000])
    assert abs(int(got[:32767].sum()) - 16384) <= 1


def test_oracle_slicer_forms_agree():
    """oracle/fec_oracle.c carries the reference's decision trees (llr_demapper.cpp:257-276, 395-436, 567-654) literally and in a
    branch-free form that the cpu_baseline leg runs: the two agree on a dense sweep of every axis, at every threshold and its float
    neighbours, at 0 and at NaN -- incl. the 64-QAM negative side that never decides the outermost point."""
    import ctypes
    import oracle_lib as ol
    fn = ol.oracle().ora_slice_selfcheck
    fn.restype = ctypes.c_int
    assert fn() == 0
