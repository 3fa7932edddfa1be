"""Frame de-multiplexer of the time de-interleaver for several PLPs (host logic of the C ABI, t2gpu_ti_frame_plan) against the
oracle's cell-by-cell walk of time_deinterleaver::execute (/root/reference/src/DVB_T2/time_deinterleaver.cpp:288-376)."""
import numpy as np
import pytest

import oracle_lib as ol


def cpf_of(mod, fec_type):
    return (64800 if fec_type else 16200) // (2 * (mod + 1))


def both(plps, dyns, frame_cells, state=0):
    """plps: [(mod, fec_type, time_il_length, time_il_type, num_blocks_max)], dyns: [(id, start, num_blocks)]"""
    from sdr_receiver_dvb_t2_amd import fec
    P = [dict(plp_mod=p[0], plp_fec_type=p[1], time_il_length=p[2], time_il_type=p[3], plp_num_blocks_max=p[4]) for p in plps]
    D = [dict(id=d[0], start=d[1], num_blocks=d[2]) for d in dyns]
    got = fec.ti_frame_plan(P, D, frame_cells, plp_state=state)
    want = ol.ora_ti_frame_walk([p[:4] for p in plps], dyns, frame_cells, plp_state=state)
    return got, want


def test_single_plp_three_ti_blocks(built):
    cpf = cpf_of(2, 0)
    got, want = both([(2, 0, 3, 0, 200)], [(0, 0, 20)], 20 * cpf + 1000)
    assert got == want
    blocks, state = got
    assert [b[2] for b in blocks] == [6, 7, 7] and [b[0] for b in blocks] == [0, 0, 0]
    assert [b[1] for b in blocks] == [0, 6 * cpf, 13 * cpf] and state == 0


def test_two_plps_back_to_back(built):
    cpf = cpf_of(1, 0)
    plps = [(1, 0, 1, 0, 100), (1, 0, 2, 0, 100)]
    dyns = [(0, 0, 7), (1, 7 * cpf, 9)]
    got, want = both(plps, dyns, 16 * cpf + 500)
    assert got == want
    blocks, state = got
    assert blocks == [(0, 0, 7, 7 * cpf), (1, 7 * cpf, 4, 4 * cpf), (1, 11 * cpf, 5, 5 * cpf)]
    assert state == 1
    # next frame: starts in PLP 0 again because its PLP_START is 0
    got2, want2 = both(plps, dyns, 16 * cpf + 500, state=state)
    assert got2 == want2 == got


def test_second_plp_first_in_frame_and_id_as_index(built):
    """the frame starts with whichever PLP has PLP_START 0; the successor is found through PLP_START - 1 == last cell"""
    cpf = cpf_of(0, 0)
    plps = [(0, 0, 1, 0, 50), (0, 0, 1, 0, 50), (0, 0, 1, 0, 50)]
    dyns = [(0, 5 * cpf, 3), (1, 0, 5), (2, 8 * cpf, 2)]
    got, want = both(plps, dyns, 10 * cpf)
    assert got == want
    assert [b[0] for b in got[0]] == [1, 0, 2]
    assert [b[1] for b in got[0]] == [0, 5 * cpf, 8 * cpf]


def test_slice_end_uses_plp1_cell_count(built):
    """time_deinterleaver.cpp:273-274: the end of every PLP is computed with cells_per_fec_block[1]. PLP 0 is 16-QAM, PLP 1 is
    64-QAM: PLP 0's end is missed, no switch happens and PLP 0's geometry is applied to PLP 1's cells (the reference does
    exactly that; the product reproduces it rather than guessing what was meant)."""
    c0, c1 = cpf_of(1, 0), cpf_of(2, 0)
    plps = [(1, 0, 1, 0, 50), (2, 0, 1, 0, 50)]
    dyns = [(0, 0, 4), (1, 4 * c0, 6)]
    got, want = both(plps, dyns, 4 * c0 + 6 * c1 + 100)
    assert got == want
    assert all(b[0] == 0 for b in got[0]) and len(got[0]) == (4 * c0 + 6 * c1 + 100) // (4 * c0)


def test_no_switch_keeps_last_block_size(built):
    """when no successor is found the same PLP continues and its next TI block keeps the size of the one just finished
    (:361-371 recompute the size only on a switch or inside an interleaving frame)"""
    cpf = cpf_of(3, 0)
    plps = [(3, 0, 2, 0, 50)]
    dyns = [(0, 0, 5)]                       # TI blocks of 2 and 3 FEC blocks
    got, want = both(plps, dyns, 40 * cpf)
    assert got == want
    assert [b[2] for b in got[0]][:5] == [2, 3, 3, 3, 3]


def test_ti_type_1_is_refused(built):
    from sdr_receiver_dvb_t2_amd import fec
    from sdr_receiver_dvb_t2_amd._lib import T2GpuError
    with pytest.raises(T2GpuError):
        fec.ti_frame_plan([dict(plp_mod=0, plp_fec_type=0, time_il_length=2, time_il_type=1, plp_num_blocks_max=10)],
                          [dict(id=0, start=0, num_blocks=4)], 100000)
    assert ol.ora_ti_frame_walk([(0, 0, 2, 1)], [(0, 0, 4)], 100000)[0] == -2


def test_random_configurations(built):
    rng = np.random.Generator(np.random.PCG64(77))
    for trial in range(300):
        n = int(rng.integers(1, 5))
        same = rng.random() < 0.7
        mod0, fec0 = int(rng.integers(0, 4)), int(rng.integers(0, 2))
        plps, dyns, pos = [], [], 0
        order = rng.permutation(n)
        starts = {}
        for k in order:                      # slices back to back in a random PLP order, sometimes with a gap
            mod, fec_type = (mod0, fec0) if same else (int(rng.integers(0, 4)), int(rng.integers(0, 2)))
            nb = int(rng.integers(1, 12))
            starts[int(k)] = (pos, nb, mod, fec_type)
            pos += nb * cpf_of(mod, fec_type) + (int(rng.integers(1, 50)) if rng.random() < 0.15 else 0)
        for i in range(n):
            s, nb, mod, fec_type = starts[i]
            plps.append((mod, fec_type, int(rng.integers(1, 4)), 0, 64))
            dyns.append((i, s, nb))
        frame_cells = pos + int(rng.integers(0, 3000))
        if rng.random() < 0.2:
            frame_cells = max(0, pos - int(rng.integers(1, 4000)))          # last PLP cut by the frame end
        got, want = both(plps, dyns, frame_cells, state=int(rng.integers(0, n)))
        assert got == want, (trial, plps, dyns, frame_cells)
