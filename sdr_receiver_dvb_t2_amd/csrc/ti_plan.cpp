// ti_plan.cpp -- which TI blocks a T2 frame's cell stream holds, for any number of PLPs: the frame de-multiplexer part of
// time_deinterleaver (/root/reference/src/DVB_T2/time_deinterleaver.cpp:38-145 start, :268-286 l1_dyn_execute, :288-376
// execute) in closed form. The reference walks the stream cell by cell with a small state machine (current PLP, current TI
// block, cell counter); nothing in it depends on cell values, so the sequence of (PLP, first cell, size) it emits ti_block
// for is a function of the L1-post signalling alone. Host code, as in the reference; the cells themselves are moved by
// ti_scatter_kernel, one TI block per call.
#include "../../include/t2gpu.h"
#include "t2gpu_common.h"

#include <cmath>
#include <vector>

using t2gpu::set_error;

namespace {

int cells_per_fec_block(const t2gpu_l1_plp &p)
{
    if (p.plp_mod < 0 || p.plp_mod > 3 || p.plp_fec_type < 0 || p.plp_fec_type > 1) return -1;
    return (p.plp_fec_type == 1 ? 64800 : 16200) / (2 * (p.plp_mod + 1));        // time_deinterleaver.cpp:63-118
}

}  // namespace

extern "C" int t2gpu_ti_frame_plan(int num_plp, const t2gpu_l1_plp *plp, const t2gpu_l1_dyn_plp *dyn, int frame_cells,
                                   int *plp_state, t2gpu_ti_block *out, int max_out)
{
    if (num_plp < 1 || num_plp > 255 || !plp || !dyn || frame_cells < 0 || !plp_state || !out || max_out < 0) {
        set_error("t2gpu_ti_frame_plan: bad arguments");
        return -1;
    }
    std::vector<int> cpf(num_plp), n_ti(num_plp);
    for (int i = 0; i < num_plp; ++i) {
        cpf[i] = cells_per_fec_block(plp[i]);
        if (cpf[i] < 0) { set_error("t2gpu_ti_frame_plan: PLP modulation / FEC type out of range"); return -1; }
        if (plp[i].time_il_type != 0) {
            // TIME_IL_TYPE 1 (one TI block over several T2 frames): the reference sizes its per-PLP arrays for one TI block
            // and then fills time_il_length entries (:122-129 vs :277-284) and never gathers cells over frames
            set_error("t2gpu_ti_frame_plan: TIME_IL_TYPE 1 is not decodable by the reference stage either");
            return -2;
        }
        n_ti[i] = plp[i].time_il_length;
        if (n_ti[i] < 1 || dyn[i].num_blocks < 0 || dyn[i].num_blocks > plp[i].plp_num_blocks_max) {
            set_error("t2gpu_ti_frame_plan: TIME_IL_LENGTH < 1 or PLP_NUM_BLOCKS above PLP_NUM_BLOCKS_MAX");
            return -1;
        }
    }
    // FEC blocks of TI block j of PLP i (:275-284): the later blocks take the remainder
    auto blocks_of = [&](int i, int j) {
        const int nb = dyn[i].num_blocks;
        int f = (int)std::floor((float)nb / (float)n_ti[i]);
        if (j >= n_ti[i] - nb % n_ti[i]) f += 1;
        return f;
    };
    // where the reference believes PLP i ends: with the cell count of PLP *1* whatever i is (:273-274); a single PLP has no
    // successor to switch to, so the out-of-range read it does there has no consequence
    auto slice_end = [&](int i) { return dyn[i].start + dyn[i].num_blocks * cpf[num_plp > 1 ? 1 : 0] - 1; };

    int cur = *plp_state;
    for (int i = 0; i < num_plp; ++i)
        if (dyn[i].start == 0) cur = i;                                            // :301-303 (the last match stays)
    if (cur < 0 || cur >= num_plp) { set_error("t2gpu_ti_frame_plan: no PLP starts the frame"); return -1; }
    int n = 0, pos = 0, j = 0;
    int size = blocks_of(cur, 0) * cpf[cur];                                       // :308
    while (size > 0 && pos + size <= frame_cells) {                                // an empty block never completes (:337)
        if (n >= max_out) { set_error("t2gpu_ti_frame_plan: more TI blocks than the caller's array holds"); return -3; }
        out[n++] = t2gpu_ti_block{cur, pos, size / cpf[cur], size};
        pos += size;
        if (++j == n_ti[cur]) {                                                    // :354-368
            j = 0;
            const int idx_cell = pos - 1;
            if (idx_cell == slice_end(cur)) {
                for (int i = 0; i < num_plp; ++i)
                    if (idx_cell == dyn[i].start - 1) {
                        if (dyn[i].id < 0 || dyn[i].id >= num_plp) {
                            set_error("t2gpu_ti_frame_plan: PLP_ID used as an index is out of range (:360)");
                            return -1;
                        }
                        cur = dyn[i].id;
                        size = blocks_of(cur, 0) * cpf[cur];
                    }
            }
            // no successor found: the same PLP goes on and its next TI block keeps the size of the one just finished --
            // the reference recomputes the size only on a switch or inside an interleaving frame (:361-371)
        } else {
            size = blocks_of(cur, j) * cpf[cur];
        }
    }
    *plp_state = cur;
    return n;
}
