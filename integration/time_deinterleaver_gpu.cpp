// integration/time_deinterleaver_gpu.cpp -- the bodies of
//     void time_deinterleaver::l1_dyn_execute(l1_postsignalling _l1_post, int _len_in, complex* _ofdm_cell)
//     void time_deinterleaver::execute(int _len_in, complex* _ofdm_cell)
//     (/root/reference/src/DVB_T2/time_deinterleaver.h:44-45; the reference's bodies: time_deinterleaver.cpp:268-376)
// start() stays the reference's (it sizes buffer_a / buffer_b and sets p2_start_idx_cell, :38-145). The cell-by-cell state machine of
// execute -- which TI block of which PLP a cell belongs to, where the frame's blocks end -- becomes one call per T2 frame
// (t2gpu_ti_frame_plan: the blocks of the frame in the order the reference emits ti_block for them, its quirks included, include/t2gpu.h);
// the cells of a symbol are then pushed to the PLP's de-interleaver on the GPU (cell permutation + the cyclic Q delay), and a complete TI
// block comes back in the stage's A / B buffer and goes to llr_demapper through the same signal with the same hand-shake (:335-352).
#include <algorithm>

#include "time_deinterleaver.h"    // the reference's
#include "t2gpu_ref_glue.h"

namespace {
struct lane { t2gpu_ti *h = nullptr; int num_blocks_max = 0, mod = -1, fec = -1; };
std::vector<lane> lanes;                                  // one de-interleaver per PLP, made at first use
std::vector<t2gpu_l1_plp> plps;
std::vector<t2gpu_l1_dyn_plp> dyns;
std::vector<t2gpu_ti_block> plan;                         // the frame's TI blocks (plp, offset, num_blocks, size)
size_t block = 0;                                         // the block being filled
int pos = 0;                                              // cells of the frame (behind the L1 cells) taken so far
int plp_state = 0;                                        // the PLP the previous frame ended in (the reference's plp_id member role)
}

void time_deinterleaver::l1_dyn_execute(l1_postsignalling _l1_post, int _len_in, complex* _ofdm_cell)
{
    l1_post = _l1_post;
    t2glue::plps_of(l1_post, plps, dyns);
    lanes.resize(plps.size());
    for (size_t i = 0; i < plps.size(); ++i) {
        lane &l = lanes[i];
        if (l.h && l.mod == plps[i].plp_mod && l.fec == plps[i].plp_fec_type && l.num_blocks_max >= plps[i].plp_num_blocks_max) continue;
        if (l.h) t2gpu_ti_destroy(l.h);
        l.h = t2gpu_ti_create(plps[i].plp_mod, plps[i].plp_fec_type, plps[i].plp_num_blocks_max, /*device*/0);
        l.mod = plps[i].plp_mod; l.fec = plps[i].plp_fec_type; l.num_blocks_max = plps[i].plp_num_blocks_max;
        if (!l.h) t2glue::complain("t2gpu_ti_create");
    }
    plan.resize(4096);
    const int n = t2gpu_ti_frame_plan((int)plps.size(), plps.data(), dyns.data(), 1 << 22, &plp_state, plan.data(), (int)plan.size());
    if (n < 0) { t2glue::complain("t2gpu_ti_frame_plan"); plan.clear(); }
    else plan.resize((size_t)n);
    block = 0;
    pos = 0;
    start_t2_frame = true;
    execute(_len_in, _ofdm_cell);
}

void time_deinterleaver::execute(int _len_in, complex* _ofdm_cell)
{
    mutex_in->lock();
    signal_in->wakeOne();
    int n = _len_in;
    complex *cells = _ofdm_cell;
    if (start_t2_frame) {                                 // the P2 symbol: the PLP cells start behind the L1 cells (:296-300)
        start_t2_frame = false;
        n -= p2_start_idx_cell;
        cells += p2_start_idx_cell;
    }
    while (n > 0 && block < plan.size()) {
        const t2gpu_ti_block &b = plan[block];
        t2gpu_ti *h = lanes[(size_t)b.plp].h;
        if (!h) break;
        if (pos == b.offset && t2gpu_ti_begin(h, b.num_blocks) != 0) { t2glue::complain("t2gpu_ti_begin"); break; }
        const int take = std::min(n, b.offset + b.size - pos);
        complex *out = swap_buffers ? buffer_a : buffer_b;
        const int done = t2gpu_ti_push(h, reinterpret_cast<const float *>(cells), take, reinterpret_cast<float *>(out));
        if (done < 0) { t2glue::complain("t2gpu_ti_push"); break; }
        cells += take; n -= take; pos += take;
        if (done == 1) {                                  // the TI block is complete and lies in `out`
            swap_buffers = !swap_buffers;
            ++block;
            mutex_out->lock();
            emit ti_block(b.size, out, b.plp, l1_post);
            signal_out->wait(mutex_out);
            mutex_out->unlock();
        }
    }
    mutex_in->unlock();
}
