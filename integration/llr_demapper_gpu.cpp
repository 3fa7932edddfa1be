// integration/llr_demapper_gpu.cpp -- the body of
//     void llr_demapper::execute(int _ti_block_size, complex* _time_deint_cell, int _plp_id, l1_postsignalling _l1_post)
//     (/root/reference/src/DVB_T2/llr_demapper.h:44-45; the reference's body: llr_demapper.cpp:132-158 with qpsk / qam16 / qam64 / qam256,
//     :160-776)
// One TI block of de-interleaved cells in; the GPU forms the reference's two sequential float sums, its LLR scale and every LLR (the
// wrapping int8 cast included) and returns whole FEC frames; they are collected into the stage's A / B buffer until SIZEOF_SIMD frames are
// there -- across TI blocks and T2 frames, as the reference's `static int blocks` does (:742-764) -- and then go to ldpc_decoder through the
// same signal with the same hand-shake.
#include <cmath>
#include <cstring>

#include "llr_demapper.h"          // the reference's
#include "t2gpu_ref_glue.h"

namespace {
t2gpu_demap *gpu = nullptr;
int gpu_key = -1, gpu_cells = 0;
std::vector<int8_t> frames_llr;          // [frames of one TI block][fec_size]
int blocks = 0;                          // frames in the batch being filled (the reference's function-local static)
int idx_plp_simd[SIZEOF_SIMD];
}

void llr_demapper::execute(int _ti_block_size, complex* _time_deint_cell, int _plp_id, l1_postsignalling _l1_post)
{
    mutex_in->lock();
    signal_in->wakeOne();
    const l1_postsignalling_plp &plp = _l1_post.plp[_plp_id];
    const int fec_size = plp.plp_fec_type ? FEC_SIZE_NORMAL : FEC_SIZE_SHORT;
    const int cells_per_fec = fec_size / (2 * (plp.plp_mod + 1));
    const int key = plp.plp_mod | (plp.plp_fec_type << 4) | (plp.plp_cod << 8) | (plp.plp_rotation << 12);
    if (!gpu || key != gpu_key || _ti_block_size > gpu_cells) {
        if (gpu) t2gpu_demap_destroy(gpu);
        gpu_cells = std::max(plp.plp_num_blocks_max * cells_per_fec, _ti_block_size);
        gpu = t2gpu_demap_create(plp.plp_mod, plp.plp_fec_type, plp.plp_cod, plp.plp_rotation, gpu_cells, /*device*/0);
        gpu_key = key;
        if (!gpu) { t2glue::complain("t2gpu_demap_create"); mutex_in->unlock(); return; }
    }
    frames_llr.resize((size_t)(_ti_block_size / cells_per_fec + 1) * (size_t)fec_size);
    float sums[3] = {0.0f, 0.0f, 0.0f};                     // sum_s, sum_e, precision (:659-676)
    const int frames = t2gpu_demap_execute(gpu, reinterpret_cast<const float *>(_time_deint_cell), _ti_block_size, frames_llr.data(), sums);
    if (frames < 0) { t2glue::complain("t2gpu_demap_execute"); mutex_in->unlock(); return; }
    emit signal_noise_ratio(plp.plp_mod == MOD_QPSK ? 10.0f * std::log10(sums[0] / sums[1]) : 20.0f * std::log10(sums[0] / sums[1]));
    for (int f = 0; f < frames; ++f) {
        int8_t *batch = swap_buffer ? buffer_a : buffer_b;
        std::memcpy(batch + (size_t)blocks * (size_t)fec_size, frames_llr.data() + (size_t)f * (size_t)fec_size, (size_t)fec_size);
        idx_plp_simd[blocks] = _plp_id;
        if (++blocks == SIZEOF_SIMD) {
            blocks = 0;
            swap_buffer = !swap_buffer;
            mutex_out->lock();
            emit soft_multiplexer_de_twist(idx_plp_simd, _l1_post, fec_size * SIZEOF_SIMD, batch);
            signal_out->wait(mutex_out);
            mutex_out->unlock();
        }
    }
    mutex_in->unlock();
}
