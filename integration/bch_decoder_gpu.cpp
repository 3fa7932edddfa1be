// integration/bch_decoder_gpu.cpp -- the body of
//     void bch_decoder::execute(int* _idx_plp_simd, l1_postsignalling _l1_post, int _len_in, uint8_t* _in)
//     (/root/reference/src/DVB_T2/bch_decoder.h:41; the reference's body: bch_decoder.cpp:63-164)
// The reference strips the BCH parity without decoding (`// TODO BCH decode`, :136) and descrambles; t2gpu_bch_descramble does the same for
// all frames of the batch in one launch (the bits are still on the device when they come from ldpc_decoder's buffer, include/t2gpu.h
// "host-buffer hand-over"). One bit_descramble per FEC frame, alternating the stage's two buffers, as :139-160.
// T2GPU_BINDING_OUTER_CODE=1 in the environment: the decoder the reference leaves out runs first (t2gpu_bch_decode, in place).
#include <cstdlib>
#include <cstring>

#include "bch_decoder.h"           // the reference's
#include "t2gpu_ref_glue.h"

void bch_decoder::execute(int* _idx_plp_simd, l1_postsignalling _l1_post, int _len_in, uint8_t* _in)
{
    mutex_in->lock();
    signal_in->wakeOne();
    const l1_postsignalling_plp &plp = _l1_post.plp[_idx_plp_simd[0]];
    int k_bch = 0, n_bch = 0;
    if (t2gpu_bch_info(plp.plp_fec_type, plp.plp_cod, nullptr, nullptr, &k_bch, &n_bch) != 0) { mutex_in->unlock(); return; }
    const int frames = _len_in / n_bch;
    static const bool outer_code = std::getenv("T2GPU_BINDING_OUTER_CODE") && std::atoi(std::getenv("T2GPU_BINDING_OUTER_CODE")) != 0;
    static std::vector<uint8_t> rows;                       // [frames][k_bch]
    static std::vector<int32_t> status;
    rows.resize((size_t)frames * (size_t)k_bch);
    if (outer_code) {
        status.resize((size_t)frames);
        if (t2gpu_bch_decode(plp.plp_fec_type, plp.plp_cod, _in, frames, status.data()) != frames) t2glue::complain("t2gpu_bch_decode");
    }
    if (t2gpu_bch_descramble(plp.plp_fec_type, plp.plp_cod, _in, frames, rows.data()) != k_bch) {
        t2glue::complain("t2gpu_bch_descramble");
        mutex_in->unlock();
        return;
    }
    for (int n = 0; n < frames; ++n) {
        uint8_t *frame = swap_buffer ? buffer_a : buffer_b;
        std::memcpy(frame, rows.data() + (size_t)n * (size_t)k_bch, (size_t)k_bch);
        swap_buffer = !swap_buffer;
        mutex_out->lock();
        emit bit_descramble(_idx_plp_simd[n], _l1_post, k_bch, frame);
        signal_out->wait(mutex_out);
        mutex_out->unlock();
    }
    out = swap_buffer ? buffer_a : buffer_b;
    mutex_in->unlock();
}
