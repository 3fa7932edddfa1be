// integration/ldpc_decoder_gpu.cpp -- the body of
//     void ldpc_decoder::execute(int* _idx_plp_simd, l1_postsignalling _l1_post, int _len_in, int8_t* _in)
//     (/root/reference/src/DVB_T2/ldpc_decoder.h:90; the reference's body: ldpc_decoder.cpp:157-301)
// as one call into libt2gpu.so: the 32 frames of the SIMD batch are decoded on the GPU with the reference's own stop rule (all 32 parity
// checks or 25 sweeps; a batch that does not converge is dropped with the reference's message, :264-268), the hard bits land in the
// stage's A / B buffer and go on to bch_decoder through the same signal with the same hand-shake (:281-298).
#include "ldpc_decoder.h"          // the reference's
#include "t2gpu_ref_glue.h"

namespace { t2gpu_ldpc *gpu[2][6]; }     // one handle per (PLP_FEC_TYPE, PLP_COD), made at first use

void ldpc_decoder::execute(int* _idx_plp_simd, l1_postsignalling _l1_post, int _len_in, int8_t* _in)
{
    mutex_in->lock();
    signal_in->wakeOne();
    const l1_postsignalling_plp &plp = _l1_post.plp[_idx_plp_simd[0]];
    const int fec = plp.plp_fec_type, cod = plp.plp_cod;
    if (fec < 0 || fec > 1 || cod < 0 || cod > 5) { mutex_in->unlock(); return; }
    if (!gpu[fec][cod] && !(gpu[fec][cod] = t2gpu_ldpc_create(fec, cod, SIZEOF_SIMD, /*device*/0))) {
        t2glue::complain("t2gpu_ldpc_create");
        mutex_in->unlock();
        return;
    }
    uint8_t *out = swap_buffer ? buffer_a : buffer_b;       // [SIZEOF_SIMD][k_ldpc], one bit per byte: what :270-277 fill
    int trials_left = -1;                                   // the value LDPCDecoder::operator() returns (:263)
    if (t2gpu_ldpc_execute(gpu[fec][cod], _in, _len_in, out, &trials_left) != 0) {
        t2glue::complain("t2gpu_ldpc_execute");
        mutex_in->unlock();
        return;
    }
    if (trials_left < 0) {
        fprintf(stderr, "LDPC decoder could not recover the codeword! %d\n", trials_left);
        mutex_in->unlock();
        return;
    }
    int k_ldpc = 0;
    t2gpu_ldpc_info(gpu[fec][cod], nullptr, &k_ldpc, nullptr, nullptr);
    swap_buffer = !swap_buffer;
    mutex_out->lock();
    emit bit_bch(_idx_plp_simd, _l1_post, k_ldpc * SIZEOF_SIMD, out);
    signal_out->wait(mutex_out);
    mutex_out->unlock();
    bch_fec = swap_buffer ? buffer_a : buffer_b;
    mutex_in->unlock();
}
