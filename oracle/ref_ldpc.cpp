/*
 * oracle/ref_ldpc.cpp -- TEST INFRASTRUCTURE ONLY. Driver around the REFERENCE's own LDPC decoder.
 *
 * This file contains no reference source: it #includes the reference's Qt-free headers where they lie
 * (/root/reference/src/DVB_T2/LDPC/{dvb_t2_tables,algorithms,layered_decoder}.hh, via -I in oracle/Makefile)
 * and instantiates exactly the types ldpc_decoder.h:28-63 selects (int8 code_type, FACTOR 2,
 * SIMD<int8_t,32>, NormalUpdate, OffsetMinSumAlgorithm, layered decoder). The frame shuffling around the call is our
 * restatement of ldpc_decoder.cpp:248-277. Output: oracle/_ref/libref_ldpc.so (git-ignored, travels to the GPU box).
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include "dvb_t2_tables.hh"
#include "algorithms.hh"
#include "layered_decoder.hh"

#define DEF(T) constexpr int T::DEG[]; constexpr int T::LEN[]; constexpr int T::POS[];
DEF(DVB_T2_TABLE_NORMAL_C1_2) DEF(DVB_T2_TABLE_NORMAL_C3_5) DEF(DVB_T2_TABLE_NORMAL_C2_3)
DEF(DVB_T2_TABLE_NORMAL_C3_4) DEF(DVB_T2_TABLE_NORMAL_C4_5) DEF(DVB_T2_TABLE_NORMAL_C5_6)
DEF(DVB_T2_TABLE_SHORT_C1_2) DEF(DVB_T2_TABLE_SHORT_C3_5) DEF(DVB_T2_TABLE_SHORT_C2_3)
DEF(DVB_T2_TABLE_SHORT_C3_4) DEF(DVB_T2_TABLE_SHORT_C4_5) DEF(DVB_T2_TABLE_SHORT_C5_6)

namespace {
const int W = 32;                                   // SIZEOF_SIMD under __AVX2__ (ldpc_decoder.h:28-32)
typedef SIMD<int8_t, W> simd_type;
typedef NormalUpdate<simd_type> update_type;
typedef OffsetMinSumAlgorithm<simd_type, update_type, 2> algorithm_type;
typedef LDPCDecoder<simd_type, algorithm_type> decoder_type;

struct Code { int n, k, q; decoder_type *dec; };
Code codes[12];
bool ready[12];

template <typename T> void mk(int id, int k, int q)
{
    LDPCInterface *it = new LDPC<T>();
    codes[id].n = T::N; codes[id].k = k; codes[id].q = q;
    codes[id].dec = new decoder_type();             // never destroyed (reference dtor frees new[] memory with free())
    codes[id].dec->init(it);
}
Code *get(int id)
{
    if (id < 0 || id > 11) return nullptr;
    if (!ready[id]) {
        switch (id) {                               // k_ldpc / q_ldpc pairs: ldpc_decoder.cpp:177-246
        case 0: mk<DVB_T2_TABLE_SHORT_C1_2>(id, 7200, 25); break;
        case 1: mk<DVB_T2_TABLE_SHORT_C3_5>(id, 9720, 18); break;
        case 2: mk<DVB_T2_TABLE_SHORT_C2_3>(id, 10800, 15); break;
        case 3: mk<DVB_T2_TABLE_SHORT_C3_4>(id, 11880, 12); break;
        case 4: mk<DVB_T2_TABLE_SHORT_C4_5>(id, 12600, 10); break;
        case 5: mk<DVB_T2_TABLE_SHORT_C5_6>(id, 13320, 8); break;
        case 6: mk<DVB_T2_TABLE_NORMAL_C1_2>(id, 32400, 90); break;
        case 7: mk<DVB_T2_TABLE_NORMAL_C3_5>(id, 38880, 72); break;
        case 8: mk<DVB_T2_TABLE_NORMAL_C2_3>(id, 43200, 60); break;
        case 9: mk<DVB_T2_TABLE_NORMAL_C3_4>(id, 48600, 45); break;
        case 10: mk<DVB_T2_TABLE_NORMAL_C4_5>(id, 51840, 36); break;
        case 11: mk<DVB_T2_TABLE_NORMAL_C5_6>(id, 54000, 30); break;
        }
        ready[id] = true;
    }
    return &codes[id];
}
}

/* llr_in [blocks<=32][n] frame-major; bits_out [blocks][k] (written only on success, as the reference emits nothing on
 * failure); llr_out [blocks][n] optional final LLRs re-shuffled back to the input order. Returns trials left or -1. */
extern "C" int ref_ldpc_decode(int code_id, const int8_t *llr_in, int blocks, int max_trials,
                               uint8_t *bits_out, int8_t *llr_out)
{
    Code *c = get(code_id);
    if (!c || blocks < 1 || blocks > W) return -2;
    const int n = c->n, k = c->k, q = c->q;
    simd_type *simd = new (std::align_val_t(sizeof(simd_type))) simd_type[n];
    std::memset(simd, 0, sizeof(simd_type) * n);
    for (int b = 0; b < blocks; ++b) {              // ldpc_decoder.cpp:248-260
        const int8_t *in = llr_in + (size_t)b * n;
        for (int i = 0; i < k; ++i) reinterpret_cast<int8_t *>(simd + i)[b] = in[i];
        for (int t = 0; t < q; ++t)
            for (int s = 0; s < 360; ++s)
                reinterpret_cast<int8_t *>(simd + k + q * s + t)[b] = in[k + 360 * t + s];
    }
    int count = (*c->dec)(simd, simd + k, max_trials, blocks);   // ldpc_decoder.cpp:262-263
    for (int b = 0; b < blocks; ++b) {
        if (count >= 0 && bits_out)                 // ldpc_decoder.cpp:270-277
            for (int i = 0; i < k; ++i) bits_out[(size_t)b * k + i] = reinterpret_cast<int8_t *>(simd + i)[b] < 0;
        if (llr_out) {
            int8_t *o = llr_out + (size_t)b * n;
            for (int i = 0; i < k; ++i) o[i] = reinterpret_cast<int8_t *>(simd + i)[b];
            for (int t = 0; t < q; ++t)
                for (int s = 0; s < 360; ++s) o[k + 360 * t + s] = reinterpret_cast<int8_t *>(simd + k + q * s + t)[b];
        }
    }
    operator delete[](simd, std::align_val_t(sizeof(simd_type)));
    return count;
}

extern "C" int ref_ldpc_simd_width(void) { return W; }
