// l1_signalling.cpp -- host-side L1-pre / L1-post signalling extraction from the equalised P2 cells (C-ABI t2gpu_l1_*).
//
// Replaces p2_symbol::l1_pre_info (/root/reference/src/DVB_T2/p2_symbol.cpp:301-532) and p2_symbol::l1_post_info with its
// field parsers (:534-1089). Like the reference it reads the systematic bits only -- hard decisions on the L1 cells, no
// BCH/LDPC decoding of the L1 blocks -- and accepts a block when its CRC-32 matches. A few hundred bits per
// T2 frame: host code, as in the reference. Field layout: ETSI EN 302 755 7.2.2 (L1-pre, 200 bits) and 7.2.3 (L1-post).
#include "../../include/t2gpu.h"
#include "t2gpu_common.h"
#include <cmath>
#include <cstring>
#include <vector>

using namespace t2gpu;

namespace {
struct BitReader {
    const uint8_t *b; int pos;
    int32_t get(int n) { int32_t v = 0; for (int i = 0; i < n; ++i) v = (v << 1) | (b[pos++] & 1); return v; }
};
uint32_t crc32_bits(const uint8_t *bits, int n)      // generator 0x04C11DB7, register preset to all ones, MSB first
{
    uint32_t crc = 0xffffffffu;
    for (int i = 0; i < n; ++i) {
        uint32_t b = (bits[i] & 1u) ^ ((crc >> 31) & 1u);
        crc <<= 1;
        if (b) crc ^= 0x04C11DB7u;
    }
    return crc;
}
}

extern "C" int t2gpu_l1_pre_parse(const float *p2_cells, t2gpu_l1_pre *out)
{
    if (!p2_cells || !out) { set_error("t2gpu_l1_pre_parse: bad arguments"); return -1; }
    uint8_t bits[200];
    for (int i = 0; i < 200; ++i) bits[i] = p2_cells[2 * i] > 0 ? 0 : 1;          // BPSK: bit 0 -> +Re (:309,351)
    BitReader r{bits, 0};
    out->type = r.get(8); out->bwt_ext = r.get(1); out->s1 = r.get(3); out->s2_field1 = r.get(3); out->s2_field2 = r.get(1);
    out->l1_repetition_flag = r.get(1); out->guard_interval = r.get(3); out->papr = r.get(4); out->l1_post_mod = r.get(4);
    out->l1_cod = r.get(2); out->l1_fec_type = r.get(2); out->l1_post_size = r.get(18); out->l1_post_info_size = r.get(18);
    out->pilot_pattern = r.get(4); out->tx_id_availability = r.get(8); out->cell_id = r.get(16); out->network_id = r.get(16);
    out->t2_system_id = r.get(16); out->num_t2_frames = r.get(8); out->num_data_symbols = r.get(12); out->regen_flag = r.get(3);
    out->l1_post_extension = r.get(1); out->num_rf = r.get(3); out->current_rf_index = r.get(3); out->t2_version = r.get(4);
    out->l1_post_scrambled = r.get(1); out->t2_base_lite = r.get(1); out->reserved = r.get(4);
    uint32_t field = 0;
    for (int i = 168; i < 200; ++i) field = (field << 1) | bits[i];
    out->crc_32 = field;
    return crc32_bits(bits, 168) == field ? 1 : 0;
}

extern "C" int t2gpu_l1_post_parse(const float *l1_post_cells, const t2gpu_l1_pre *pre, t2gpu_l1_post *post, t2gpu_l1_plp *plp,
                                   t2gpu_l1_dyn_plp *dyn, int max_plp)
{
    if (!l1_post_cells || !pre || !post || !plp || !dyn || max_plp < 1) { set_error("t2gpu_l1_post_parse: bad arguments"); return -1; }
    const int mod = pre->l1_post_mod;
    if (mod < 0 || mod > 3 || pre->l1_post_size < 1 || pre->l1_post_info_size < 1) { set_error("t2gpu_l1_post_parse: bad L1-pre"); return -1; }
    const int bits_per_cell = mod == 0 ? 1 : 2 * mod;                 // BPSK, QPSK, 16-QAM, 64-QAM
    const int n_post = pre->l1_post_size * bits_per_cell;
    if (pre->l1_post_info_size + 32 > n_post) { set_error("t2gpu_l1_post_parse: info size exceeds the block"); return -1; }
    // hard demap (:596-634), bit-to-cell demultiplexer (:636-642), column/row de-interleaver (:654-665, 16/64-QAM only)
    static const int mux16[8] = {7, 1, 3, 5, 2, 4, 6, 0}, mux64[12] = {11, 8, 5, 2, 10, 7, 4, 1, 9, 6, 3, 0}, mux1[1] = {0};
    const int substreams = mod == 2 ? 8 : (mod == 3 ? 12 : 1);
    const int *mux = mod == 2 ? mux16 : (mod == 3 ? mux64 : mux1);
    const float d16 = 0.316227766f, d64 = 0.15430335f;
    const float amp4 = mod == 2 ? d16 * 2 : d64 * 4, amp2 = d64 * 2;
    std::vector<uint8_t> inter(n_post + 16), bits(n_post + 16);
    int idx_mux = 0, w = 0;
    for (int c = 0; c < pre->l1_post_size; ++c) {
        const float re = l1_post_cells[2 * c], im = l1_post_cells[2 * c + 1];
        uint8_t b[6];
        b[0] = re > 0 ? 0 : 1; b[1] = im > 0 ? 0 : 1;
        b[2] = std::fabs(re) > amp4 ? 0 : 1; b[3] = std::fabs(im) > amp4 ? 0 : 1;
        b[4] = std::fabs(std::fabs(re) - amp4) > amp2 ? 0 : 1; b[5] = std::fabs(std::fabs(im) - amp4) > amp2 ? 0 : 1;
        for (int k = 0; k < bits_per_cell; ++k) {
            inter[mux[idx_mux] + w] = b[k];
            if (++idx_mux == substreams) { idx_mux = 0; w += substreams; }
        }
    }
    const int columns = mod == 2 ? 8 : (mod == 3 ? 12 : 0), rows = columns ? n_post / columns : 0, size_block = rows * columns;
    const bool scrambled = pre->t2_version > 1 && pre->l1_post_scrambled == 1;       // (:655)
    uint32_t sr = 0x4A80;                                                          // same sequence as the BB scrambler (:77-87)
    std::vector<uint8_t> prbs(n_post);
    for (int i = 0; i < n_post; ++i) { uint32_t bb = (sr ^ (sr >> 1)) & 1u; prbs[i] = (uint8_t)bb; sr = (sr >> 1) | (bb << 14); }
    int step = 0, l = 0;
    for (int i = 0; i < n_post; ++i) {
        const int j = l + step;
        bits[j] = inter[i] ^ (scrambled ? prbs[j] : 0);
        step += rows;
        if (step == size_block) { step = 0; ++l; }
    }
    uint32_t field = 0;
    for (int i = 0; i < 32; ++i) field = (field << 1) | bits[pre->l1_post_info_size + i];
    if (crc32_bits(bits.data(), pre->l1_post_info_size) != field) return 0;         // "CRC_32 ERROR" (:647-668)
    // configurable part
    BitReader r{bits.data(), 0};
    memset(post, 0, sizeof(*post));
    post->sub_slices_per_frame = r.get(15); post->num_plp = r.get(8); post->num_aux = r.get(4); post->aux_config_rfu = r.get(8);
    if (post->num_plp > max_plp || pre->num_rf > 7 || pre->num_rf < 0) { set_error("t2gpu_l1_post_parse: more PLPs / RFs than the caller provides room for"); return -1; }
    {   // the signalled counts fix the length of everything that follows (EN 302 755 7.2.3): check it BEFORE walking the fields -- the
        // reference indexes its bit array with offsets derived from the same counts and never looks at the block size (:680-697)
        const long need = 35L + 35L * pre->num_rf + ((pre->s2_field2 & 1) ? 34 : 0) + 89L * post->num_plp + 32 + 32L * post->num_aux
                          + 71 + 48L * post->num_plp + 8 + 48L * post->num_aux;
        if (need > pre->l1_post_info_size) { set_error("t2gpu_l1_post_parse: fields run past L1_POST_INFO_SIZE"); return -1; }
    }
    for (int i = 0; i < pre->num_rf; ++i) { post->rf_idx[i] = r.get(3); post->frequency[i] = (uint32_t)r.get(32); }
    if (pre->s2_field2 & 1) { post->fef_type = r.get(4); post->fef_length = r.get(22); post->fef_interval = r.get(8); }
    for (int i = 0; i < post->num_plp; ++i) {
        t2gpu_l1_plp &p = plp[i];
        p.id = r.get(8); p.plp_type = r.get(3); p.plp_payload_type = r.get(5); p.ff_flag = r.get(1); p.first_rf_idx = r.get(3);
        p.first_frame_idx = r.get(8); p.plp_group_id = r.get(8); p.plp_cod = r.get(3); p.plp_mod = r.get(3); p.plp_rotation = r.get(1);
        p.plp_fec_type = r.get(2); p.plp_num_blocks_max = r.get(10); p.frame_interval = r.get(8); p.time_il_length = r.get(8);
        p.time_il_type = r.get(1); p.in_band_a_flag = r.get(1); p.in_band_b_flag = r.get(1); p.reserved_1 = r.get(11);
        p.plp_mode = r.get(2); p.static_flag = r.get(1); p.static_padding_flag = r.get(1);
    }
    post->fef_length_msb = r.get(2); post->reserved_2 = r.get(30);
    r.pos += 32 * post->num_aux;                                                   // AUX_STREAM_TYPE / AUX_PRIVATE_CONF: not kept
    // dynamic part (current frame)
    post->frame_idx = r.get(8); post->sub_slice_interval = r.get(22); post->type_2_start = r.get(22);
    post->l1_change_counter = r.get(8); post->start_rf_idx = r.get(3); post->dyn_reserved_1 = r.get(8);
    for (int i = 0; i < post->num_plp; ++i) {
        dyn[i].id = r.get(8); dyn[i].start = r.get(22); dyn[i].num_blocks = r.get(10); dyn[i].reserved_2 = r.get(8);
    }
    post->dyn_reserved_3 = r.get(8);
    if (r.pos + 48 * post->num_aux > pre->l1_post_info_size) { set_error("t2gpu_l1_post_parse: fields run past L1_POST_INFO_SIZE"); return -1; }
    return 1;
}
