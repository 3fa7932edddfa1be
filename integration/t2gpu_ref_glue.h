// integration/t2gpu_ref_glue.h -- shared by the slot bodies in this directory.
//
// The files here are the REFERENCE-SIDE BINDING: what a maintainer of Oleg-Malyutin/sdr_receiver_dvb_t2 adds to put libt2gpu.so behind
// the existing Qt signal flow. Each *.cpp defines the body of one of the reference's slots (same class, same signature, same hand-shake
// with the neighbouring stages' mutex / wait-condition pairs) as a call into the C ABI of include/t2gpu.h; the reference's own .cpp keeps
// everything else (constructors, threads, connect() chains, buffers). They include the reference's OWN headers and are compiled against
// them -- oracle/Makefile, target `binding`, does that here with the image's Qt 5.9.7 SDK, and links the result with the reference's
// objects into oracle/_ref/libref_t2rx_gpu*.so, where the replaced slot bodies take the place of the reference's (objcopy
// --localize-symbol on the reference's object file: no reference source is touched or copied). tests/test_binding.py builds them,
// tests/test_binding_gpu.py runs the reference's receiver with them on a GPU.
//
// Handles of the library live at file scope here (a maintainer would make them members of the classes); one device (0).
#ifndef T2GPU_REF_GLUE_H
#define T2GPU_REF_GLUE_H
#include <cstdio>
#include <vector>

#include "dvbt2_definition.h"      // the reference's (src/DVB_T2 on the include path)
#include "t2gpu.h"

namespace t2glue {

// l1_postsignalling_plp / dynamic_plp (dvbt2_definition.h:284-316) -> the ABI's plain structs, field by field
inline t2gpu_l1_plp to_c(const l1_postsignalling_plp &p)
{
    t2gpu_l1_plp c;
    c.id = p.id; c.plp_type = p.plp_type; c.plp_payload_type = p.plp_payload_type; c.ff_flag = p.ff_flag; c.first_rf_idx = p.first_rf_idx;
    c.first_frame_idx = p.first_frame_idx; c.plp_group_id = p.plp_group_id; c.plp_cod = p.plp_cod; c.plp_mod = p.plp_mod;
    c.plp_rotation = p.plp_rotation; c.plp_fec_type = p.plp_fec_type; c.plp_num_blocks_max = p.plp_num_blocks_max;
    c.frame_interval = p.frame_interval; c.time_il_length = p.time_il_length; c.time_il_type = p.time_il_type;
    c.in_band_a_flag = p.in_band_a_flag; c.in_band_b_flag = p.in_band_b_flag; c.reserved_1 = p.reserved_1; c.plp_mode = p.plp_mode;
    c.static_flag = p.static_flag; c.static_padding_flag = p.static_padding_flag;
    return c;
}
inline t2gpu_l1_dyn_plp to_c(const dynamic_plp &d)
{
    t2gpu_l1_dyn_plp c;
    c.id = d.id; c.start = d.start; c.num_blocks = d.num_blocks; c.reserved_2 = d.reserved_2;
    return c;
}
inline void plps_of(const l1_postsignalling &post, std::vector<t2gpu_l1_plp> &plp, std::vector<t2gpu_l1_dyn_plp> &dyn)
{
    plp.resize((size_t)post.num_plp);
    dyn.resize((size_t)post.num_plp);
    for (int i = 0; i < post.num_plp; ++i) { plp[(size_t)i] = to_c(post.plp[i]); dyn[(size_t)i] = to_c(post.dyn.plp[i]); }
}
inline void complain(const char *what) { std::fprintf(stderr, "%s: %s\n", what, t2gpu_last_error()); }

}  // namespace t2glue
#endif
