// integration/dvbt2_demodulator_gpu.cpp -- the body of the boundary slot
//     void dvbt2_demodulator::execute(int _len_in, int16_t* _i_in, int16_t* _q_in, signal_estimate* signal_)
//     (/root/reference/src/DVB_T2/dvbt2_demodulator.h:78; the reference's body: dvbt2_demodulator.cpp:145-254 with symbol_acquisition,
//     :267-448)
// Everything the slot drives up to the time de-interleaver -- sample loop, resampler, P1, symbol buffering, guard correlation, FFT, the three
// equalisers, L1-pre / L1-post, acquisition and the tracking loops -- is one handle of the library (t2gpu_demod_*); the class keeps its mutex,
// its signals and its wiring: the handle's callbacks re-emit what the reference emits (l1_dyn_execute, data, amount_plp,
// replace_null_indicator) with the reference's hand-shake towards the de-interleaver's thread (:344-349, :359-364, :390-393), and call
// time_deinterleaver::start directly as :386 does.
// THREADS: with the library's default (the tracking loops on the device) the `data` callback comes from a thread of the handle's own; it
// does exactly what the reference's thread does at that point -- lock mutex_out, emit, wait -- and the emitting thread is of no concern to
// a queued connection. t2gpu_demod_set_device_loop(gpu, 0) keeps every callback on the calling thread.
#include "dvbt2_demodulator.h"     // the reference's
#include "t2gpu_ref_glue.h"

namespace {
t2gpu_demod *gpu = nullptr;
dvbt2_demodulator *owner = nullptr;

// what the callbacks hand on must outlive the call (queued connections copy l1_postsignalling by value, its arrays by pointer)
std::vector<l1_postsignalling_plp> post_plp[2];
std::vector<dynamic_plp> post_dyn[2];
int post_cur = 0;
}

// the ABI's plain L1-post structs back into the reference's (the callbacks below are lambdas inside the member function, so they reach the
// class's private members as any member does; in the reference's tree they would be private static members)
struct dvbt2_demodulator_gpu_access {
    static l1_postsignalling unpack(const t2gpu_l1_post *post, const t2gpu_l1_plp *plp, const t2gpu_l1_dyn_plp *dyn)
    {
        post_cur ^= 1;
        std::vector<l1_postsignalling_plp> &p = post_plp[post_cur];
        std::vector<dynamic_plp> &d = post_dyn[post_cur];
        p.resize((size_t)post->num_plp);
        d.resize((size_t)post->num_plp);
        for (int i = 0; i < post->num_plp; ++i) {
            l1_postsignalling_plp &q = p[(size_t)i];
            const t2gpu_l1_plp &c = plp[i];
            q.id = c.id; q.plp_type = c.plp_type; q.plp_payload_type = c.plp_payload_type; q.ff_flag = c.ff_flag; q.first_rf_idx = c.first_rf_idx;
            q.first_frame_idx = c.first_frame_idx; q.plp_group_id = c.plp_group_id; q.plp_cod = c.plp_cod; q.plp_mod = c.plp_mod;
            q.plp_rotation = c.plp_rotation; q.plp_fec_type = c.plp_fec_type; q.plp_num_blocks_max = c.plp_num_blocks_max;
            q.frame_interval = c.frame_interval; q.time_il_length = c.time_il_length; q.time_il_type = c.time_il_type;
            q.in_band_a_flag = c.in_band_a_flag; q.in_band_b_flag = c.in_band_b_flag; q.reserved_1 = c.reserved_1; q.plp_mode = c.plp_mode;
            q.static_flag = c.static_flag; q.static_padding_flag = c.static_padding_flag;
            d[(size_t)i].id = dyn[i].id; d[(size_t)i].start = dyn[i].start; d[(size_t)i].num_blocks = dyn[i].num_blocks;
            d[(size_t)i].reserved_2 = dyn[i].reserved_2;
        }
        l1_postsignalling out;
        out.sub_slices_per_frame = post->sub_slices_per_frame; out.num_plp = post->num_plp; out.num_aux = post->num_aux;
        out.aux_config_rfu = post->aux_config_rfu; out.fef_type = post->fef_type; out.fef_length = post->fef_length;
        out.fef_interval = post->fef_interval; out.fef_length_msb = post->fef_length_msb; out.reserved_2 = post->reserved_2;
        out.plp = p.data();
        out.dyn.frame_idx = post->frame_idx; out.dyn.sub_slice_interval = post->sub_slice_interval; out.dyn.type_2_start = post->type_2_start;
        out.dyn.l1_change_counter = post->l1_change_counter; out.dyn.start_rf_idx = post->start_rf_idx; out.dyn.plp = d.data();
        return out;
    }
};

void dvbt2_demodulator::execute(int _len_in, int16_t* _i_in, int16_t* _q_in, signal_estimate* signal_)
{
    mutex->lock();
    if (!gpu) {
        owner = this;
        gpu = t2gpu_demod_create(static_cast<int>(id_device), sample_rate, /*device*/0);
        if (!gpu) { t2glue::complain("t2gpu_demod_create"); mutex->unlock(); return; }
        t2gpu_demod_signals cb{};
        cb.user = this;
        cb.start = [](void *u, const t2gpu_l1_pre *pre, const t2gpu_l1_post *post, const t2gpu_l1_plp *plp, const t2gpu_l1_dyn_plp *dyn) {
            dvbt2_demodulator *d = static_cast<dvbt2_demodulator *>(u);
            l1_presignalling p;                              // time_deinterleaver::start reads l1_post_size only (time_deinterleaver.cpp:44)
            p.l1_post_size = pre->l1_post_size;
            d->deinterleaver->start(d->dvbt2, p, dvbt2_demodulator_gpu_access::unpack(post, plp, dyn));          // :386
        };
        cb.l1_dyn_execute = [](void *u, const t2gpu_l1_post *post, const t2gpu_l1_plp *plp, const t2gpu_l1_dyn_plp *dyn, int len, const float *cells) {
            dvbt2_demodulator *d = static_cast<dvbt2_demodulator *>(u);
            d->mutex_out->lock();
            emit d->l1_dyn_execute(dvbt2_demodulator_gpu_access::unpack(post, plp, dyn), len, reinterpret_cast<complex *>(const_cast<float *>(cells)));   // :390-393
            d->signal_out->wait(d->mutex_out);
            d->mutex_out->unlock();
        };
        cb.data = [](void *u, int len, const float *cells) {
            dvbt2_demodulator *d = static_cast<dvbt2_demodulator *>(u);
            d->mutex_out->lock();
            emit d->data(len, reinterpret_cast<complex *>(const_cast<float *>(cells)));                           // :344-349, :359-364
            d->signal_out->wait(d->mutex_out);
            d->mutex_out->unlock();
        };
        cb.amount_plp = [](void *u, int n) { emit static_cast<dvbt2_demodulator *>(u)->amount_plp(n); };          // :388
        cb.replace_null_indicator = [](void *u, float b1, float b2) { emit static_cast<dvbt2_demodulator *>(u)->replace_null_indicator(b1, b2); };   // :444
        t2gpu_demod_connect(gpu, &cb);
    }
    t2gpu_signal_estimate s;
    s.change_frequency = signal_->change_frequency; s.coarse_freq_offset = signal_->coarse_freq_offset;
    s.frequency_changed = signal_->frequency_changed; s.change_gain = signal_->change_gain; s.gain_offset = signal_->gain_offset;
    s.gain_changed = signal_->gain_changed; s.correct_resample = signal_->correct_resample; s.reset = signal_->reset; s.p1_reset = signal_->p1_reset;
    if (t2gpu_demod_execute(gpu, _len_in, _i_in, _q_in, &s) != 0) t2glue::complain("t2gpu_demod_execute");
    signal_->change_frequency = s.change_frequency != 0; signal_->coarse_freq_offset = s.coarse_freq_offset;
    signal_->frequency_changed = s.frequency_changed != 0; signal_->change_gain = s.change_gain != 0; signal_->gain_offset = s.gain_offset;
    signal_->gain_changed = s.gain_changed != 0; signal_->correct_resample = s.correct_resample; signal_->reset = s.reset != 0;
    signal_->p1_reset = s.p1_reset != 0;
    mutex->unlock();
}
