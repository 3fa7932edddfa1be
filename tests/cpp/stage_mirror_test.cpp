// Driver for tests/test_host_mirror_gpu.py: pushes vectors written by the Python test through the C++ stage classes of
// include/t2gpu_stages.hpp (the reference's object / slot / signal shapes over the C ABI) and writes what comes out.
//   stage_mirror_test decim  in.c64 out.c64 chunk
//   stage_mirror_test farrow in.c64 out.c64 chunk resample
//   stage_mirror_test fec    llr.i8 out.u8 fec_type code_rate       (LLR batches of 32 frames -> descrambled BBFRAME bits)
//   stage_mirror_test p1     in.c64 out.txt level
//   stage_mirror_test cells  cells.c64 out.u8 need_plp ts.u8 l1_post_size cod n_sym sizes... num_plp {mod fec rot nb_max til start nb}...
//                            (equalised cells of one T2 frame, symbol by symbol as data_symbol hands them over -> the whole
//                            FEC side: time_deinterleaver -> llr_demapper -> ldpc_decoder -> bch_decoder -> bb_de_header)
//   stage_mirror_test rx     i.s16 q.s16 out.ts buf_len need_plp log.txt
//                            (int16 I/Q as the SDR thread delivers it -> dvbt2_demodulator::execute buffer by buffer -> ... -> TS;
//                            the loop around it is rx_sdrplay::start with the tuner emulated, no AGC)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

#include "t2gpu_stages.hpp"

template <class T> std::vector<T> slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<T> v(raw.size() / sizeof(T));
    std::memcpy(v.data(), raw.data(), v.size() * sizeof(T));
    return v;
}
template <class T> void dump(const char *path, const std::vector<T> &v)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

// STAGE_EXIT_TRACE=1: marks on stderr on the way out (a process that had written all its results and did not end was seen once in ~1000
// runs on some boxes: tools/hang_hunt.py)
static void exit_mark(const char *what)
{
    static const bool on = std::getenv("STAGE_EXIT_TRACE") != nullptr;
    if (on) { std::fprintf(stderr, "exit trace: %s\n", what); std::fflush(stderr); }
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    std::atexit([] { exit_mark("atexit handler registered first in main (runs last of the atexit handlers)"); });
    const std::string mode = argv[1];
    try {
        if (mode == "decim" || mode == "farrow") {
            std::vector<t2::complex> in = slurp<t2::complex>(argv[2]), out;
            const int chunk = std::atoi(argv[4]);
            double resample = mode == "farrow" ? std::atof(argv[5]) : 0.5;
            t2::filter_decimator dec(1 << 16);
            t2::interpolator_farrow far(1 << 16);
            std::vector<t2::complex> buf((size_t)chunk * 4 + 64);
            for (size_t pos = 0; pos < in.size(); pos += (size_t)chunk) {
                const int n = (int)std::min<size_t>((size_t)chunk, in.size() - pos);
                int len_out = 0;
                if (mode == "decim") dec.execute(n, in.data() + pos, len_out, buf.data());
                else far(n, in.data() + pos, resample, len_out, buf.data());
                out.insert(out.end(), buf.begin(), buf.begin() + len_out);
            }
            dump(argv[3], out);
        } else if (mode == "fec") {
            std::vector<int8_t> llr = slurp<int8_t>(argv[2]);
            const int fec_type = std::atoi(argv[4]), cod = std::atoi(argv[5]);
            const int fec_size = fec_type ? 64800 : 16200;
            t2::l1_postsignalling l1;
            l1.plp.resize(1);
            l1.plp[0].plp_fec_type = fec_type; l1.plp[0].plp_cod = cod;
            t2::ldpc_decoder ldpc(0, 8, std::getenv("STAGE_THREADS") && std::atoi(std::getenv("STAGE_THREADS")) != 0,
                                  !(std::getenv("STAGE_MERGE") && std::atoi(std::getenv("STAGE_MERGE")) == 0));   // STAGE_MERGE=0: one launch per SIMD batch
            t2::bch_decoder bch;
            bch.outer_code = argc > 6 && std::string(argv[6]) == "outer";       // the library's opt-in BCH correction in front
            std::vector<uint8_t> out;
            // the reference's connect() chain: ldpc.bit_bch -> bch.execute, bch.bit_descramble -> (here) collect
            ldpc.bit_bch = [&](int *idx, const t2::l1_postsignalling &p, int len, uint8_t *bits) {
                bch.execute(idx, p, len, bits);
                for (int32_t st : bch.outer_code_status) std::fprintf(stderr, "outer %d\n", st);
            };
            bch.bit_descramble = [&](int plp_id, const t2::l1_postsignalling &, int len, uint8_t *bits) {
                out.push_back((uint8_t)plp_id);
                out.insert(out.end(), bits, bits + len);
            };
            int idx_plp_simd[t2::SIZEOF_SIMD] = {};
            const size_t batch = (size_t)fec_size * t2::SIZEOF_SIMD;
            for (size_t pos = 0; pos + batch <= llr.size(); pos += batch) ldpc.execute(idx_plp_simd, l1, (int)batch, llr.data() + pos);
            ldpc.flush();                                              // the stage keeps batches in flight: the last ones come out here
            dump(argv[3], out);
        } else if (mode == "p1") {
            std::vector<t2::complex> in = slurp<t2::complex>(argv[2]);
            t2::p1_symbol p1((int)in.size());
            std::vector<t2::complex> buffer_sym(4096);
            int consume = 0, idx_buffer_sym = 0, preamble = -1, fft_mode = -1;
            double cfo = 0;
            bool decoded = false, reset = false;
            const bool hit = p1.execute(true, (float)std::atof(argv[4]), (int)in.size(), in.data(), consume, buffer_sym.data(), idx_buffer_sym,
                                        preamble, fft_mode, cfo, decoded, reset);
            std::FILE *f = std::fopen(argv[3], "w");
            std::fprintf(f, "%d %d %d %d %d %d %.3f\n", hit ? 1 : 0, consume, idx_buffer_sym, preamble, fft_mode, decoded ? 1 : 0, cfo);
            std::fclose(f);
        } else if (mode == "cells") {
            std::vector<t2::complex> cells = slurp<t2::complex>(argv[2]);
            int a = 4;
            const int need_plp = std::atoi(argv[a++]);
            const char *ts_path = argv[a++];
            t2gpu_l1_pre pre{};
            pre.l1_post_size = std::atoi(argv[a++]);
            const int cod = std::atoi(argv[a++]);
            std::vector<int> sizes((size_t)std::atoi(argv[a++]));
            for (int &v : sizes) v = std::atoi(argv[a++]);
            t2::l1_postsignalling l1;
            l1.post.num_plp = std::atoi(argv[a++]);
            l1.plp.resize((size_t)l1.post.num_plp);
            l1.dyn_plp.resize(l1.plp.size());
            for (size_t i = 0; i < l1.plp.size(); ++i) {
                t2gpu_l1_plp &p = l1.plp[i];
                p.id = (int)i; p.plp_cod = cod;
                p.plp_mod = std::atoi(argv[a++]); p.plp_fec_type = std::atoi(argv[a++]); p.plp_rotation = std::atoi(argv[a++]);
                p.plp_num_blocks_max = std::atoi(argv[a++]); p.time_il_length = std::atoi(argv[a++]);
                l1.dyn_plp[i].id = (int)i;
                l1.dyn_plp[i].start = std::atoi(argv[a++]); l1.dyn_plp[i].num_blocks = std::atoi(argv[a++]);
            }
            // the reference's connect() chain from the time de-interleaver down (time_deinterleaver.cpp:30, llr_demapper.cpp:83,
            // ldpc_decoder.cpp:149, bch_decoder.cpp:43)
            // STAGE_THREADS=1: the de-interleaver and the LDPC stage each emit on a thread of their own (as the reference's stage objects do)
            const bool threads = std::getenv("STAGE_THREADS") && std::atoi(std::getenv("STAGE_THREADS")) != 0;
            t2::time_deinterleaver ti(0, threads);
            t2::llr_demapper qam;
            t2::ldpc_decoder ldpc(0, 8, threads, !(std::getenv("STAGE_MERGE") && std::atoi(std::getenv("STAGE_MERGE")) == 0));
            t2::bch_decoder bch;
            t2::bb_de_header deheader(need_plp);
            // STAGE_HANDOFF=0: the hand-over of stage outputs by address, which the stage classes switch on, off again
            if (std::getenv("STAGE_HANDOFF") && std::atoi(std::getenv("STAGE_HANDOFF")) == 0) t2::handoff(false);
            std::vector<uint8_t> out, ts;
            // STAGE_DUMP=<prefix>: what crosses the first two signals is also appended to <prefix>.ti.c64 / <prefix>.llr.i8
            const char *dump_to = std::getenv("STAGE_DUMP");
            auto append = [&](const char *ext, const void *p, size_t bytes) {
                if (!dump_to) return;
                std::FILE *f = std::fopen((std::string(dump_to) + ext).c_str(), "ab");
                if (f) { std::fwrite(p, 1, bytes, f); std::fclose(f); }
            };
            ti.ti_block = [&](int n, t2::complex *c, int plp, const t2::l1_postsignalling &p) { append(".ti.c64", c, (size_t)n * sizeof(t2::complex)); qam.execute(n, c, plp, p); };
            qam.soft_multiplexer_de_twist = [&](int *idx, const t2::l1_postsignalling &p, int len, int8_t *llr) { append(".llr.i8", llr, (size_t)len); ldpc.execute(idx, p, len, llr); };
            ldpc.bit_bch = [&](int *idx, const t2::l1_postsignalling &p, int len, uint8_t *bits) { bch.execute(idx, p, len, bits); };
            bch.bit_descramble = [&](int plp_id, const t2::l1_postsignalling &p, int len, uint8_t *bits) {
                out.push_back((uint8_t)plp_id);
                out.insert(out.end(), bits, bits + len);
                deheader.execute(plp_id, p, len, bits);
            };
            deheader.write_out = [&](const uint8_t *b, int n) { ts.insert(ts.end(), b, b + n); };
            ti.start(pre, l1);
            size_t pos = 0;
            for (size_t l = 0; l < sizes.size(); ++l) {                      // dvbt2_demodulator.cpp:348-375: P2 first, then every symbol
                if (l == 0) ti.l1_dyn_execute(l1, sizes[l], cells.data() + pos);
                else ti.execute(sizes[l], cells.data() + pos);
                pos += (size_t)sizes[l];
            }
            ti.flush();
            ldpc.flush();
            dump(argv[3], out);
            dump(ts_path, ts);
        } else if (mode == "rx") {
            std::vector<int16_t> vi = slurp<int16_t>(argv[2]), vq = slurp<int16_t>(argv[3]);
            const char *ts_path = argv[4];
            const int buf_len = std::atoi(argv[5]), need_plp = std::atoi(argv[6]);
            std::FILE *log = std::fopen(argv[7], "w");
            // STAGE_SAMPLE_RATE_OFF=<Hz>: the sample rate the demodulator is TOLD, minus the recording's true 64e6 / 7 (a receiver clock that is
            // off: what the sample-rate tracker has to find, dvbt2_demodulator.cpp:430-439)
            const float rate_off = std::getenv("STAGE_SAMPLE_RATE_OFF") ? (float)std::atof(std::getenv("STAGE_SAMPLE_RATE_OFF")) : 0.0f;
            t2::dvbt2_demodulator demodulator(t2::id_sdrplay, 64.0e6f / 7.0f + rate_off);
            // STAGE_TRACE=<path>: the loop trajectory (t2gpu_demod_set_trace), raw doubles
            std::vector<double> trace;
            if (std::getenv("STAGE_TRACE")) { trace.resize((size_t)T2GPU_DEMOD_TRACE_W * 65536); demodulator.set_trace(trace.data(), 65536); }
            // STAGE_TUNER_MOVES="k:hz,k:hz": the recording sits behind an EXTERNAL emulated tuner (tests/ref_cases.py RX_OFFSET_CASES): it has been
            // rotated already by the moves the reference asked for, from buffer k on. A re-tune request must then come at one of those buffers;
            // the SDR thread's bookkeeping runs with the listed value (what the tuner was told) and set_tuner is not used.
            std::vector<std::pair<long, double>> ext_moves;
            const bool ext_tuner = std::getenv("STAGE_TUNER_MOVES") != nullptr;
            if (ext_tuner) {
                const char *q = std::getenv("STAGE_TUNER_MOVES");
                while (*q) {
                    char *e = nullptr;
                    const long k = std::strtol(q, &e, 10);
                    if (*e != ':') break;
                    const double hz = std::strtod(e + 1, &e);
                    ext_moves.emplace_back(k, hz);
                    q = *e == ',' ? e + 1 : e;
                }
            }
            long buf_no = 0;
            // STAGE_DEVICE_LOOP=0 / 1: the tracking loops of a frame's data symbols on the host / on the device (the library's default; same cells, same TS)
            if (std::getenv("STAGE_DEVICE_LOOP")) demodulator.set_device_loop(std::atoi(std::getenv("STAGE_DEVICE_LOOP")) != 0);
            // STAGE_CHAIN_ONE=0: the completing chunk and the symbol's transform as two launches (t2gpu_demod_set_chain_one)
            if (std::getenv("STAGE_CHAIN_ONE") && std::atoi(std::getenv("STAGE_CHAIN_ONE")) == 0) demodulator.set_chain_one(false);
            // STAGE_PIN=1: the I/Q buffers page-locked (t2gpu_host_pin) -- they then come over chunk by chunk inside the chunks' launches
            // (t2gpu_demod_set_copy_ahead; STAGE_COPY_AHEAD=0: the whole buffer by one launch at the head of each call)
            if (std::getenv("STAGE_PIN") && std::atoi(std::getenv("STAGE_PIN")) != 0) { t2gpu_host_pin(vi.data(), vi.size() * 2); t2gpu_host_pin(vq.data(), vq.size() * 2); }
            if (std::getenv("STAGE_COPY_AHEAD") && std::atoi(std::getenv("STAGE_COPY_AHEAD")) == 0) demodulator.set_copy_ahead(false);
            t2::llr_demapper qam;
            t2::ldpc_decoder ldpc;
            t2::bch_decoder bch;
            t2::bb_de_header deheader(need_plp);
            std::vector<uint8_t> ts;
            long bbframes = 0;
            const char *dump_fec = std::getenv("STAGE_DUMP");
            auto append_fec = [&](const char *ext, const void *p, size_t bytes) {
                if (!dump_fec) return;
                std::FILE *f = std::fopen((std::string(dump_fec) + ext).c_str(), "ab");
                if (f) { std::fwrite(p, 1, bytes, f); std::fclose(f); }
            };
            demodulator.deinterleaver->ti_block = [&](int n, t2::complex *c, int plp, const t2::l1_postsignalling &p) {
                if (dump_fec) {
                    const t2gpu_l1_plp &q = p.plp.at((size_t)plp);
                    std::fprintf(log, "ti_block %d plp %d mod %d cod %d fec %d rot %d blocks_max %d til %d type %d dyn start %d blocks %d\n", n, plp, q.plp_mod, q.plp_cod,
                                 q.plp_fec_type, q.plp_rotation, q.plp_num_blocks_max, q.time_il_length, q.time_il_type, p.dyn_plp.at((size_t)plp).start, p.dyn_plp.at((size_t)plp).num_blocks);
                    append_fec(".ti.c64", c, (size_t)n * sizeof(t2::complex));
                }
                qam.execute(n, c, plp, p);
            };
            qam.soft_multiplexer_de_twist = [&](int *idx, const t2::l1_postsignalling &p, int len, int8_t *llr) { append_fec(".llr.i8", llr, (size_t)len); ldpc.execute(idx, p, len, llr); };
            ldpc.bit_bch = [&](int *idx, const t2::l1_postsignalling &p, int len, uint8_t *bits) { bch.execute(idx, p, len, bits); };
            bch.bit_descramble = [&](int plp_id, const t2::l1_postsignalling &p, int len, uint8_t *bits) { ++bbframes; deheader.execute(plp_id, p, len, bits); };
            deheader.write_out = [&](const uint8_t *b, int n) { ts.insert(ts.end(), b, b + n); };
            demodulator.amount_plp = [&](int n) { std::fprintf(log, "amount_plp %d\n", n); };
            if (const char *dump_to = std::getenv("STAGE_DUMP")) {             // the cells of every signal, appended to <prefix>.cells.c64 (+ a line per signal in the log)
                const std::string path = std::string(dump_to) + ".cells.c64";
                auto append = [path](const t2::complex *c, int n) {
                    std::FILE *f = std::fopen(path.c_str(), "ab");
                    if (f) { std::fwrite(c, sizeof(t2::complex), (size_t)n, f); std::fclose(f); }
                };
                auto fwd_l1 = demodulator.l1_dyn_execute;
                auto fwd_data = demodulator.data;
                demodulator.l1_dyn_execute = [=](const t2::l1_postsignalling &p, int len, t2::complex *c) { std::fprintf(log, "cells p2 %d\n", len); append(c, len); fwd_l1(p, len, c); };
                demodulator.data = [=](int len, t2::complex *c) { std::fprintf(log, "cells data %d\n", len); append(c, len); fwd_data(len, c); };
            }
            // ---- the SDR thread (rx_sdrplay.cpp:135-261) over a recording
            t2::signal_estimate signal;
            double rf_frequency = 0, ch_frequency = 626.0e6, tuner_hz = 0;
            bool frequency_changed = true, gain_changed = true;
            auto set_rf_frequency = [&]() {                                  // :158-176
                if (!signal.frequency_changed) signal.frequency_changed = frequency_changed;
                if (signal.change_frequency) {
                    signal.change_frequency = false;
                    frequency_changed = false;
                    signal.frequency_changed = false;
                    if (ext_tuner && signal.coarse_freq_offset != 0.0) {
                        double listed = 0.0;
                        bool found = false;
                        for (const auto &mv : ext_moves) if (mv.first == buf_no) { listed = mv.second; found = true; }
                        std::fprintf(log, "set_rf_ext %ld %.6f %s\n", buf_no, signal.coarse_freq_offset, found ? "listed" : "UNLISTED");
                        if (found) signal.coarse_freq_offset = listed;
                    }
                    signal.correct_resample = signal.coarse_freq_offset / rf_frequency;
                    rf_frequency += signal.coarse_freq_offset;
                    tuner_hz += signal.coarse_freq_offset;                   // mir_sdr_SetRf: the recording cannot be re-tuned, the demodulator's
                    if (!ext_tuner) demodulator.set_tuner(tuner_hz);         // extra NCO term stands in for the local oscillator
                    std::fprintf(log, "set_rf %.3f\n", tuner_hz);
                }
            };
            auto set_gain = [&]() {                                          // :178-197 with agc = false
                if (!signal.gain_changed) signal.gain_changed = gain_changed;
            };
            auto reset = [&]() {                                             // :135-156
                signal.reset = false;
                rf_frequency = ch_frequency;
                tuner_hz = 0;
                if (!ext_tuner) demodulator.set_tuner(0.0);
                signal.coarse_freq_offset = 0.0;
                signal.change_frequency = true;
                signal.correct_resample = 0.0;
                signal.gain_offset = 0;
                signal.change_gain = true;
                set_rf_frequency();
                set_gain();
                std::fprintf(log, "reset\n");
            };
            reset();
            const auto t_begin = std::chrono::steady_clock::now();
            for (size_t pos = 0; pos + (size_t)buf_len <= vi.size(); pos += (size_t)buf_len) {
                buf_no = (long)(pos / (size_t)buf_len);
                frequency_changed = true;                                    // rf_changed / gr_changed arrive with the next packets (:216-223)
                gain_changed = true;
                if (signal.reset) { reset(); continue; }                     // :229-235 (the buffer is dropped)
                set_rf_frequency();
                set_gain();
                demodulator.execute(buf_len, vi.data() + pos, vq.data() + pos, &signal);
                const t2gpu_demod_info st = demodulator.status();
                std::fprintf(log, "buf %zu next %d p2_init %d init %d deint %d crc %d gi %d sym %ld frames %ld resets %ld level %.5f cfo %.2f fe %.3e res %.9f\n",
                             pos / (size_t)buf_len, st.next_symbol_type, st.p2_init, st.demodulator_init, st.deint_start, st.crc32_l1_pre,
                             st.guard_interval_size, (long)st.symbols, (long)st.frames, (long)st.resets, st.level_detect, signal.coarse_freq_offset,
                             st.frequency_est_filtered, st.arbitrary_resample);
            }
            demodulator.flush();
            ldpc.flush();
            if (const char *tp = std::getenv("STAGE_TRACE")) {
                long n = demodulator.trace_count();
                if (n > 65536) n = 65536;
                std::FILE *f = std::fopen(tp, "wb");
                if (f) { std::fwrite(trace.data(), sizeof(double), (size_t)n * T2GPU_DEMOD_TRACE_W, f); std::fclose(f); }
                demodulator.set_trace(nullptr, 0);
            }
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
            std::fprintf(log, "bbframes %ld ts %zu\n", bbframes, ts.size());
            std::fprintf(log, "wall %.3f s for %zu samples = %.2f Msamples/s (real time: 9.14)\n", secs, vi.size(), vi.size() / secs / 1e6);
            std::fclose(log);
            dump(ts_path, ts);
            exit_mark("rx: results written, the stage objects go out of scope");
        } else return 2;
        exit_mark("the mode's objects are destroyed");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "stage_mirror_test: %s\n", e.what());
        return 1;
    }
    exit_mark("main returns");
    return 0;
}
