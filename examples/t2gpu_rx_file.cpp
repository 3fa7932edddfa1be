// t2gpu_rx_file -- a Qt-free stand-in for the reference's source and sink around the accelerated path: int16 I/Q from files in,
// transport stream out to a UDP port or a file.
//
//   source  rx_sdrplay::start / set_rf_frequency / set_gain / reset   (/root/reference/src/rx_sdrplay.cpp:135-261): the loop that
//           hands buffers to dvbt2_demodulator::execute and serves its requests (re-tune, reset). A recording has no tuner: the
//           requested moves of the local oscillator go to t2::dvbt2_demodulator::set_tuner, there is no AGC.
//   sink    bb_de_header's output (/root/reference/src/DVB_T2/bb_de_header.cpp:433-443, set_out :500-525): one UDP datagram per
//           BBFRAME to 127.0.0.1:<port> (the reference's `vlc udp://@:7654`), or the raw bytes appended to a file.
//   between t2::dvbt2_demodulator -> time_deinterleaver -> llr_demapper -> ldpc_decoder -> bch_decoder -> bb_de_header, wired as the
//           reference's constructors wire them (include/t2gpu_stages.hpp), every stage a call into libt2gpu.so.
//
// build:  g++ -O2 -std=c++17 -I../include t2gpu_rx_file.cpp -L../sdr_receiver_dvb_t2_amd -lt2gpu -Wl,-rpath,$PWD/../sdr_receiver_dvb_t2_amd -o t2gpu_rx_file
// usage:  t2gpu_rx_file i.s16 q.s16 (--out ts.bin | --udp 7654) [--plp 0] [--buf 172032] [--device 0] [--warm 0] [--json 1] [--saturate 0] [--threads 1] [--device-loop 1] [--ldpc-in-flight 8] [--ldpc-merge 1] [--chain-one 1] [--copy-ahead 1]
//         --buf: samples per execute() call (the reference's SDRplay thread hands over norm_blocks x 384 = 172 032, rx_sdrplay.h:64)
//         --warm n: the first n buffers (acquisition: P1, guard search, L1) run before the clock starts; --json 1: one JSON line on
//         stdout with the throughput of the timed buffers (bench.py's drop_in leg reads it); --saturate 1: LLRs clamped to int8
//         instead of the reference's wrapping cast (an extension, t2::llr_demapper::saturate_llr -- with the cast a 256-QAM PLP in
//         AWGN loses every SIMD batch in the LDPC stage, in the reference as here); --threads 1 (default): the LDPC stage emits on a
//         thread of its own, as the reference's stage objects live on their own QThreads -- BCH, de-framing and the sink then run beside
//         the demodulator; 0: everything on the calling thread. With
//         1 the time de-interleaver likewise hands its TI blocks (demapping, batch forming, LDPC submission) to a thread of its own
#include <arpa/inet.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include "t2gpu_stages.hpp"

static std::vector<int16_t> slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<int16_t> v(raw.size() / 2);
    std::memcpy(v.data(), raw.data(), v.size() * 2);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s i.s16 q.s16 (--out file | --udp port) [--plp n] [--buf samples] [--device n]\n", argv[0]);
        return 2;
    }
    const char *out_path = nullptr;
    int udp_port = 0, need_plp = 0, buf_len = 172032, device = 0, warm = 0, json = 0, saturate = 0, threads = 1, device_loop = 1, in_flight = 8, ldpc_merge = 1, chain_one = 1, copy_ahead = 1;
    for (int a = 3; a + 1 < argc; a += 2) {
        if (!std::strcmp(argv[a], "--out")) out_path = argv[a + 1];
        else if (!std::strcmp(argv[a], "--udp")) udp_port = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--plp")) need_plp = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--buf")) buf_len = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--device")) device = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--warm")) warm = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--json")) json = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--saturate")) saturate = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--threads")) threads = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--device-loop")) device_loop = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--ldpc-in-flight")) in_flight = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--ldpc-merge")) ldpc_merge = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--chain-one")) chain_one = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--copy-ahead")) copy_ahead = std::atoi(argv[a + 1]);
    }
    if ((!out_path && !udp_port) || buf_len < 4096) return 2;
    const std::vector<int16_t> vi = slurp(argv[1]), vq = slurp(argv[2]);
    // an SDR source hands over the same few buffers again and again and would page-lock them once; the recording stands in for them
    t2gpu_host_pin(const_cast<int16_t *>(vi.data()), vi.size() * 2);
    t2gpu_host_pin(const_cast<int16_t *>(vq.data()), vq.size() * 2);

    std::FILE *file = out_path ? std::fopen(out_path, "wb") : nullptr;
    int sock = -1;
    sockaddr_in to{};
    if (udp_port) {
        sock = socket(AF_INET, SOCK_DGRAM, 0);
        to.sin_family = AF_INET; to.sin_port = htons((uint16_t)udp_port); to.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    }
    try {
        t2::dvbt2_demodulator demodulator(t2::id_sdrplay, 64.0e6f / 7.0f, device, threads != 0);
        // --device-loop 1 (default): the tracking loops of a frame's data symbols on the device, their cells handed on by a thread of the
        // demodulator's own (t2gpu_demod_set_device_loop); 0: on the host, one round trip per symbol, every signal on this thread
        demodulator.set_device_loop(device_loop != 0);
        demodulator.set_chain_one(chain_one != 0);                        // (A/B: the completing chunk and the symbol's transform as one launch or two)
        demodulator.set_copy_ahead(copy_ahead != 0);                      // (A/B: the I/Q chunk by chunk inside the chunks' launches, or the whole buffer at the head of a call)
        t2::llr_demapper qam(device);
        qam.saturate_llr = saturate != 0;
        t2::ldpc_decoder ldpc(device, in_flight, threads != 0, ldpc_merge != 0, ldpc_merge > 1 ? ldpc_merge : 250);   // --ldpc-merge 0: one launch per SIMD batch; n > 1: the linger in us                   // --threads 1: the LDPC stage and what is wired behind it on a thread of their own
        t2::bch_decoder bch;
        t2::bb_de_header deheader(need_plp);
        long bbframes = 0, ts_bytes = 0;
        demodulator.deinterleaver->ti_block = [&](int n, t2::complex *c, int plp, const t2::l1_postsignalling &p) { qam.execute(n, c, plp, p); };
        qam.soft_multiplexer_de_twist = [&](int *idx, const t2::l1_postsignalling &p, int len, int8_t *llr) { ldpc.execute(idx, p, len, llr); };
        ldpc.bit_bch = [&](int *idx, const t2::l1_postsignalling &p, int len, uint8_t *bits) { bch.execute(idx, p, len, bits); };
        bch.bit_descramble = [&](int plp_id, const t2::l1_postsignalling &p, int len, uint8_t *bits) { ++bbframes; deheader.execute(plp_id, p, len, bits); };
        deheader.write_out = [&](const uint8_t *b, int n) {
            ts_bytes += n;
            if (file) std::fwrite(b, 1, (size_t)n, file);
            if (sock >= 0) sendto(sock, b, (size_t)n, 0, reinterpret_cast<const sockaddr *>(&to), sizeof to);
        };
        demodulator.amount_plp = [&](int n) { std::fprintf(stderr, "PLPs: %d\n", n); };

        t2::signal_estimate signal;
        double rf_frequency = 0, tuner_hz = 0;
        const double ch_frequency = 626.0e6;
        bool frequency_changed = true, gain_changed = true;
        auto set_rf_frequency = [&]() {                                   // rx_sdrplay.cpp:158-176
            if (!signal.frequency_changed) signal.frequency_changed = frequency_changed;
            if (signal.change_frequency) {
                signal.change_frequency = false;
                frequency_changed = false;
                signal.frequency_changed = false;
                signal.correct_resample = signal.coarse_freq_offset / rf_frequency;
                rf_frequency += signal.coarse_freq_offset;
                tuner_hz += signal.coarse_freq_offset;
                demodulator.set_tuner(tuner_hz);
                std::fprintf(stderr, "re-tune: %+.1f Hz (total %+.1f)\n", signal.coarse_freq_offset, tuner_hz);
            }
        };
        auto set_gain = [&]() { if (!signal.gain_changed) signal.gain_changed = gain_changed; };     // :178-197, agc off
        auto reset = [&]() {                                              // :135-156
            signal.reset = false;
            rf_frequency = ch_frequency;
            tuner_hz = 0;
            demodulator.set_tuner(0.0);
            signal.coarse_freq_offset = 0.0;
            signal.change_frequency = true;
            signal.correct_resample = 0.0;
            signal.gain_offset = 0;
            signal.change_gain = true;
            set_rf_frequency();
            set_gain();
        };
        reset();
        auto t0 = std::chrono::steady_clock::now();
        size_t pos = 0, timed_from = 0;
        long frames0 = 0, bb0 = 0, ts0 = 0, n_buf = 0;
        for (; pos + (size_t)buf_len <= vi.size(); pos += (size_t)buf_len, ++n_buf) {
            if (n_buf == warm) {                                          // acquisition is behind us: the clock starts here
                demodulator.flush();
                demodulator.deinterleaver->flush();
                ldpc.flush();
                const t2gpu_demod_info w = demodulator.status();
                frames0 = (long)w.frames; bb0 = bbframes; ts0 = ts_bytes; timed_from = pos;
                t0 = std::chrono::steady_clock::now();
            }
            frequency_changed = true;                                     // what rf_changed / gr_changed report with the next packets (:216-223)
            gain_changed = true;
            if (signal.reset) { reset(); continue; }                      // :229-235
            set_rf_frequency();
            set_gain();
            demodulator.execute(buf_len, const_cast<int16_t *>(vi.data()) + pos, const_cast<int16_t *>(vq.data()) + pos, &signal);
        }
        demodulator.flush();                                              // the last symbols' cells reach the de-interleaver before it is flushed
        demodulator.deinterleaver->flush();
        ldpc.flush();                                                     // batches still in the decoder come out (and through BCH / de-framer) inside the clock
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const t2gpu_demod_info st = demodulator.status();
        if (json)
            std::printf("{\"samples\": %zu, \"seconds\": %.6f, \"msamples_per_s\": %.3f, \"buffers\": %ld, \"buf_len\": %d, \"t2_frames\": %ld, "
                        "\"bbframes\": %ld, \"ts_bytes\": %ld, \"symbols\": %ld, \"resets\": %ld, \"deint_start\": %d}\n",
                        pos - timed_from, secs, (pos - timed_from) / secs / 1e6, n_buf - warm, buf_len, (long)st.frames - frames0, bbframes - bb0,
                        ts_bytes - ts0, (long)st.symbols, (long)st.resets, (int)st.deint_start);
        t2::prof_report(stderr);                                          // T2GPU_RX_PROF=1: host time by library call of the stage classes
        std::fprintf(stderr, "%zu samples in %.3f s (%.1f Msamples/s, real time 9.14), %ld symbols, %ld T2 frames, %ld BBFRAMEs, %ld TS bytes\n",
                     pos, secs, pos / secs / 1e6, (long)st.symbols, (long)st.frames, bbframes, ts_bytes);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "t2gpu_rx_file: %s\n", e.what());
        return 1;
    }
    if (file) std::fclose(file);
    if (sock >= 0) close(sock);
    return 0;
}
