// Driver for tests/test_host_mirror_gpu.py: pushes vectors written by the Python test through the C++ stage classes of
// include/t2gpu_stages.hpp (the reference's object / slot / signal shapes over the C ABI) and writes what comes out.
//   stage_mirror_test decim  in.c64 out.c64 chunk
//   stage_mirror_test farrow in.c64 out.c64 chunk resample
//   stage_mirror_test fec    llr.i8 out.u8 fec_type code_rate       (LLR batches of 32 frames -> descrambled BBFRAME bits)
//   stage_mirror_test p1     in.c64 out.txt level
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

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
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
            t2::ldpc_decoder ldpc;
            t2::bch_decoder bch;
            std::vector<uint8_t> out;
            // the reference's connect() chain: ldpc.bit_bch -> bch.execute, bch.bit_descramble -> (here) collect
            ldpc.bit_bch = [&](int *idx, const t2::l1_postsignalling &p, int len, uint8_t *bits) { bch.execute(idx, p, len, bits); };
            bch.bit_descramble = [&](int plp_id, const t2::l1_postsignalling &, int len, uint8_t *bits) {
                out.push_back((uint8_t)plp_id);
                out.insert(out.end(), bits, bits + len);
            };
            int idx_plp_simd[t2::SIZEOF_SIMD] = {};
            const size_t batch = (size_t)fec_size * t2::SIZEOF_SIMD;
            for (size_t pos = 0; pos + batch <= llr.size(); pos += batch) ldpc.execute(idx_plp_simd, l1, (int)batch, llr.data() + pos);
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
        } else return 2;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "stage_mirror_test: %s\n", e.what());
        return 1;
    }
    return 0;
}
