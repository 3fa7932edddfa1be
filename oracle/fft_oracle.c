/*
 * fft_oracle.c -- CPU restatement of the OFDM demodulator's transform for the cpu_baseline leg of bench.py and the CPU tier.
 *
 * TEST INFRASTRUCTURE ONLY (see the other files of oracle/): nothing in sdr_receiver_dvb_t2_amd/ links, loads or calls it.
 *
 * Follows fast_fourier_transform::execute, /root/reference/src/DSP/fast_fourier_transform.h:62-70: a forward complex-to-complex
 * transform of fft_size single-precision cells (the reference plans FFTW3f with FFTW_FORWARD, FFTW_ESTIMATE, :58) followed by the
 * swap of the two halves (:65-68, "fftshift"). FFTW3 is a third-party dependency whose binary the reference ships
 * (Linux_Mint_20/bin/lib/libfftw3f.so.3) and which does not travel to the GPU box; the transform itself is the textbook DFT
 *     X[k] = sum_n x[n] exp(-2 pi i n k / N),
 * evaluated here as a Stockham autosort radix-4 FFT (+ one radix-2 pass when log2 N is odd) in float with twiddles rounded from
 * double. Pinned against tests/golden/fft_golden.npz (outputs of the reference's FFTW binary on committed inputs) to 2e-5 of the
 * spectrum's rms -- the tolerance the HIP kernel is held to; floating-point sums of different orders cannot agree closer.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int n; float *tw; float *scratch; } ora_fft;     /* tw: per radix-4 pass, (w1, w2, w3) per p */

ora_fft *ora_fft_create(int n)
{
    if (n < 2 || (n & (n - 1))) return NULL;
    ora_fft *f = (ora_fft *)calloc(1, sizeof *f);
    f->n = n;
    size_t cnt = 0;
    for (int nc = n; nc >= 4; nc /= 4) cnt += (size_t)(nc / 4) * 6;
    f->tw = (float *)malloc(sizeof(float) * (cnt ? cnt : 1));
    f->scratch = (float *)malloc(sizeof(float) * 2 * (size_t)n * 2);
    float *t = f->tw;
    for (int nc = n; nc >= 4; nc /= 4)
        for (int p = 0; p < nc / 4; p++)
            for (int k = 1; k <= 3; k++) {
                const double a = -2.0 * M_PI * (double)p * k / nc;
                *t++ = (float)cos(a); *t++ = (float)sin(a);
            }
    return f;
}
void ora_fft_destroy(ora_fft *f) { if (f) { free(f->tw); free(f->scratch); free(f); } }

static void pass4(int nc, int s, const float *restrict x, float *restrict y, const float *restrict tw)
{
    const int n1 = nc / 4;
    for (int p = 0; p < n1; p++) {
        const float w1r = tw[6 * p], w1i = tw[6 * p + 1], w2r = tw[6 * p + 2], w2i = tw[6 * p + 3], w3r = tw[6 * p + 4], w3i = tw[6 * p + 5];
        const float *a = x + 2 * (size_t)s * p, *b = a + 2 * (size_t)s * n1, *c = b + 2 * (size_t)s * n1, *d = c + 2 * (size_t)s * n1;
        float *o0 = y + 2 * (size_t)s * 4 * p, *o1 = o0 + 2 * (size_t)s, *o2 = o1 + 2 * (size_t)s, *o3 = o2 + 2 * (size_t)s;
        for (int q = 0; q < s; q++) {
            const float ar = a[2 * q], ai = a[2 * q + 1], br = b[2 * q], bi = b[2 * q + 1];
            const float cr = c[2 * q], ci = c[2 * q + 1], dr = d[2 * q], di = d[2 * q + 1];
            const float apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;
            const float bpdr = br + dr, bpdi = bi + di;
            const float jr = -(bi - di), ji = br - dr;                     /* j (b - d) */
            o0[2 * q] = apcr + bpdr; o0[2 * q + 1] = apci + bpdi;
            const float t1r = amcr - jr, t1i = amci - ji, t2r = apcr - bpdr, t2i = apci - bpdi, t3r = amcr + jr, t3i = amci + ji;
            o1[2 * q] = w1r * t1r - w1i * t1i; o1[2 * q + 1] = w1r * t1i + w1i * t1r;
            o2[2 * q] = w2r * t2r - w2i * t2i; o2[2 * q + 1] = w2r * t2i + w2i * t2r;
            o3[2 * q] = w3r * t3r - w3i * t3i; o3[2 * q + 1] = w3r * t3i + w3i * t3r;
        }
    }
}

/* in, out: interleaved (re, im), n cells each; out receives the transform with its halves swapped when shift != 0 */
int ora_fft_execute(ora_fft *f, const float *in, float *out, int shift)
{
    const int n = f->n;
    float *x = f->scratch, *y = f->scratch + 2 * (size_t)n;
    memcpy(x, in, sizeof(float) * 2 * (size_t)n);
    const float *tw = f->tw;
    int nc = n, s = 1;
    for (; nc >= 4; nc /= 4, s *= 4) {
        pass4(nc, s, x, y, tw);
        tw += (size_t)(nc / 4) * 6;
        float *t = x; x = y; y = t;
    }
    if (nc == 2) {
        for (int q = 0; q < s; q++) {
            const float ar = x[2 * q], ai = x[2 * q + 1], br = x[2 * (q + s)], bi = x[2 * (q + s) + 1];
            y[2 * q] = ar + br; y[2 * q + 1] = ai + bi;
            y[2 * (q + s)] = ar - br; y[2 * (q + s) + 1] = ai - bi;
        }
        float *t = x; x = y; y = t;
    }
    if (shift) {
        memcpy(out, x + n, sizeof(float) * (size_t)n);
        memcpy(out + n, x, sizeof(float) * (size_t)n);
    } else memcpy(out, x, sizeof(float) * 2 * (size_t)n);
    return n;
}
