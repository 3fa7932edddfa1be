// front_plan.cpp -- see front_plan.h. Host-only; built with -ffp-contract=off.
#include "front_plan.h"

#include <cmath>
#include <limits>

namespace {

constexpr float PI_X_2 = 3.14159274101257324219f * 2.0f;          // M_PI_X_2, dvbt2_definition.h:27
constexpr int MAX_RUN = 1 << 24;                                  // keeps k * step exact in double (k < 2^24, step <= 24 bits)

inline int expo(double v) { return v == 0.0 ? std::numeric_limits<int>::min() / 2 : std::ilogb(v); }

// one reference NCO update (dvbt2_demodulator.cpp:187-193); *wrapped is set when a wrap loop ran
inline float nco_next(float v, float fe, bool *wrapped)
{
    float t = v - fe;
    *wrapped = false;
    while (t > PI_X_2) { t -= PI_X_2; *wrapped = true; }
    while (t < -PI_X_2) { t += PI_X_2; *wrapped = true; }
    return t;
}

// largest t in [0, cap] such that lo <= V + k*D < hi holds for every k in [0, t]; -1 if it fails at k = 0
long span(double V, double D, double lo, double hi, long cap)
{
    if (!(V >= lo && V < hi)) return -1;
    if (D == 0.0) return cap;
    double lim = D > 0.0 ? (hi - V) / D : (V - lo) / -D;
    long t = lim >= (double)cap ? cap : (long)lim;
    if (t > cap) t = cap;
    while (t > 0 && !(V + (double)t * D >= lo && V + (double)t * D < hi)) --t;      // division rounding, at most a step or two
    return t;
}

}  // namespace

float t2_wrap_2pi(float a)
{
    while (a > PI_X_2) a -= PI_X_2;
    while (a < -PI_X_2) a += PI_X_2;
    return a;
}

void t2_plan_nco(float &acc, int i_begin, int n, float fe, float phase_nco, std::vector<FrontRun> &runs)
{
    int i = 0;
    float prev = acc;
    bool w;
    while (i < n) {
        const float v0 = nco_next(prev, fe, &w);
        long len = 1;
        double step = 0.0;
        bool w0;
        if (nco_next(v0, fe, &w0) == v0 && !w0) {
            len = n - i;                                                      // the accumulator no longer moves (fe = 0 or below half an ulp)
        } else if (n - i >= 3 && v0 != 0.0f) {
            bool w1, w2;
            const float v1 = nco_next(v0, fe, &w1), v2 = nco_next(v1, fe, &w2);
            const double s1 = (double)v1 - (double)v0, s2 = (double)v2 - (double)v1;
            const int e = expo(v0);
            const bool same = !w1 && !w2 && s1 == s2 && v1 != 0.0f && v2 != 0.0f && expo(v1) == e && expo(v2) == e &&
                              std::signbit(v0) == std::signbit(v1) && std::signbit(v0) == std::signbit(v2);
            if (same) {
                // Inside the binade [2^e, 2^(e+1)) every value is a multiple of u and v - fe rounds to v - R, R = fe rounded to
                // u (a tie is resolved to even and stays even: s1 == s2 establishes that state). Stay one u clear of both
                // ends so the exact difference is in the binade too, and below 2*pi so no wrap loop runs.
                const double u = std::ldexp(1.0, e - 23);
                const double lo = std::ldexp(1.0, e) + u;
                double hi = std::ldexp(1.0, e + 1) - u;
                if (hi > (double)PI_X_2 - u) hi = (double)PI_X_2 - u;
                const double mag = std::fabs((double)v0);
                const double dmag = std::signbit(v0) ? -s1 : s1;              // change of |v| per sample
                long t = span(mag, dmag, lo, hi, MAX_RUN);
                if (t > n - i - 1) t = n - i - 1;
                if (t >= 2) { len = t + 1; step = s1; }
            }
        }
        runs.push_back(FrontRun{i_begin + i, 0, (double)v0, step, 0, phase_nco});
        prev = (float)((double)v0 + (double)(len - 1) * step);
        i += (int)len;
    }
    acc = prev;
}

namespace {

struct FarrowStep {
    int c;                   // outputs of this input sample
    float next;              // x1 after the sample
    bool exact;              // no float operation rounded
    double v[260];           // v[k] = position of output k (k < c), v[c] = first position >= 0.5, v[c + 1] = next
};

// one input sample of interpolator_farrow::operator() (interpolator_farrow.hh:57-63), positions only
bool farrow_step(float x1, float d, FarrowStep &s)
{
    float p = x1;
    s.c = 0;
    s.exact = true;
    s.v[0] = p;
    while (p < 0.5f) {
        if (s.c >= 256) return false;
        const float q = p + d;
        if ((double)q != (double)p + (double)d) s.exact = false;
        p = q;
        s.v[++s.c] = p;
    }
    s.next = p - 1.0f;
    if ((double)s.next != (double)p - 1.0) s.exact = false;
    s.v[s.c + 1] = s.next;
    return true;
}

}  // namespace

long t2_plan_farrow(float &x1, int i_begin, long o_begin, int n, float d, std::vector<FrontRun> &runs)
{
    if (!(d > 1.0f / 256.0f && d < 4.0f)) return -1;
    static thread_local FarrowStep a, b;
    long o = o_begin;
    int i = 0;
    float x = x1;
    int skip = 0, fails = 0;                 // back-off: where runs do not form (a ratio far from 1/2 whose float has an odd
                                             // last bit rounds at almost every sample) the attempt is not repeated at once
    while (i < n) {
        long len = 1;
        double step = 0.0;
        int c = 0;
        float next = 0.0f;
        if (skip == 0 && n - i >= 3) {
            if (!farrow_step(x, d, a)) return -1;
            c = a.c; next = a.next;
            bool ok = a.exact && farrow_step(a.next, d, b) && b.exact && b.c == a.c;
            if (ok) {
                // Two consecutive samples without any rounding and with the same output count: every position moves by
                // D = c*d - 1 per input sample. Position slot k stays representable while it is a multiple of the ulp g_k of
                // the larger of its two binades and does not leave that binade upwards; the count stays c while
                // v[c-1] < 0.5 <= v[c].
                const double D = (double)a.next - (double)x;
                long t = MAX_RUN;
                for (int k = 0; k <= a.c + 1 && t >= 2; ++k) {
                    const double V0 = a.v[k], V1 = b.v[k];
                    if (V1 - V0 != D) { t = 0; break; }
                    const int e = std::max(expo(V0), expo(V1));
                    double lo = -std::numeric_limits<double>::infinity(), hi = std::numeric_limits<double>::infinity();
                    if (e > std::numeric_limits<int>::min() / 4) {
                        const double g = std::ldexp(1.0, e - 23), top = std::ldexp(1.0, e + 1);
                        if (std::fmod(V0, g) != 0.0 || std::fmod(V1, g) != 0.0) { t = 0; break; }
                        lo = -top + g; hi = top;                              // |v| < 2^(e+1)
                    } else if (D != 0.0) { t = 0; break; }                    // both zero yet moving: cannot happen
                    if (k < a.c) { if (hi > 0.5) hi = 0.5; }                   // still an output position
                    else if (k == a.c) { if (lo < 0.5) lo = 0.5; }             // still the exit position
                    const long tk = span(V0, D, lo, hi, t);
                    if (tk < t) t = tk;
                }
                if (t > n - i - 1) t = n - i - 1;
                if (t >= 2) { len = t + 1; step = D; }
            }
            if (len > 1) fails = 0;
            else { if (fails < 6) ++fails; skip = (1 << fails) - 1; }
        } else {
            if (skip > 0) --skip;
            float p = x;                                                       // interpolator_farrow.hh:57-63, positions only
            while (p < 0.5f) { p = p + d; if (++c > 256) return -1; }
            next = p - 1.0f;
        }
        runs.push_back(FrontRun{i_begin + i, (int32_t)o, (double)x, step, c, 0.0f});
        o += (long)c * len;
        if (len > 1) {
            // state after the run: position of its last sample, advanced once with real float operations
            if (!farrow_step((float)((double)x + (double)(len - 1) * step), d, a)) return -1;
            next = a.next;
        }
        x = next;
        i += (int)len;
    }
    x1 = x;
    return o - o_begin;
}
