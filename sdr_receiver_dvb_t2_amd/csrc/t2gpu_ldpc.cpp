// t2gpu_ldpc.cpp -- C-ABI of the LDPC stage (include/t2gpu.h). Host side only: builds the graph, owns the device
// buffers, sizes the persistent grid and launches ldpc_kernel2.hip. There is no CPU decode path in this library.
#include "../../include/t2gpu.h"
#include "ldpc_graph.h"
#include "ldpc_kernel.h"
#include "t2gpu_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

using namespace t2gpu;

struct t2gpu_ldpc {
    LdpcGraph g;
    int device = 0, max_frames = 0, group = T2GPU_SIMD_BATCH, max_trials = T2GPU_LDPC_TRIALS;
    int num_cu = 0;
    LdpcLayerDev *d_layers = nullptr;
    uint32_t *d_layer_words = nullptr;   // the same, four packed dwords per layer (ldpc_kernel.h: LdpcKernelParams::layer_words)
    uint32_t *d_entries = nullptr;
    uint32_t *d_cninfo = nullptr;
    unsigned *d_sync = nullptr;
    unsigned *d_ticket = nullptr;  // batch tickets of the two-frame kernel (ldpc_kernel.h)
    size_t ticket_words = 0;
    size_t sync_words = 0;
    int *d_error = nullptr;
    unsigned *d_resident = nullptr;     // signal memory: workgroups started, cumulative
    unsigned resident_total = 0;        // value d_resident reaches once every workgroup of every launch so far has started
    long long *d_prof = nullptr;   // diagnostics, allocated by t2gpu_ldpc_profile()
    // host-call staging
    int8_t *d_in = nullptr;
    uint8_t *d_out = nullptr;
    int *d_trials = nullptr;
    int last_status = 0;
    // asynchronous host-call form (t2gpu_ldpc_submit / _collect): pinned staging, a stream and an event of the handle's own
    int8_t *p_in = nullptr;
    uint8_t *p_out = nullptr;
    int *p_trials = nullptr;            // [max_frames] verdicts, then the error word
    hipStream_t a_stream = nullptr;
    bool a_stream_masked = false;       // a_stream was made with a CU mask: goes back to the cache instead of being destroyed (submit_stream_release)
    int a_cu_reserve = 32;              // CUs the submit stream's mask leaves to everybody else (t2gpu_ldpc_set_submit_cu_reserve)
    hipEvent_t a_done = nullptr, a_fence = nullptr;
    int a_frames = 0;                   // frames of the pending submit (0: none)
    int s_frames = 0;                   // frames added to the decode being put together (t2gpu_ldpc_submit_add)
    bool s_side = false, s_host = false; // ... of which some came from twins on the device / from host buffers
    bool a_ready = false;               // stream, event and pinned staging of the asynchronous form all exist
    bool plain_launch = false;          // set around the launches of a submit (ldpc_kernel2_launch)
    bool plain_always = false;          // t2gpu_ldpc_set_plain_launch
    int slot_cap = 0;                   // t2gpu_ldpc_set_max_slots: resident batch slots a decode of this handle takes at most (0: all the device holds)
    // the kernel's geometry (ldpc_kernel2.hip: two frames per workgroup)
    int p_blocks_per_cu = 0, p_lds_bytes = 0, p_lds_ctl_offset = 0, p_lds_rec_offset = 0, p_lds_sign_offset = 0, p_lds_ent_offset = 0, p_lds_base = 0,
        p_rec_dwords = 0;
    uint32_t *d_entries2p = nullptr;
    uint32_t *d_state2 = nullptr;
    size_t state2_blocks = 0;
};

// workgroups of one SIMD batch: two frames each, the last one a single frame when the group is odd (ldpc_kernel2.hip)
static int wg_per_batch(const t2gpu_ldpc *h) { return (h->group + 1) / 2; }
static int resident_blocks(const t2gpu_ldpc *h) { return h->num_cu * h->p_blocks_per_cu; }

// ---- what the plain launches of t2gpu_ldpc_submit may hold of a device at one time. The workgroups of a SIMD batch meet at
// every sweep, so every workgroup of every decode in flight must be resident: a cooperative launch has the runtime's word for that, a
// plain one has only this ledger. A submit books the CUs its grid needs (grid / workgroups per CU of its code) against the device's CUs
// and waits -- for decodes already in flight to finish, which they do on their own -- when the device is booked out; a grid that could
// never fit is launched cooperatively instead. Short kernels of other streams only delay a workgroup's start; they do not hold CUs.
namespace {
struct Booking { const t2gpu_ldpc *h; hipEvent_t done; int cus; bool armed; };   // armed: `done` has been recorded behind the decode
std::mutex g_book_m;
std::map<int, std::vector<Booking>> g_books;          // per device

int booked_locked(std::vector<Booking> &b, bool query)
{
    int sum = 0;
    for (size_t i = 0; i < b.size();) {
        // (asking the runtime about every decode in flight on every submit cost the slot-shaped path a quarter of its rate: the events
        // are only looked at when the ledger says the device is full; otherwise a booking goes with its handle's collect / next submit)
        if (query && b[i].armed && hipEventQuery(b[i].done) == hipSuccess) b.erase(b.begin() + (long)i);   // that decode is through: its CUs are free
        else { sum += b[i].cus; ++i; }
    }
    if (query) (void)hipGetLastError();                                                 // hipErrorNotReady is not an error
    return sum;
}
// returns false when the grid cannot fit the device at all (the caller launches cooperatively then)
bool book_cus(const t2gpu_ldpc *h, hipEvent_t done, int cus, int capacity)
{
    if (cus > capacity) return false;
    for (;;) {
        hipEvent_t oldest = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_book_m);
            std::vector<Booking> &b = g_books[h->device];
            for (size_t i = 0; i < b.size();) { if (b[i].h == h) b.erase(b.begin() + (long)i); else ++i; }   // a handle has one decode in flight
            if (booked_locked(b, false) + cus <= capacity || booked_locked(b, true) + cus <= capacity) { b.push_back(Booking{h, done, cus, false}); return true; }
            for (const Booking &x : b) if (x.armed) { oldest = x.done; break; }
        }
        if (oldest) hipEventSynchronize(oldest);     // booked out: until the oldest decode in flight is through
        else std::this_thread::yield();              // (another thread is between its booking and its launch)
    }
}
void arm_booking(const t2gpu_ldpc *h)
{
    std::lock_guard<std::mutex> lk(g_book_m);
    for (Booking &x : g_books[h->device]) if (x.h == h) x.armed = true;
}
void unbook(const t2gpu_ldpc *h)
{
    std::lock_guard<std::mutex> lk(g_book_m);
    auto it = g_books.find(h->device);
    if (it == g_books.end()) return;
    std::vector<Booking> &b = it->second;
    for (size_t i = 0; i < b.size();) { if (b[i].h == h) b.erase(b.begin() + (long)i); else ++i; }
}
// Submit streams made with a CU mask are never destroyed: they go back to a per-device cache and the next handle with the same mask takes
// them from there. hipStreamDestroy of such a stream -- idle, synchronised -- did not return in about one of a thousand processes
// (tools/hang_hunt.py: the process had written all its results and stood in t2gpu_ldpc_destroy's hipStreamDestroy); the same happened to
// t2gpu_rx's streams, which are kept for the same reason (csrc/t2gpu_rx.cpp, StreamBundle).
struct CachedStream { int device, num_cu, reserve; hipStream_t s; };
std::mutex g_submit_m;
std::vector<CachedStream> g_submit_free;
bool submit_stream_cached(t2gpu_ldpc *h)
{
    std::lock_guard<std::mutex> lk(g_submit_m);
    for (size_t k = 0; k < g_submit_free.size(); ++k) {
        const CachedStream &c = g_submit_free[k];
        if (c.device == h->device && c.num_cu == h->num_cu && c.reserve == h->a_cu_reserve) {
            h->a_stream = c.s;
            g_submit_free.erase(g_submit_free.begin() + (long)k);
            return true;
        }
    }
    return false;
}
void submit_stream_release(t2gpu_ldpc *h)
{
    if (!h->a_stream) return;
    if (h->a_stream_masked) {
        std::lock_guard<std::mutex> lk(g_submit_m);
        g_submit_free.push_back(CachedStream{h->device, h->num_cu, h->a_cu_reserve, h->a_stream});
    } else {
        hipStreamDestroy(h->a_stream);
    }
    h->a_stream = nullptr;
}
}  // namespace
static size_t prof_blocks(const t2gpu_ldpc *h) { return h->state2_blocks; }

extern "C" int t2gpu_ldpc_graph_stats(int fec_type, int code_rate, int *links_total, int *layers,
                                      int *levels_total, int *max_cnt)
{
    LdpcGraph g;
    if (!ldpc_build_graph(ldpc_code_id(fec_type, code_rate), g)) { set_error("unknown LDPC code"); return -1; }
    if (links_total) *links_total = g.links_total;
    if (layers) *layers = g.q;
    if (levels_total) *levels_total = g.total_levels;
    if (max_cnt) *max_cnt = g.max_cnt;
    return 0;
}

extern "C" t2gpu_ldpc *t2gpu_ldpc_create(int fec_type, int code_rate, int max_frames, int device)
{
    int id = ldpc_code_id(fec_type, code_rate);
    if (id < 0 || max_frames < 1) { set_error("t2gpu_ldpc_create: bad arguments"); return nullptr; }
    int ndev = 0;
    hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || device < 0 || device >= ndev) {
        set_error(std::string("t2gpu_ldpc_create: no usable HIP device (this library has no CPU path): device ") +
                  std::to_string(device) + " of " + std::to_string(ndev) + ", hipGetDeviceCount: " + hipGetErrorString(de));
        return nullptr;
    }
    t2gpu_ldpc *h = new t2gpu_ldpc();
    h->device = device; h->max_frames = max_frames;
    if (!ldpc_build_graph(id, h->g)) { delete h; set_error("graph build failed"); return nullptr; }
    auto fail = [&](const char *what, hipError_t e) -> t2gpu_ldpc * {
        hip_ok(e, what);
        t2gpu_ldpc_destroy(h);
        return nullptr;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail("hipGetDeviceProperties", e);
    h->num_cu = prop.multiProcessorCount;
    std::vector<LdpcLayerDev> ld(h->g.q);
    for (int i = 0; i < h->g.q; ++i)
        ld[i] = LdpcLayerDev{h->g.layers[i].first_entry, h->g.layers[i].cnt, h->g.layers[i].lmax, h->g.layers[i].n_conflict,
                             h->g.layers[i].kind, h->g.layers[i].step, h->g.layers[i].band,
                             h->g.layers[i].band_prefetch | (h->g.layers[i].no_close << 1)};
    if ((e = hipMalloc(&h->d_layers, ld.size() * sizeof(LdpcLayerDev))) != hipSuccess) return fail("hipMalloc", e);
    {   // the two-frame kernel's form of the same table: four dwords per layer, held in registers (one layer per lane) and read with
        // v_readlane -- no memory access and no wait at the head of a layer (ldpc_kernel2.hip)
        std::vector<uint32_t> lw(4 * (size_t)h->g.q);
        for (int i = 0; i < h->g.q; ++i) {
            const LdpcLayerDev &l = ld[i];
            if (l.cnt > 31 || l.nc > 31 || l.lmax > 511 || l.band > 63 || l.step > 0xffff || l.first_entry > 0xffff || l.kind > 3) {
                set_error("LDPC layer table does not fit its packed form");
                t2gpu_ldpc_destroy(h);
                return nullptr;
            }
            const bool next_generic = i + 1 < h->g.q && ld[i + 1].kind == 2;     // T2_LAYER_GENERIC
            lw[4 * i] = (uint32_t)l.kind | (uint32_t)l.cnt << 2 | (uint32_t)l.nc << 7 | (uint32_t)l.lmax << 12 | (uint32_t)(l.band_prefetch & 1) << 21 |
                        (uint32_t)(l.band_prefetch >> 1) << 22 | (uint32_t)l.band << 23 | (next_generic ? 1u << 29 : 0u);
            lw[4 * i + 1] = (uint32_t)l.step | (uint32_t)l.first_entry << 16;
            lw[4 * i + 2] = h->g.entries[l.first_entry];
            lw[4 * i + 3] = (uint32_t)ld[(i + 1) % h->g.q].first_entry;      // (the last layer names layer 0)
        }
        if ((e = hipMalloc(&h->d_layer_words, lw.size() * 4)) != hipSuccess) return fail("hipMalloc", e);
        if ((e = hipMemcpy(h->d_layer_words, lw.data(), lw.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
    }
    if ((e = hipMalloc(&h->d_entries, h->g.entries.size() * 4)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc(&h->d_cninfo, h->g.cninfo.size() * 4)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMemcpy(h->d_layers, ld.data(), ld.size() * sizeof(LdpcLayerDev), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
    if ((e = hipMemcpy(h->d_entries, h->g.entries.data(), h->g.entries.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
    if ((e = hipMemcpy(h->d_cninfo, h->g.cninfo.data(), h->g.cninfo.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);

    {   // LDS = interleaved LLRs of both frames | control words | 2 x 360 chain records | sign words of both
        // frames | the table entries as (LDS address of bit 0 of the link's run, shift) pairs
        const int n2 = 2 * h->g.n;
        h->p_lds_ctl_offset = (n2 + 15) & ~15;
        h->p_lds_rec_offset = h->p_lds_ctl_offset + 64;
        h->p_lds_sign_offset = h->p_lds_rec_offset + 2 * 360 * 4;
        // the sign words of the parity check; between checks the same room holds the band-walk records of a layer (2 x 360 x 8 B)
        h->p_lds_ent_offset = (h->p_lds_sign_offset + std::max(2 * (h->g.n / 360) * 13 * 4, 2 * 360 * 8) + 7) & ~7;
        h->p_lds_bytes = h->p_lds_ent_offset + (int)h->g.entries.size() * 8;
        h->p_rec_dwords = ldpc_kernel2_record_dwords(h->g.min_cnt, h->g.max_cnt);
        if ((e = ldpc_kernel2_attributes(h->g.min_cnt, h->g.max_cnt, h->p_lds_bytes, &h->p_blocks_per_cu, &h->p_lds_base)) != hipSuccess)
            return fail("kernel attributes", e);
        if (h->p_blocks_per_cu < 1) { set_error("LDPC kernel does not fit a CU"); t2gpu_ldpc_destroy(h); return nullptr; }
        {
            {
                std::vector<uint32_t> e2p(2 * h->g.entries.size());
                for (size_t i = 0; i < h->g.entries.size(); ++i) {
                    const uint32_t base = h->g.entries[i] & 0xffffu, shift = h->g.entries[i] >> 16;
                    e2p[2 * i] = (uint32_t)h->p_lds_base + 2u * (base - shift);      // the address is this + 2 * (j >= shift ? j : j + 360)
                    e2p[2 * i + 1] = shift;
                }
                h->state2_blocks = (size_t)h->num_cu * h->p_blocks_per_cu;
                if ((e = hipMalloc(&h->d_entries2p, e2p.size() * 4)) != hipSuccess) return fail("hipMalloc", e);
                if ((e = hipMemcpy(h->d_entries2p, e2p.data(), e2p.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
                if ((e = hipMalloc(&h->d_state2, h->state2_blocks * h->g.q * 720 * h->p_rec_dwords * 4)) != hipSuccess) return fail("hipMalloc state", e);
            }
        }
    }
    h->sync_words = (size_t)(max_frames + 1) * 64;   // enough for group >= 1 and max_trials <= 63
    if ((e = hipMalloc(&h->d_sync, h->sync_words * 4)) != hipSuccess) return fail("hipMalloc sync", e);
    if ((e = hipMalloc(&h->d_error, 4)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipExtMallocWithFlags(reinterpret_cast<void **>(&h->d_resident), 8, hipMallocSignalMemory)) != hipSuccess) return fail("hipExtMallocWithFlags", e);
    if ((e = hipMemset(h->d_resident, 0, 8)) != hipSuccess) return fail("hipMemset", e);
    return h;
}

extern "C" void t2gpu_ldpc_destroy(t2gpu_ldpc *h)
{
    if (!h) return;
    t2_exit_mark("t2gpu_ldpc_destroy");
    hipFree(h->d_layers); hipFree(h->d_layer_words); hipFree(h->d_entries); hipFree(h->d_cninfo);
    hipFree(h->d_resident); hipFree(h->d_entries2p); hipFree(h->d_state2);
    hipFree(h->d_sync); hipFree(h->d_ticket); hipFree(h->d_error); hipFree(h->d_prof); hipFree(h->d_in); hipFree(h->d_out); hipFree(h->d_trials);
    t2_exit_mark("t2gpu_ldpc_destroy: device memory freed, submit stream next");
    if (h->a_stream) { hipStreamSynchronize(h->a_stream); t2_exit_mark("t2gpu_ldpc_destroy: submit stream idle"); submit_stream_release(h); }
    t2_exit_mark("t2gpu_ldpc_destroy: submit stream gone");
    unbook(h);
    if (h->a_done) hipEventDestroy(h->a_done);
    if (h->a_fence) hipEventDestroy(h->a_fence);
    if (h->d_out) twin_retire_dev(h->d_out, (size_t)h->max_frames * h->g.k);
    hipHostFree(h->p_in); hipHostFree(h->p_out); hipHostFree(h->p_trials);
    delete h;
}

extern "C" int t2gpu_ldpc_configure(t2gpu_ldpc *h, int group, int max_trials)
{
    if (!h || group < 1 || max_trials < 0 || max_trials > 63) { set_error("t2gpu_ldpc_configure: bad arguments"); return -1; }
    if (resident_blocks(h) < (group + 1) / 2) { set_error("device cannot keep one batch resident"); return -1; }
    h->group = group; h->max_trials = max_trials;
    return 0;
}

extern "C" int t2gpu_ldpc_info(const t2gpu_ldpc *h, int *fec_size, int *k_ldpc, int *q_ldpc, int *k_bch)
{
    if (!h) return -1;
    if (fec_size) *fec_size = h->g.n;
    if (k_ldpc) *k_ldpc = h->g.k;
    if (q_ldpc) *q_ldpc = h->g.q;
    if (k_bch) *k_bch = ldpc_k_bch(h->g.id);
    return 0;
}

// What the decode launches occupy on the device: {workgroups resident per CU, wavefronts per workgroup, dynamic LDS bytes per workgroup,
// FEC frames per workgroup, CUs, SIMD batches resident at once}
extern "C" int t2gpu_ldpc_occupancy(const t2gpu_ldpc *h, int *out6)
{
    if (!h || !out6) { set_error("t2gpu_ldpc_occupancy: bad arguments"); return -1; }
    out6[0] = h->p_blocks_per_cu;
    out6[1] = T2GPU_LDPC_THREADS / 64;                                                   // 768 lanes: two per check node of a layer (ldpc_cn3.h)
    out6[2] = h->p_lds_bytes;
    out6[3] = 2;
    out6[4] = h->num_cu;
    out6[5] = resident_blocks(h) / wg_per_batch(h);
    return 0;
}

// Workgroups a decode of n_frames launches (all of them resident for its whole duration), or -1.
extern "C" int t2gpu_ldpc_launch_workgroups(const t2gpu_ldpc *h, int n_frames)
{
    if (!h || n_frames < 1) { set_error("t2gpu_ldpc_launch_workgroups: bad arguments"); return -1; }
    const int nbatches = (n_frames + h->group - 1) / h->group;
    int maxslots = resident_blocks(h) / wg_per_batch(h);
    if (maxslots < 1) return -1;
    if (h->slot_cap >= 1 && h->slot_cap < maxslots) maxslots = h->slot_cap;
    return (nbatches < maxslots ? nbatches : maxslots) * wg_per_batch(h);
}

// n >= 1: a decode of this handle keeps at most n SIMD batches resident (16 workgroups = 16 CUs each for the 64800-bit codes) and hands the
// rest out by ticket as slots come free -- the CUs it leaves alone are where other streams' kernels run beside it (a resident decode workgroup
// fills its CU's LDS); 0: as many as the device holds.
extern "C" int t2gpu_ldpc_set_max_slots(t2gpu_ldpc *h, int n)
{
    if (!h || n < 0) { set_error("t2gpu_ldpc_set_max_slots: bad arguments"); return -1; }
    h->slot_cap = n;
    return 0;
}

// plain != 0: the decodes of this handle are ordinary launches, not cooperative ones. Cooperative launches of different
// streams run one after the other; plain ones run side by side -- the caller then sees to it that what is in flight together fits the
// device (t2gpu_ldpc_launch_workgroups; the frames of a batch meet at every sweep, so their workgroups must all be resident).
extern "C" int t2gpu_ldpc_set_plain_launch(t2gpu_ldpc *h, int plain)
{
    if (!h) { set_error("t2gpu_ldpc_set_plain_launch: null handle"); return -1; }
    h->plain_always = plain != 0;
    return 0;
}

extern "C" int t2gpu_ldpc_execute_dev(t2gpu_ldpc *h, const int8_t *d_llr, int n_frames, uint8_t *d_bits,
                                      int8_t *d_llr_out, int *d_trials_left, void *stream)
{
    if (!h || !d_llr || !d_trials_left || n_frames < 1 || n_frames > h->max_frames) {
        set_error("t2gpu_ldpc_execute_dev: bad arguments");
        return -1;
    }
    T2_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int group = h->group;
    const int nbatches = (n_frames + group - 1) / group;
    const int wg_per_batch = ::wg_per_batch(h);                               // two frames per workgroup
    int maxslots = resident_blocks(h) / wg_per_batch;
    if (maxslots < 1) { set_error("device cannot keep one batch resident"); return -1; }
    if (const char *lim = std::getenv("T2GPU_LDPC_MAX_SLOTS")) {              // experiments: leave part of the device to other streams
        const int v = std::atoi(lim);
        if (v >= 1 && v < maxslots) maxslots = v;
    }
    if (h->slot_cap >= 1 && h->slot_cap < maxslots) maxslots = h->slot_cap;
    int nslots = nbatches < maxslots ? nbatches : maxslots;
    int grid = nslots * wg_per_batch;
    if ((size_t)nbatches * (h->max_trials + 1) > h->sync_words) { set_error("sync scratch too small"); return -1; }

    // One persistent launch walks all batches; with more batches than resident slots they are handed out by ticket, so that slots whose
    // batches stop early take more of them (best when batches take different numbers of sweeps). The ticket words are zeroed with the
    // rendezvous words, by the one clear launch (a hipMemsetAsync is a runtime fill kernel of its own: ~20 us in the decode's stream).
    size_t ticket_words = 0;
    if (nbatches > nslots) {
        ticket_words = 1 + (size_t)nslots * nbatches;
        if (ticket_words > h->ticket_words) {
            T2_HIP(hipStreamSynchronize(s));
            hipFree(h->d_ticket); h->d_ticket = nullptr; h->ticket_words = 0;
            T2_HIP(hipMalloc(&h->d_ticket, ticket_words * 4));
            h->ticket_words = ticket_words;
        }
    }
    T2_HIP(ldpc_clear(reinterpret_cast<unsigned *>(h->d_sync), (size_t)nbatches * (h->max_trials + 1), reinterpret_cast<unsigned *>(h->d_error), 1, s,
                      reinterpret_cast<unsigned *>(h->d_ticket), ticket_words));
    LdpcKernelParams p;
    p.n = h->g.n; p.k = h->g.k; p.q = h->g.q;
    p.layers = h->d_layers; p.layer_words = reinterpret_cast<const uint4 *>(h->d_layer_words); p.entries = h->d_entries; p.entries2 = h->d_entries2p;
    p.lds_base = h->p_lds_base; p.cninfo = h->d_cninfo;
    p.llr = d_llr; p.n_frames = n_frames; p.group = group; p.max_trials = h->max_trials;
    p.bits = d_bits; p.llr_out = d_llr_out; p.trials_left = d_trials_left;
    p.state = reinterpret_cast<uint2 *>(h->d_state2); p.sync = h->d_sync; p.error = h->d_error;
    p.spin_timeout_ticks = 200000000LL;   // 2 s at 100 MHz
    p.ticket = nullptr; p.ticket_rounds = 0;
    p.lds_ctl_offset = h->p_lds_ctl_offset;
    p.lds_rec_offset = h->p_lds_rec_offset;
    p.lds_sign_offset = h->p_lds_sign_offset;
    p.lds_ent_offset = h->p_lds_ent_offset;
    p.n_entries = (int)h->g.entries.size();
    p.prof = h->d_prof;
    p.prof_blocks = (int)prof_blocks(h);
    p.resident = h->d_resident;
    if (h->d_prof) T2_HIP(hipMemsetAsync(h->d_prof, 0, (prof_blocks(h) * 8 + 64) * sizeof(long long), s));
    if (nbatches > nslots) { p.ticket = h->d_ticket; p.ticket_rounds = nbatches; }
    h->resident_total += (unsigned)grid;
    T2_HIP(ldpc_kernel2_launch(h->g.min_cnt, h->g.max_cnt, p, grid, h->p_lds_bytes, s, !(h->plain_launch || h->plain_always)));
    return 0;
}

// The decoder's workgroups are persistent and the 32 of a SIMD batch meet at every sweep: a workgroup that finds no room on
// the device because another stream's kernels got there first stalls its whole batch. A stream that wants to run beside the
// decoder enqueues this wait first: it holds the stream until every workgroup of every decode enqueued so far has started.
extern "C" int t2gpu_ldpc_wait_resident(t2gpu_ldpc *h, void *stream)
{
    if (!h) return -1;
    T2_HIP(hipStreamWaitValue32((hipStream_t)stream, h->d_resident, h->resident_total, hipStreamWaitValueGte, 0xffffffffu));
    return 0;
}

extern "C" int t2gpu_ldpc_profile_layers(t2gpu_ldpc *h, long long *out64)
{
    if (!h || !h->d_prof || !out64) return -1;
    T2_HIP(hipMemcpy(out64, h->d_prof + prof_blocks(h) * 8, 64 * sizeof(long long), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int t2gpu_ldpc_profile(t2gpu_ldpc *h, long long *out8)
{
    if (!h) return -1;
    if (!h->d_prof) {           // first call arms the counters for subsequent launches
        T2_HIP(hipMalloc(&h->d_prof, (prof_blocks(h) * 8 + 64) * sizeof(long long)));
        T2_HIP(hipMemset(h->d_prof, 0, (prof_blocks(h) * 8 + 64) * sizeof(long long)));
    }
    if (out8) {
        const size_t nblk = prof_blocks(h);
        std::vector<long long> v(nblk * 8);
        T2_HIP(hipMemcpy(v.data(), h->d_prof, v.size() * sizeof(long long), hipMemcpyDeviceToHost));
        for (int k = 0; k < 6; ++k) {
            long long sum = 0;
            for (size_t b = 0; b < nblk; ++b) sum += v[b * 8 + k];
            out8[k] = sum;
        }
        // [6] = sum of workgroup lifetimes, [7] = span first start .. last end (100 MHz wall clock ticks)
        long long life = 0, t_min = 0, t_max = 0;
        bool first = true;
        for (size_t b = 0; b < nblk; ++b) {
            long long a = v[b * 8 + 6], z = v[b * 8 + 7];
            if (!a || !z) continue;
            life += z - a;
            if (first || a < t_min) t_min = a;
            if (first || z > t_max) t_max = z;
            first = false;
        }
        out8[6] = life; out8[7] = t_max - t_min;
    }
    return 0;
}

extern "C" int t2gpu_ldpc_status(t2gpu_ldpc *h)
{
    if (!h) return -1;
    int e = 0;
    T2_HIP(hipMemcpy(&e, h->d_error, 4, hipMemcpyDeviceToHost));
    h->last_status = e;
    if (e) set_error(e == 2 ? "LDPC: entry table built for another LDS layout" : "LDPC batch rendezvous timed out");
    return e;
}

extern "C" int t2gpu_ldpc_execute(t2gpu_ldpc *h, const int8_t *in, int len_in, uint8_t *out, int *trials_left)
{
    if (!h || !in || !out || !trials_left || len_in < h->g.n || len_in % h->g.n) {
        set_error("t2gpu_ldpc_execute: bad arguments");
        return -1;
    }
    const int n_frames = len_in / h->g.n;
    if (n_frames > h->max_frames) { set_error("t2gpu_ldpc_execute: more frames than max_frames"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    if (!h->d_in) {
        T2_HIP(hipMalloc(&h->d_in, (size_t)h->max_frames * h->g.n));
        T2_HIP(hipMalloc(&h->d_out, (size_t)h->max_frames * h->g.k));
        T2_HIP(hipMalloc(&h->d_trials, (size_t)h->max_frames * sizeof(int)));
    }
    const int nbatches = (n_frames + h->group - 1) / h->group;
    T2_HIP(hipMemcpy(h->d_in, in, (size_t)len_in, hipMemcpyHostToDevice));
    if (t2gpu_ldpc_execute_dev(h, h->d_in, n_frames, h->d_out, nullptr, h->d_trials, nullptr)) return -1;
    T2_HIP(hipDeviceSynchronize());
    T2_HIP(hipMemcpy(out, h->d_out, (size_t)n_frames * h->g.k, hipMemcpyDeviceToHost));
    T2_HIP(hipMemcpy(trials_left, h->d_trials, (size_t)nbatches * sizeof(int), hipMemcpyDeviceToHost));
    return t2gpu_ldpc_status(h) ? -1 : 0;
}

// ---- the host-buffer slot, split in two (include/t2gpu.h): submit enqueues copy-in, decode and copy-out on the handle's own stream
// and returns; collect waits for (or polls) the result. Several handles in flight keep several SIMD batches on the device at once --
// a caller with the reference's call shape (ldpc_decoder::execute, one batch of 32 per call) otherwise leaves 15/16 of the CUs idle.
// CUs of the device the decodes of t2gpu_ldpc_submit keep off (default 32; 0: none). Before the handle's first submit.
extern "C" int t2gpu_ldpc_set_submit_cu_reserve(t2gpu_ldpc *h, int n_cus)
{
    if (!h || n_cus < 0 || h->a_stream) { set_error("t2gpu_ldpc_set_submit_cu_reserve: bad arguments (or the handle has submitted already)"); return -1; }
    h->a_cu_reserve = n_cus;
    return 0;
}

// One more SIMD batch (or several) into the decode the handle is putting together: its LLRs are on their way into the handle's input behind
// what has been added before; nothing is launched.
extern "C" int t2gpu_ldpc_submit_add(t2gpu_ldpc *h, const int8_t *in, int len_in)
{
    if (!h || !in || len_in < h->g.n || len_in % h->g.n) { set_error("t2gpu_ldpc_submit: bad arguments"); return -1; }
    const int n_frames = len_in / h->g.n;
    if (h->s_frames % h->group) { set_error("t2gpu_ldpc_submit_add: only the last addition may end in a partial SIMD batch"); return -1; }
    if (h->s_frames + n_frames > h->max_frames) { set_error("t2gpu_ldpc_submit: more frames than max_frames"); return -1; }
    if (h->a_frames) { set_error("t2gpu_ldpc_submit: the previous submit has not been collected"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    if (!h->a_ready) {
        // (a_ready only once everything is there: a call that failed half way starts over instead of running on null buffers)
        // Default priority, and everything short that must never sit behind a decode of milliseconds in a shared hardware queue -- the
        // caller's per-symbol launches (t2gpu_demod's streams), the side stream -- at the HIGHEST: the runtime keeps a separate set of
        // hardware queues per priority. (Rounds 4-5 had it the other way round, decodes at the lowest priority under a default-priority
        // null stream: with waves of a lowest-priority queue resident, every launch of any other queue took ~45 us instead of ~6
        // (tools/small_kernel_beside_submits.py) -- half of the slot-shaped path's time while a frame's batches were being decoded.)
        // ... and on a CU mask that leaves the device's first a_cu_reserve CUs alone (t2gpu_ldpc_set_submit_cu_reserve; 32 of 256 by default):
        // with the decodes' workgroups -- twelve wavefronts and 155 KB of LDS for a millisecond each -- free to settle on every CU, EVERY
        // launch of every other stream of the process took ~45 us longer while three or more decodes were resident (a 4 us scatter: 45 us;
        // the slot-shaped path's symbols 110 - 380 us instead of 55 for a third of each frame; profiles/HISTORY.md, round 5). With a few CUs
        // the decodes never touch, the short launches are short again.
        if (const char *e = std::getenv("T2GPU_LDPC_QUEUE_SKIP"))        // EXPERIMENT: hardware queues made (and left idle) ahead of the submit stream's
            for (int k = 0; k < std::atoi(e) && !h->a_stream; ++k) {
                hipStream_t dummy = nullptr;
                uint32_t all[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
                if (hipExtStreamCreateWithCUMask(&dummy, 8, all) != hipSuccess) (void)hipGetLastError();
            }
        if (!h->a_stream && h->a_cu_reserve > 0 && h->num_cu - h->a_cu_reserve >= 16 && h->num_cu <= 1024) {
            uint32_t mask[32] = {};
            for (int c = h->a_cu_reserve; c < h->num_cu; ++c) mask[c >> 5] |= 1u << (c & 31);
            if (!submit_stream_cached(h) &&
                hipExtStreamCreateWithCUMask(&h->a_stream, (uint32_t)((h->num_cu + 31) / 32), mask) != hipSuccess) { (void)hipGetLastError(); h->a_stream = nullptr; h->a_cu_reserve = 0; }
            h->a_stream_masked = h->a_stream != nullptr;
        } else if (!h->a_stream) {
            h->a_cu_reserve = 0;
        }
        if (!h->a_stream) T2_HIP(hipStreamCreateWithFlags(&h->a_stream, hipStreamNonBlocking));
        if (!h->a_done) T2_HIP(hipEventCreateWithFlags(&h->a_done, hipEventDisableTiming));
        if (!h->p_in) T2_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->p_in), (size_t)h->max_frames * h->g.n, hipHostMallocDefault));
        if (!h->p_out) T2_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->p_out), (size_t)h->max_frames * h->g.k, hipHostMallocDefault));
        if (!h->p_trials) T2_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->p_trials), ((size_t)h->max_frames + 1) * sizeof(int), hipHostMallocDefault));
        h->a_ready = true;
    }
    if (!h->d_in) T2_HIP(hipMalloc(&h->d_in, (size_t)h->max_frames * h->g.n));
    if (!h->d_out) T2_HIP(hipMalloc(&h->d_out, (size_t)h->max_frames * h->g.k));
    if (!h->d_trials) T2_HIP(hipMalloc(&h->d_trials, (size_t)h->max_frames * sizeof(int)));
    if (h->s_frames == 0) {
        twin_retire_dev(h->d_out, (size_t)h->max_frames * h->g.k);  // the previous result's bits are about to be overwritten
        h->s_side = false; h->s_host = false;
    }
    const size_t at = (size_t)h->s_frames * h->g.n;
    const void *twin = twin_lookup(in, (size_t)len_in, h->device);
    if (h->s_frames && (twin ? h->s_host : h->s_side)) {
        set_error("t2gpu_ldpc_submit_add: batches with and without a device twin in one decode (launch what has been added first)");
        return -1;
    }
    if (twin) {
        // a SIMD batch assembled from the demapper's output (t2gpu_twin_copy) is on the device already: its twin is written on the
        // device's side stream, so the copy into this handle's input is put there too (in order with those writes, and the twin is
        // free for the next batch when it is through) and this handle's stream starts behind it. Nothing on the side stream (or the
        // null stream) ever waits for this handle's stream: a wait the other way round held the caller's per-symbol kernels up behind
        // whole decodes whenever two handles' streams shared a hardware queue.
        hipStream_t side = side_stream(h->device);
        if (!side) return -1;
        T2_HIP(hipMemcpyAsync(h->d_in + at, twin, (size_t)len_in, hipMemcpyDeviceToDevice, side));
        h->s_side = true;
    } else {
        std::memcpy(h->p_in + at, in, (size_t)len_in);              // the caller's buffer is free again when this returns
        h->s_host = true;
    }
    h->s_frames += n_frames;
    return 0;
}

// ... and the decode of everything added, as ONE launch
extern "C" int t2gpu_ldpc_submit_go(t2gpu_ldpc *h)
{
    if (!h || !h->s_frames || !h->a_ready) { set_error("t2gpu_ldpc_submit_go: nothing has been added"); return -1; }
    T2_HIP(hipSetDevice(h->device));
    const int n_frames = h->s_frames;
    const int nbatches = (n_frames + h->group - 1) / h->group;
    hipStream_t s = h->a_stream;
    const int8_t *d_llr = h->d_in;
    if (h->s_host) d_llr = h->p_in;                                 // the decoder reads every LLR once: straight from the page-locked copy
    if (h->s_side) {
        hipStream_t side = side_stream(h->device);
        if (!side) return -1;
        if (!h->a_fence) T2_HIP(hipEventCreateWithFlags(&h->a_fence, hipEventDisableTiming));
        T2_HIP(hipEventRecord(h->a_fence, side));
        T2_HIP(hipStreamWaitEvent(s, h->a_fence, 0));
    }
    h->s_frames = 0;
    // several submits run side by side (cooperative launches would not). What they hold of the device together is booked (above): this
    // call waits while the decodes in flight leave no room for its grid. The grid itself is sized for the CUs this stream can reach: its
    // CU mask leaves a_cu_reserve CUs out, and a ticketed persistent grid larger than what fits there would have workgroups that are
    // never dispatched while the resident ones wait for them at the batch rendezvous (the cooperative launch's occupancy check
    // counts the whole device, not the mask). So the resident slots are capped to whole batches inside the mask -- the other batches are
    // handed out by ticket as slots come free -- and the launch is always a plain one.
    const int per_cu = h->p_blocks_per_cu;
    const int reach = (h->num_cu - h->a_cu_reserve) * per_cu / wg_per_batch(h);
    if (per_cu < 1 || reach < 1) { set_error("t2gpu_ldpc_submit: the CUs this stream can reach cannot keep one batch resident"); return -1; }
    const int cap_was = h->slot_cap;
    h->slot_cap = cap_was >= 1 && cap_was < reach ? cap_was : reach;
    const int wgs = t2gpu_ldpc_launch_workgroups(h, n_frames);
    int rc = -1;
    if (wgs >= 1 && book_cus(h, h->a_done, (wgs + per_cu - 1) / per_cu, h->num_cu - h->a_cu_reserve)) {
        h->plain_launch = true;
        rc = t2gpu_ldpc_execute_dev(h, d_llr, n_frames, h->d_out, nullptr, h->d_trials, s);
        h->plain_launch = false;
    } else {
        set_error("t2gpu_ldpc_submit: the decode's grid does not fit the CUs of its stream");
    }
    h->slot_cap = cap_was;
    if (rc) { unbook(h); return -1; }
    // results to the page-locked staging by a kernel of this stream, not by the copy engine (ldpc_kernel2.hip: one in-order copy queue
    // for all streams made the decodes of different handles wait for each other)
    T2_HIP(ldpc_results_to_host(h->d_out, (size_t)n_frames * h->g.k, h->d_trials, nbatches, h->d_error, h->p_out, h->p_trials, h->p_trials + h->max_frames, s));
    T2_HIP(hipEventRecord(h->a_done, s));
    arm_booking(h);
    h->a_frames = n_frames;
    return 0;
}

extern "C" int t2gpu_ldpc_submit(t2gpu_ldpc *h, const int8_t *in, int len_in)
{
    if (h && h->s_frames) { set_error("t2gpu_ldpc_submit: a decode is being put together (t2gpu_ldpc_submit_add)"); return -1; }
    if (t2gpu_ldpc_submit_add(h, in, len_in) != 0) return -1;
    return t2gpu_ldpc_submit_go(h);
}

// wait != 0: blocks until the pending submit is through; wait == 0: returns 1 at once when it is not. On 0, *out points at
// a_frames * k_ldpc information bits (one per byte) and *trials_left at the per-batch verdicts, both in the handle's pinned staging:
// valid until the next submit on this handle.
extern "C" int t2gpu_ldpc_collect(t2gpu_ldpc *h, int wait, const uint8_t **out, const int **trials_left, int *n_frames)
{
    if (!h || !out || !trials_left) { set_error("t2gpu_ldpc_collect: bad arguments"); return -1; }
    if (!h->a_frames) { set_error("t2gpu_ldpc_collect: nothing submitted"); return -1; }
    if (wait) T2_HIP(hipEventSynchronize(h->a_done));
    else {
        const hipError_t e = hipEventQuery(h->a_done);
        if (e == hipErrorNotReady) return 1;
        T2_HIP(e);
    }
    *out = h->p_out; *trials_left = h->p_trials;
    unbook(h);                                                       // the decode is through: its CUs are free
    twin_publish(h->p_out, h->d_out, (size_t)h->a_frames * h->g.k, h->device, false);   // bch_decoder is handed these bits next
    if (n_frames) *n_frames = h->a_frames;
    h->a_frames = 0;
    h->last_status = h->p_trials[h->max_frames];
    if (h->last_status) { set_error("LDPC batch rendezvous timed out"); return -1; }
    return 0;
}
