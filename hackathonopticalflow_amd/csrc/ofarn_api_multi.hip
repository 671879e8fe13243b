// ofarn_api_multi.hip -- multi-GPU in ONE process at the C-ABI (include/ofarn.h, ofarn_multi_*), as SURVEY.md 8(e) lays it out:
// frame pairs are independent units (DenseOF.py:519-525 has no state beyond prev_gray), so device g of G owns the contiguous pair
// range shard_pairs(n, g, G); one host thread + one HIP stream + one ofarn_ctx per device run the shards; the only exchange is
// ONE all-gather of the per-pair danger maps (uint8 mask + uint8 V per grid point; 64 x 2304 x 2 B = 295 KB per device at batch
// 512 over 8 GPUs) over RCCL (ncclCommInitAll over the local devices, ncclAllGather inside one ncclGroup).  Flow fields are
// never gathered (8.5 GB per 512 pairs): each stays on, or is copied to the host from, the device that made it.
//
// librccl is opened with dlopen when the first ofarn_multi is created: libofarn.so has no link-time dependency on it (single-GPU
// users never load it), and a process that already holds an RCCL (PyTorch bundles one) keeps exactly that one.
#include "ofarn_host.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <memory>
#include <functional>
#include <mutex>
#include <thread>

using namespace ofarn;
using namespace ofarn_host;

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string error;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.error = std::string("librccl could not be opened: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n; return p; };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    });
    return r;
}

}  // namespace

// One persistent host thread per device, created by ofarn_multi_create and joined by ofarn_multi_destroy (round 3 created and
// joined 2 x G std::threads per call and ran the n = 1 case on the caller's thread, which left the caller's current device
// changed).  A worker sets its device once; everything that touches a device -- phases 1 and 3 of a batch, the RCCL group (on
// worker 0), creation, destruction, synchronisation -- runs on workers, so the calling thread's current device is never touched.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = false, quit = false;
    int device = 0;
    bool device_ok = false;
};

struct ofarn_multi {
    std::vector<std::unique_ptr<Worker>> workers;
    int n = 0;
    int max_w = 0, max_h = 0, max_batch = 0;
    ofarn_params prm{};
    std::vector<int> dev;
    std::vector<ofarn_ctx *> ctx;
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;
    // per device: frames of the shard, its flow, the padded gather buffers and the compact gathered maps
    struct Buf {
        uint8_t *frames = nullptr, *gmask = nullptr, *gv = nullptr, *mask_all = nullptr, *v_all = nullptr;
        float *flow = nullptr;
        size_t frames_cap = 0, flow_cap = 0, gmask_cap = 0, gv_cap = 0, mask_all_cap = 0, v_all_cap = 0;
    };
    std::vector<Buf> buf;
    unsigned long long gathers = 0;     // ncclAllGather calls issued so far (two per device and batch)
    // OFARN_MULTI_LOOPBACK=1 (tests): the listed devices may repeat -- several ranks on ONE GPU, each with its own context, worker and
    // stream -- and the all-gather is replaced by the copies it stands for (every rank fetches every rank's block with
    // hipMemcpyAsync behind an event of the producing rank).  RCCL cannot put two ranks on one device, and the GPU box has one GPU:
    // this is how the G > 1 branches (shards, in-place offsets, the `even` shortcut, compaction, host copies, the worker threads)
    // run on hardware at all.  Never set in production.
    bool loopback = false;
    std::vector<hipEvent_t> ev_maps;    // loopback: rank r's danger maps are written
    double last_ms = 0;
};

namespace {

template <typename T>
int grow(T **p, size_t *cap, size_t need, const char *what)
{
    if (need <= *cap && *p) return OFARN_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    if (need < 1) need = 1;
    if (hipMalloc((void **)p, need + 256) != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        return fail(OFARN_E_NOMEM, "%s of %zu bytes does not fit", what, need);
    }
    *cap = need;
    return OFARN_OK;
}

// contiguous, balanced: the first n % world ranks get one pair more (the Python mirror's shard_pairs)
void shard(int n_pairs, int rank, int world, int &start, int &count)
{
    if (n_pairs < 0) n_pairs = 0;
    const int base = n_pairs / world, rem = n_pairs % world;
    start = rank * base + (rank < rem ? rank : rem);
    count = base + (rank < rem ? 1 : 0);
}

// Every index a multi-device batch uses, in one host-only place (ofarn_gather_plan exports it; tests/test_host_abi.py checks it on
// the CPU against the Python mirror for ragged and even pair counts, 1 ... 8 devices -- the paths a one-GPU box cannot run):
//   rank r owns pairs [start[r], start[r] + count[r]);
//   cap = the largest shard: every rank contributes cap rows of P bytes to the all-gather, its own at row r * cap of a padded
//         array u8[world][cap][P] (send buffer = receive buffer + gather_off[r]: in place);
//   even = every shard is full, so the padded array IS the global u8[n_pairs][P] array and nothing is compacted;
//   otherwise rank r's count[r] rows move from gather_off[r] to global_off[r] = start[r] * P of the compact array.
struct GatherPlan {
    int world = 0, n_pairs = 0, P = 0, cap = 0;
    bool even = true;
    std::vector<int> start, count;
    std::vector<size_t> gather_off, global_off;
    size_t padded_bytes() const { return (size_t)world * cap * P; }
    size_t rank_bytes() const { return (size_t)cap * P; }
};

GatherPlan gather_plan(int n_pairs, int world, int P)
{
    GatherPlan g;
    g.world = world; g.n_pairs = n_pairs < 0 ? 0 : n_pairs; g.P = P < 0 ? 0 : P;
    g.start.resize(world); g.count.resize(world); g.gather_off.resize(world); g.global_off.resize(world);
    for (int r = 0; r < world; r++) { shard(g.n_pairs, r, world, g.start[r], g.count[r]); g.cap = std::max(g.cap, g.count[r]); }
    g.even = g.n_pairs == g.cap * world;
    for (int r = 0; r < world; r++) {
        g.gather_off[r] = (size_t)r * g.cap * g.P;
        g.global_off[r] = (size_t)g.start[r] * g.P;
    }
    return g;
}

void worker_main(Worker *w)
{
    w->device_ok = hipSetDevice(w->device) == hipSuccess;
    if (!w->device_ok) (void)hipGetLastError();
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->has_job) {
            std::function<void()> job = std::move(w->job);
            w->has_job = false;
            lk.unlock();
            job();
            lk.lock();
            w->done = true;
            w->cv.notify_all();
            continue;
        }
        if (w->quit) return;
    }
}

void post(Worker *w, std::function<void()> job)
{
    std::lock_guard<std::mutex> lk(w->mu);
    w->job = std::move(job);
    w->has_job = true;
    w->done = false;
    w->cv.notify_all();
}

void wait_done(Worker *w)
{
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->done; });
    w->done = false;
}

// Runs fn(g) on device g's worker thread for g in [first, last); returns the first failing rank's code and leaves its message for
// ofarn_last_error() on the calling thread.
template <typename F>
int on_devices(ofarn_multi *m, int first, int last, F &&fn)
{
    std::vector<int> rc(m->n, OFARN_OK);
    std::vector<std::string> msg(m->n);
    for (int g = first; g < last; g++)
        post(m->workers[g].get(), [&, g] {
            if (!m->workers[g]->device_ok) { rc[g] = OFARN_E_HIP; msg[g] = "hipSetDevice failed"; return; }
            rc[g] = fn(g);
            if (rc[g]) msg[g] = ofarn_last_error();          // thread-local: carry it over to the calling thread
        });
    for (int g = first; g < last; g++) wait_done(m->workers[g].get());
    for (int g = first; g < last; g++)
        if (rc[g]) return fail(rc[g], "device %d (rank %d): %s", m->dev[g], g, msg[g].c_str());
    return OFARN_OK;
}
template <typename F> int on_all_devices(ofarn_multi *m, F &&fn) { return on_devices(m, 0, m->n, fn); }

// The one collective: after every device has written its shard's maps at rank * cap rows of its own gather buffers, all-gather them
// in place (send buffer = recv buffer + rank * cap * P), all devices inside one group.
int gather_maps(ofarn_multi *m, const GatherPlan &gp, const std::vector<uint8_t *> &gmask, const std::vector<uint8_t *> &gv)
{
    const size_t cnt = gp.rank_bytes();
    if (m->loopback) {
        // what the all-gather does, spelled out: after rank r's maps are complete (its event), every rank g copies rank r's block
        // into its own padded array at the same offset
        for (int r = 0; r < m->n; r++) HIP_TRY(hipEventRecord(m->ev_maps[r], m->stream[r]));
        for (int g = 0; g < m->n; g++)
            for (int r = 0; r < m->n; r++) {
                if (r == g) continue;
                HIP_TRY(hipStreamWaitEvent(m->stream[g], m->ev_maps[r], 0));
                HIP_TRY(hipMemcpyAsync(gmask[g] + gp.gather_off[r], gmask[r] + gp.gather_off[r], cnt, hipMemcpyDeviceToDevice, m->stream[g]));
                HIP_TRY(hipMemcpyAsync(gv[g] + gp.gather_off[r], gv[r] + gp.gather_off[r], cnt, hipMemcpyDeviceToDevice, m->stream[g]));
            }
        m->gathers += 2ull * m->n;
        return OFARN_OK;
    }
    Rccl &R = rccl();
    ncclResult_t r = R.GroupStart();
    for (int g = 0; g < m->n && r == ncclSuccess; g++) {
        r = R.AllGather(gmask[g] + gp.gather_off[g], gmask[g], cnt, ncclUint8, m->comm[g], m->stream[g]);
        if (r == ncclSuccess) r = R.AllGather(gv[g] + gp.gather_off[g], gv[g], cnt, ncclUint8, m->comm[g], m->stream[g]);
        m->gathers += 2;
    }
    const ncclResult_t e = R.GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return fail(OFARN_E_HIP, "RCCL all-gather of the danger maps failed: %s", R.GetErrorString(r));
    return OFARN_OK;
}

int multi_calc(ofarn_multi *m, const uint8_t *h_frames, const uint8_t *const *d_frames, int n_frames, int n_pairs_dev, int w, int h,
               int pairs_mode, float *h_flow, float *const *d_flow, uint8_t *h_mask, uint8_t *h_v, uint8_t *const *d_mask_all,
               uint8_t *const *d_v_all)
{
    if (!m) return fail(OFARN_E_INVALID, "multi is NULL");
    if (w < 1 || h < 1 || (size_t)w * h > (size_t)m->max_w * m->max_h)
        return fail(OFARN_E_SIZE, "frame %dx%d does not fit the contexts' %dx%d", w, h, m->max_w, m->max_h);
    if (pairs_mode != OFARN_PAIRS_INDEPENDENT && pairs_mode != OFARN_PAIRS_CONSECUTIVE)
        return fail(OFARN_E_INVALID, "pairs_mode must be 0 or 1");
    const bool host = d_frames == nullptr;
    int n_pairs;
    if (host) {
        if (!h_frames) return fail(OFARN_E_INVALID, "frames is NULL");
        n_pairs = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? n_frames - 1 : n_frames / 2;
        if (n_pairs < 0 || (pairs_mode == OFARN_PAIRS_INDEPENDENT && (n_frames & 1)))
            return fail(OFARN_E_INVALID, "n_frames=%d does not form whole pairs in mode %d", n_frames, pairs_mode);
    } else n_pairs = n_pairs_dev;
    if (n_pairs < 0) return fail(OFARN_E_INVALID, "n_pairs < 0");
    if (n_pairs == 0) return OFARN_OK;
    const bool want_maps = host ? (h_mask || h_v) : (d_mask_all || d_v_all);
    if (host ? ((h_mask == nullptr) != (h_v == nullptr)) : ((d_mask_all == nullptr) != (d_v_all == nullptr)))
        return fail(OFARN_E_INVALID, "danger mask and v must be given together");
    if ((m->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW) && (host ? !h_flow : !d_flow))
        return fail(OFARN_E_INVALID, "OPTFLOW_USE_INITIAL_FLOW needs the flow buffers (they hold the initial flows)");
    const int G = m->n;
    const int P = ofarn_grid_points(w, h, m->prm.grid_step, nullptr);
    if (P < 0) return P;
    const size_t fsz = (size_t)w * h;
    const int fstep = pairs_mode == OFARN_PAIRS_CONSECUTIVE ? 1 : 2;
    const GatherPlan gp = gather_plan(n_pairs, G, P);
    const std::vector<int> &start = gp.start, &count = gp.count;
    const bool even = gp.even;                 // every shard full: the gather buffer IS the global array
    const bool gather = want_maps && P > 0;
    const bool use_init = (m->prm.flags & OFARN_FLAG_USE_INITIAL_FLOW) != 0;
    std::vector<uint8_t *> gmask(G, nullptr), gv(G, nullptr);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // phase 1, one host thread per device: buffers, upload of the shard (host variant), the shard's flow + danger maps enqueued
    int rc = on_all_devices(m, [&](int g) -> int {
        ofarn_multi::Buf &b = m->buf[g];
        int r;
        if (gather) {
            const size_t gbytes = gp.padded_bytes();
            uint8_t *user_m = host ? nullptr : d_mask_all[g], *user_v = host ? nullptr : d_v_all[g];
            if (even && user_m) { gmask[g] = user_m; gv[g] = user_v; }           // gather straight into the caller's arrays
            else {
                if ((r = grow(&b.gmask, &b.gmask_cap, gbytes, "gather buffer")) || (r = grow(&b.gv, &b.gv_cap, gbytes, "gather buffer"))) return r;
                gmask[g] = b.gmask; gv[g] = b.gv;
            }
        }
        const int np = count[g];
        const int nf = np == 0 ? 0 : (pairs_mode == OFARN_PAIRS_CONSECUTIVE ? np + 1 : 2 * np);
        const uint8_t *frames_g;
        float *flow_g;
        if (host) {
            if ((r = grow(&b.frames, &b.frames_cap, (size_t)nf * fsz, "frame shard"))) return r;
            if (h_flow && (r = grow(&b.flow, &b.flow_cap, (size_t)np * fsz * 2 * sizeof(float), "flow shard"))) return r;
            if (nf) HIP_TRY(hipMemcpyAsync(b.frames, h_frames + (size_t)start[g] * fstep * fsz, (size_t)nf * fsz, hipMemcpyHostToDevice, m->stream[g]));
            if (use_init && np)
                HIP_TRY(hipMemcpyAsync(b.flow, h_flow + (size_t)start[g] * fsz * 2, (size_t)np * fsz * 2 * sizeof(float), hipMemcpyHostToDevice, m->stream[g]));
            frames_g = b.frames;
            flow_g = h_flow ? b.flow : nullptr;
        } else {
            frames_g = d_frames[g];
            flow_g = d_flow ? d_flow[g] : nullptr;
            if (np && !frames_g) return fail(OFARN_E_INVALID, "frames of device %d is NULL", g);
        }
        if (g == 0) {
            if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) return fail(OFARN_E_HIP, "event creation failed");
            HIP_TRY(hipEventRecord(ev0, m->stream[0]));
        }
        if (np) {
            uint8_t *mk = gather ? gmask[g] + gp.gather_off[g] : nullptr, *vv = gather ? gv[g] + gp.gather_off[g] : nullptr;
            if ((r = ofarn_calc_batch_device(m->ctx[g], frames_g, nf, w, h, pairs_mode, flow_g, mk, vv, m->stream[g]))) return r;
        }
        return OFARN_OK;
    });

    // phase 2, worker 0: the one collective, all devices in one RCCL group
    if (!rc && gather) rc = on_devices(m, 0, 1, [&](int) -> int { return gather_maps(m, gp, gmask, gv); });

    // phase 3, one host thread per device again: compaction of ragged shards, copies back to the host, synchronisation
    const int rc3 = on_all_devices(m, [&](int g) -> int {
        ofarn_multi::Buf &b = m->buf[g];
        if (!rc && gather) {
            uint8_t *out_m = nullptr, *out_v = nullptr;
            if (!host) { out_m = d_mask_all[g]; out_v = d_v_all[g]; }
            else if (g == 0 && !even) {
                int r;
                if ((r = grow(&b.mask_all, &b.mask_all_cap, (size_t)n_pairs * P, "gathered maps")) ||
                    (r = grow(&b.v_all, &b.v_all_cap, (size_t)n_pairs * P, "gathered maps")))
                    return r;
                out_m = b.mask_all; out_v = b.v_all;
            }
            if (out_m && out_m != gmask[g])
                for (int r = 0; r < G; r++) {
                    if (!count[r]) continue;
                    HIP_TRY(hipMemcpyAsync(out_m + gp.global_off[r], gmask[g] + gp.gather_off[r], (size_t)count[r] * P, hipMemcpyDeviceToDevice, m->stream[g]));
                    HIP_TRY(hipMemcpyAsync(out_v + gp.global_off[r], gv[g] + gp.gather_off[r], (size_t)count[r] * P, hipMemcpyDeviceToDevice, m->stream[g]));
                }
            if (host && g == 0) {
                const uint8_t *sm = even ? gmask[0] : b.mask_all, *sv = even ? gv[0] : b.v_all;
                HIP_TRY(hipMemcpyAsync(h_mask, sm, (size_t)n_pairs * P, hipMemcpyDeviceToHost, m->stream[0]));
                HIP_TRY(hipMemcpyAsync(h_v, sv, (size_t)n_pairs * P, hipMemcpyDeviceToHost, m->stream[0]));
            }
        }
        if (g == 0 && ev1) (void)hipEventRecord(ev1, m->stream[0]);
        if (!rc && host && h_flow && count[g])
            HIP_TRY(hipMemcpyAsync(h_flow + (size_t)start[g] * fsz * 2, b.flow, (size_t)count[g] * fsz * 2 * sizeof(float), hipMemcpyDeviceToHost, m->stream[g]));
        // the host variant is synchronous; the device variant returns with everything enqueued on the devices' streams -- and
        // synchronises too when an earlier phase failed, so that nothing is left running on buffers the caller may free
        if (host || rc) HIP_TRY(hipStreamSynchronize(m->stream[g]));
        if (g == 0) {
            if (ev0 && ev1 && host && !rc) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) m->last_ms = ms; else (void)hipGetLastError();
            }
            if (ev0) (void)hipEventDestroy(ev0);
            if (ev1) (void)hipEventDestroy(ev1);
        }
        return OFARN_OK;
    });
    return rc ? rc : rc3;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int ofarn_shard_pairs(int n_pairs, int rank, int world, int *start, int *count)
{
    if (world < 1 || rank < 0 || rank >= world) return fail(OFARN_E_INVALID, "bad rank/world %d/%d", rank, world);
    int s, c;
    shard(n_pairs, rank, world, s, c);
    if (start) *start = s;
    if (count) *count = c;
    return OFARN_OK;
}

int ofarn_gather_plan(int n_pairs, int world, int P, int *start, int *count, int *cap, int *even, uint64_t *gather_off,
                      uint64_t *global_off)
{
    if (world < 1 || n_pairs < 0 || P < 0) return fail(OFARN_E_INVALID, "bad n_pairs/world/P %d/%d/%d", n_pairs, world, P);
    const GatherPlan g = gather_plan(n_pairs, world, P);
    for (int r = 0; r < world; r++) {
        if (start) start[r] = g.start[r];
        if (count) count[r] = g.count[r];
        if (gather_off) gather_off[r] = g.gather_off[r];
        if (global_off) global_off[r] = g.global_off[r];
    }
    if (cap) *cap = g.cap;
    if (even) *even = g.even ? 1 : 0;
    return OFARN_OK;
}

int ofarn_multi_create(const ofarn_params *params, const int *devices, int n_devices, int max_w, int max_h, int max_batch_per_device,
                       ofarn_multi **out)
{
    if (!out) return fail(OFARN_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!params) return fail(OFARN_E_INVALID, "params is NULL");
    if (n_devices < 1 || n_devices > 64) return fail(OFARN_E_INVALID, "n_devices must be in [1, 64], got %d", n_devices);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(OFARN_E_HIP, "no HIP device visible; libofarn has no CPU path");
    const char *lb = getenv("OFARN_MULTI_LOOPBACK");
    const bool loopback = lb && lb[0] == '1';
    std::vector<int> dev(n_devices);
    for (int g = 0; g < n_devices; g++) {
        dev[g] = devices ? devices[g] : (loopback ? g % ndev : g);
        if (dev[g] < 0 || dev[g] >= ndev) return fail(OFARN_E_INVALID, "device %d out of range [0, %d)", dev[g], ndev);
        for (int q = 0; q < g && !loopback; q++)
            if (dev[q] == dev[g]) return fail(OFARN_E_INVALID, "device %d listed twice (one rank per GPU)", dev[g]);
    }
    Rccl &R = rccl();
    if (!loopback && !R.error.empty()) return fail(OFARN_E_HIP, "%s", R.error.c_str());
    ofarn_multi *m = new ofarn_multi();
    m->n = n_devices;
    m->dev = dev;
    m->prm = *params;
    m->max_w = max_w; m->max_h = max_h; m->max_batch = max_batch_per_device;
    m->loopback = loopback;
    m->ev_maps.assign(n_devices, nullptr);
    m->ctx.assign(n_devices, nullptr);
    m->stream.assign(n_devices, nullptr);
    m->buf.resize(n_devices);
    for (int g = 0; g < n_devices; g++) {
        m->workers.emplace_back(new Worker());
        m->workers[g]->device = dev[g];
        m->workers[g]->th = std::thread(worker_main, m->workers[g].get());
    }
    int rc = on_all_devices(m, [&](int g) -> int {
        int r = ofarn_create(params, dev[g], max_w, max_h, max_batch_per_device, &m->ctx[g]);
        if (!r && hipStreamCreateWithFlags(&m->stream[g], hipStreamNonBlocking) != hipSuccess)
            r = fail(OFARN_E_HIP, "stream creation on device %d failed", dev[g]);
        if (!r && loopback && hipEventCreateWithFlags(&m->ev_maps[g], hipEventDisableTiming) != hipSuccess)
            r = fail(OFARN_E_HIP, "event creation on device %d failed", dev[g]);
        return r;
    });
    if (!rc && !loopback)
        rc = on_devices(m, 0, 1, [&](int) -> int {
            m->comm.assign(n_devices, nullptr);
            const ncclResult_t r = R.CommInitAll(m->comm.data(), n_devices, dev.data());
            if (r != ncclSuccess) { m->comm.clear(); return fail(OFARN_E_HIP, "ncclCommInitAll over %d device(s) failed: %s", n_devices, R.GetErrorString(r)); }
            (void)hipSetDevice(dev[0]);          // ncclCommInitAll visits every device; worker 0 stays on its own
            return OFARN_OK;
        });
    if (rc) { const std::string keep = ofarn_last_error(); ofarn_multi_destroy(m); return fail(rc, "%s", keep.c_str()); }
    *out = m;
    return OFARN_OK;
}

void ofarn_multi_destroy(ofarn_multi *m)
{
    if (!m) return;
    if ((int)m->workers.size() == m->n && m->n > 0) {
        (void)on_all_devices(m, [&](int g) -> int { if (m->stream[g]) (void)hipStreamSynchronize(m->stream[g]); return OFARN_OK; });
        (void)on_devices(m, 0, 1, [&](int) -> int {
            for (ncclComm_t c : m->comm) if (c) (void)rccl().CommDestroy(c);
            (void)hipSetDevice(m->dev[0]);
            return OFARN_OK;
        });
        (void)on_all_devices(m, [&](int g) -> int {
            ofarn_multi::Buf &b = m->buf[g];
            for (void *p : {(void *)b.frames, (void *)b.flow, (void *)b.gmask, (void *)b.gv, (void *)b.mask_all, (void *)b.v_all}) if (p) (void)hipFree(p);
            if (m->stream[g]) (void)hipStreamDestroy(m->stream[g]);
            if (g < (int)m->ev_maps.size() && m->ev_maps[g]) (void)hipEventDestroy(m->ev_maps[g]);
            if (m->ctx[g]) ofarn_destroy(m->ctx[g]);
            return OFARN_OK;
        });
    }
    for (auto &w : m->workers) {
        { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; w->cv.notify_all(); }
        if (w->th.joinable()) w->th.join();
    }
    delete m;
}

int ofarn_multi_device_count(const ofarn_multi *m) { return m ? m->n : 0; }

int ofarn_multi_info(const ofarn_multi *m, int *rccl_version, unsigned long long *allgather_calls, double *last_device_ms)
{
    if (!m) return fail(OFARN_E_INVALID, "multi is NULL");
    if (rccl_version) { int v = 0; if (rccl().GetVersion) (void)rccl().GetVersion(&v); *rccl_version = v; }
    if (allgather_calls) *allgather_calls = m->gathers;
    if (last_device_ms) *last_device_ms = m->last_ms;
    return OFARN_OK;
}

void *ofarn_multi_stream(const ofarn_multi *m, int rank) { return (m && rank >= 0 && rank < m->n) ? (void *)m->stream[rank] : nullptr; }
ofarn_ctx *ofarn_multi_context(const ofarn_multi *m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }

int ofarn_multi_calc_batch(ofarn_multi *m, const uint8_t *h_frames, int n_frames, int w, int h, int pairs_mode, float *h_flow,
                           uint8_t *h_mask, uint8_t *h_v)
{
    return multi_calc(m, h_frames, nullptr, n_frames, 0, w, h, pairs_mode, h_flow, nullptr, h_mask, h_v, nullptr, nullptr);
}

int ofarn_multi_calc_batch_device(ofarn_multi *m, const uint8_t *const *d_frames, int n_pairs, int w, int h, int pairs_mode,
                                  float *const *d_flow, uint8_t *const *d_mask_all, uint8_t *const *d_v_all)
{
    if (!d_frames) return fail(OFARN_E_INVALID, "frames is NULL");
    return multi_calc(m, nullptr, d_frames, 0, n_pairs, w, h, pairs_mode, nullptr, d_flow, nullptr, nullptr, d_mask_all, d_v_all);
}

int ofarn_multi_synchronize(ofarn_multi *m)
{
    if (!m) return fail(OFARN_E_INVALID, "multi is NULL");
    return on_all_devices(m, [&](int g) -> int { HIP_TRY(hipStreamSynchronize(m->stream[g])); return OFARN_OK; });
}

#pragma GCC visibility pop
}  // extern "C"
