// Multi-GPU plans behind the C ABI.
//
// Candidates are independent optimisation problems (own x, own L-BFGS state, own corridor), so a batch is block-partitioned over
// the devices and an evaluation needs NO collective.  One host thread + one frx_problem handle (own stream) per device runs its
// shard exactly as frx_optimize would; the only exchange of a plan is the winner selection at the end:
//     all-gather of (objective, global candidate id)   16 bytes per device
//     broadcast of the winner's 6N x 3 coefficients + N durations from the device that owns it   (~9.7 KB at N = 64)
// over RCCL (ncclCommInitAll, one communicator per device of this process, xGMI between the devices of a node).  librccl.so is
// loaded on first use (dlopen), so single-GPU users of libfrx.so never pay for it.  The reference has nothing to mirror here: its
// device code is pinned to device 0 (cuda_computer.cu:414).
//
// A second, in-process communicator (plain host memory) implements the same two operations; it is used when several shards
// share one physical device (tests on a 1-GPU box: RCCL refuses duplicate devices) or when RCCL cannot be loaded, and as the
// reference the RCCL result is checked against in tests.
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/frx.h"
#include "frx_device.hpp"
#include "frx_internal.hpp"

namespace {

// ---- the two RCCL entry points we need, resolved at run time ----
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
enum { NCCL_DOUBLE = 8 };                                      // ncclDouble / ncclFloat64 (rccl.h)
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && AllGather && Broadcast && GroupStart && GroupEnd;
    }
};
Rccl &rccl() { static Rccl r; return r; }

struct Shard {
    int device = 0, lo = 0, hi = 0;                            // candidates [lo, hi) of the batch
    frx_problem *h = nullptr;
    int NX = 0, P = 0;                                         // totals of the shard
    std::vector<int> poff, xoff;                               // shard-local offsets
    hipStream_t stream = nullptr;                              // exchange stream
    double *d_pair = nullptr, *d_table = nullptr, *d_payload = nullptr;   // device buffers of the exchange
    int rc = FRX_OK;
    std::string err;
};

} // namespace

struct frx_multi {
    int B = 0, G = 0, maxN = 0;
    bool use_rccl = false;
    std::vector<Shard> sh;
    std::vector<ncclComm_t> comms;
    std::vector<int> x_off, p_off;                             // [B+1] offsets of the WHOLE batch
    int last_comm = 0;                                         // 1 = the last winner exchange went through RCCL
};

namespace {

// The entry points below walk over the shards' devices with hipSetDevice; the caller's current device is its own business (bench.py mixes
// torch and HIP in one process: a changed current device would send later allocations and launches to the wrong GPU) and is put back on
// every exit path.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void block_range(int total, int r, int world, int &lo, int &hi) {     // earlier shards take the remainder (fast-racing_amd/dist.py)
    const int base = total / world, rem = total % world;
    lo = r * base + std::min(r, rem);
    hi = lo + base + (r < rem ? 1 : 0);
}

} // namespace

extern "C" {

int frx_multi_create(const frx_config *cfg, int n_devices, const int *devices, int B, const int *coarse_n, const double *ini_state,
                     const double *fin_state, const int *h_off, const double *h_rec, const int *v_off, const double *v_rec, frx_multi **out) {
    if (!cfg || !coarse_n || !ini_state || !fin_state || !h_off || !h_rec || !v_off || !v_rec || !out || B <= 0)
        return frx::set_error(FRX_ERR_INVALID_ARG, "frx_multi_create: null argument or B <= 0");
    *out = nullptr;
    const int ndev = frx_device_count();
    if (ndev <= 0) return frx::set_error(FRX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    DeviceGuard restore_device;
    int G = n_devices > 0 ? n_devices : ndev;
    G = std::min(G, B);                                                         // "only when it exceeds one device": never more shards than candidates
    frx_multi *m = new (std::nothrow) frx_multi();
    if (!m) return frx::set_error(FRX_ERR_ALLOC, "out of host memory");
    m->B = B; m->G = G;
    m->sh.resize(G);
    bool distinct = true;
    for (int g = 0; g < G; g++) {
        m->sh[g].device = devices ? devices[g] : g % ndev;
        if (m->sh[g].device < 0 || m->sh[g].device >= ndev) { delete m; return frx::set_error(FRX_ERR_INVALID_ARG, "frx_multi_create: device ordinal out of range"); }
        for (int k = 0; k < g; k++) distinct = distinct && m->sh[k].device != m->sh[g].device;
    }
    // polytope prefix sums: candidate b owns coarse_n[b] H-polytopes and 2 coarse_n[b] - 1 V-polytopes
    std::vector<int> hp(B + 1, 0), vp(B + 1, 0);
    for (int b = 0; b < B; b++) { hp[b + 1] = hp[b] + coarse_n[b]; vp[b + 1] = vp[b] + 2 * coarse_n[b] - 1; }
    m->x_off.assign(B + 1, 0); m->p_off.assign(B + 1, 0);
    for (int g = 0; g < G; g++) {
        Shard &s = m->sh[g];
        block_range(B, g, G, s.lo, s.hi);
        const int nb = s.hi - s.lo;
        // re-base the CSR arrays of the shard
        std::vector<int> ho(hp[s.hi] - hp[s.lo] + 1), vo(vp[s.hi] - vp[s.lo] + 1);
        for (size_t i = 0; i < ho.size(); i++) ho[i] = h_off[hp[s.lo] + i] - h_off[hp[s.lo]];
        for (size_t i = 0; i < vo.size(); i++) vo[i] = v_off[vp[s.lo] + i] - v_off[vp[s.lo]];
        const int rc = frx_problem_create(cfg, s.device, nb, coarse_n + s.lo, ini_state + 9 * (size_t)s.lo, fin_state + 9 * (size_t)s.lo, ho.data(),
                                          h_rec + 6 * (size_t)h_off[hp[s.lo]], vo.data(), v_rec + 3 * (size_t)v_off[vp[s.lo]], &s.h);
        if (rc != FRX_OK) { const std::string e = frx_last_error(); frx_multi_destroy(m); return frx::set_error(rc, "shard " + std::to_string(g) + ": " + e); }
        int tot[6];
        frx_problem_totals(s.h, tot);
        s.P = tot[1]; s.NX = tot[3];
        s.poff.resize(nb + 1); s.xoff.resize(nb + 1);
        frx_problem_layout(s.h, s.poff.data(), nullptr, s.xoff.data(), nullptr);
        for (int b = 0; b < nb; b++) {
            m->x_off[s.lo + b + 1] = m->x_off[s.lo + b] + (s.xoff[b + 1] - s.xoff[b]);
            m->p_off[s.lo + b + 1] = m->p_off[s.lo + b] + (s.poff[b + 1] - s.poff[b]);
            m->maxN = std::max(m->maxN, s.poff[b + 1] - s.poff[b]);
        }
    }
    // exchange resources: RCCL when every shard has its own device (and the library loads), the in-process communicator otherwise
    const char *ce = std::getenv("FRX_MULTI_COMM");
    const bool want_rccl = !(ce && std::strcmp(ce, "host") == 0);
    if (want_rccl && distinct && rccl().load()) {
        std::vector<int> devs(G);
        for (int g = 0; g < G; g++) devs[g] = m->sh[g].device;
        m->comms.assign(G, nullptr);
        if (rccl().CommInitAll(m->comms.data(), G, devs.data()) == 0) m->use_rccl = true;
        else m->comms.clear();
    }
    const size_t payload = 1 + (size_t)m->maxN * 19;
    for (int g = 0; g < G; g++) {
        Shard &s = m->sh[g];
        if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void **)&s.d_pair, 2 * sizeof(double)) != hipSuccess || hipMalloc((void **)&s.d_table, 2 * sizeof(double) * G) != hipSuccess ||
            hipMalloc((void **)&s.d_payload, sizeof(double) * payload) != hipSuccess) {
            frx_multi_destroy(m);
            return frx::set_error(FRX_ERR_ALLOC, "frx_multi_create: exchange buffers");
        }
    }
    *out = m;
    return FRX_OK;
}

void frx_multi_destroy(frx_multi *m) {
    if (!m) return;
    DeviceGuard restore_device;
    for (size_t g = 0; g < m->comms.size(); g++) if (m->comms[g]) rccl().CommDestroy(m->comms[g]);
    for (Shard &s : m->sh) {
        (void)hipSetDevice(s.device);
        if (s.d_pair) (void)hipFree(s.d_pair);
        if (s.d_table) (void)hipFree(s.d_table);
        if (s.d_payload) (void)hipFree(s.d_payload);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.h) frx_problem_destroy(s.h);
    }
    delete m;
}

int frx_multi_info(const frx_multi *m, int *n_shards, int *uses_rccl, int *shard_lo /*[G+1] or NULL*/, int *shard_device /*[G] or NULL*/) {
    if (!m) return frx::set_error(FRX_ERR_INVALID_ARG, "null argument");
    if (n_shards) *n_shards = m->G;
    if (uses_rccl) *uses_rccl = m->use_rccl ? 1 : 0;
    for (int g = 0; g < m->G; g++) { if (shard_lo) shard_lo[g] = m->sh[g].lo; if (shard_device) shard_device[g] = m->sh[g].device; }
    if (shard_lo) shard_lo[m->G] = m->B;
    return FRX_OK;
}

int frx_multi_layout(const frx_multi *m, int *piece_off, int *x_off) {
    if (!m) return frx::set_error(FRX_ERR_INVALID_ARG, "null argument");
    if (piece_off) std::copy(m->p_off.begin(), m->p_off.end(), piece_off);
    if (x_off) std::copy(m->x_off.begin(), m->x_off.end(), x_off);
    return FRX_OK;
}

int frx_multi_initial_guess(frx_multi *m, double *x0) {
    if (!m || !x0) return frx::set_error(FRX_ERR_INVALID_ARG, "null argument");
    DeviceGuard restore_device;
    for (Shard &s : m->sh) {
        const int rc = frx_initial_guess(s.h, x0 + m->x_off[s.lo]);
        if (rc != FRX_OK) return rc;
    }
    return FRX_OK;
}

// One plan of the whole batch.  Per-candidate outputs are packed for the whole batch like frx_optimize's (offsets: frx_multi_layout).
// winner_id / winner_objective / winner_C (maxN x 18) / winner_T (maxN) / winner_n: the result of the exchange, identical on every
// device (returned once); failed candidates (status < 0) and non-finite objectives never win; ties go to the lowest id.
int frx_multi_optimize(frx_multi *m, const frx_lbfgs_params *params, double *x, double *C, double *T, double *jerk_cost, double *objective,
                       int *status, int *iters, int *evals, int *winner_id, double *winner_objective, double *winner_C, double *winner_T,
                       int *winner_n) {
    if (!m || !params || !x || !status) return frx::set_error(FRX_ERR_INVALID_ARG, "null argument");
    DeviceGuard restore_device;
    const int G = m->G;
    std::vector<double> Cbuf, Tbuf, obj;
    if (!C) { Cbuf.resize((size_t)m->p_off[m->B] * 18); C = Cbuf.data(); }
    if (!T) { Tbuf.resize(m->p_off[m->B]); T = Tbuf.data(); }
    if (!objective) { obj.resize(m->B); objective = obj.data(); }
    // ---- 1. every device optimises its shard (one host thread each) ----
    auto run = [&](int g) {
        Shard &s = m->sh[g];
        s.rc = frx_optimize(s.h, params, x + m->x_off[s.lo], C + 18 * (size_t)m->p_off[s.lo], T + m->p_off[s.lo], jerk_cost ? jerk_cost + s.lo : nullptr,
                            objective + s.lo, status + s.lo, iters ? iters + s.lo : nullptr, evals ? evals + s.lo : nullptr);
        if (s.rc != FRX_OK) s.err = frx_last_error();
    };
    {
        // (the shards' resident plans spin their mailbox threads side by side: each takes its share of the process's CPUs, frx_api.cpp host_cpu_share)
        struct PlansHint { int n; explicit PlansHint(int k) : n(k) { frx::concurrent_plans_hint(n); } ~PlansHint() { frx::concurrent_plans_hint(-n); } } hint(G - 1);
        std::vector<std::thread> th;
        for (int g = 1; g < G; g++) th.emplace_back(run, g);
        run(0);
        for (auto &t : th) t.join();
    }
    for (int g = 0; g < G; g++) if (m->sh[g].rc != FRX_OK) return frx::set_error(m->sh[g].rc, "shard " + std::to_string(g) + ": " + m->sh[g].err);
    // ---- 2. local best of every shard ----
    std::vector<double> pair(2 * (size_t)G);
    for (int g = 0; g < G; g++) {
        const Shard &s = m->sh[g];
        double best = std::numeric_limits<double>::infinity(); int id = -1;
        for (int b = s.lo; b < s.hi; b++) {
            const double f = (status[b] < 0 || !std::isfinite(objective[b])) ? std::numeric_limits<double>::infinity() : objective[b];
            if (f < best) { best = f; id = b; }
        }
        pair[2 * g] = best; pair[2 * g + 1] = (double)id;
    }
    // ---- 3. all-gather (objective, id), argmin everywhere, broadcast of the winner's trajectory ----
    const size_t payload = 1 + (size_t)m->maxN * 19;
    std::vector<double> table(2 * (size_t)G), pay(payload, 0.0);
    auto fill_payload = [&](int id) {
        const int n = m->p_off[id + 1] - m->p_off[id];
        pay[0] = (double)n;
        std::memcpy(&pay[1], C + 18 * (size_t)m->p_off[id], sizeof(double) * 18 * n);
        std::memcpy(&pay[1 + 18 * (size_t)m->maxN], T + m->p_off[id], sizeof(double) * n);
    };
    auto argmin = [&](const std::vector<double> &tb, int &owner, int &id, double &val) {
        owner = -1; id = -1; val = std::numeric_limits<double>::infinity();
        for (int g = 0; g < G; g++) {
            const double f = tb[2 * g]; const int i = (int)tb[2 * g + 1];
            if (i < 0) continue;
            if (f < val || (f == val && i < id)) { val = f; id = i; owner = g; }
        }
    };
    int owner = -1, wid = -1; double wval = 0.0;
    m->last_comm = 0;
    if (m->use_rccl) {
        Rccl &R = rccl();
        for (int g = 0; g < G; g++) {
            Shard &s = m->sh[g];
            if (hipSetDevice(s.device) != hipSuccess || hipMemcpyAsync(s.d_pair, &pair[2 * g], 2 * sizeof(double), hipMemcpyHostToDevice, s.stream) != hipSuccess)
                return frx::set_error(FRX_ERR_HIP, "frx_multi_optimize: staging the exchange");
        }
        ncclResult_t nr = R.GroupStart();
        for (int g = 0; g < G && nr == 0; g++) nr = R.AllGather(m->sh[g].d_pair, m->sh[g].d_table, 2, NCCL_DOUBLE, m->comms[g], m->sh[g].stream);
        if (nr == 0) nr = R.GroupEnd(); else (void)R.GroupEnd();
        if (nr != 0) return frx::set_error(FRX_ERR_HIP, std::string("ncclAllGather: ") + (R.GetErrorString ? R.GetErrorString(nr) : "error"));
        // every device holds the same table; the host reads device 0's copy to learn the owner (each rank would do this locally)
        (void)hipSetDevice(m->sh[0].device);
        if (hipMemcpyAsync(table.data(), m->sh[0].d_table, sizeof(double) * 2 * G, hipMemcpyDeviceToHost, m->sh[0].stream) != hipSuccess ||
            hipStreamSynchronize(m->sh[0].stream) != hipSuccess)
            return frx::set_error(FRX_ERR_HIP, "frx_multi_optimize: reading the gathered table");
        argmin(table, owner, wid, wval);
        if (owner >= 0) {
            fill_payload(wid);
            Shard &so = m->sh[owner];
            (void)hipSetDevice(so.device);
            if (hipMemcpyAsync(so.d_payload, pay.data(), sizeof(double) * payload, hipMemcpyHostToDevice, so.stream) != hipSuccess)
                return frx::set_error(FRX_ERR_HIP, "frx_multi_optimize: staging the winner");
            nr = R.GroupStart();
            for (int g = 0; g < G && nr == 0; g++) nr = R.Broadcast(m->sh[g].d_payload, m->sh[g].d_payload, payload, NCCL_DOUBLE, owner, m->comms[g], m->sh[g].stream);
            if (nr == 0) nr = R.GroupEnd(); else (void)R.GroupEnd();
            if (nr != 0) return frx::set_error(FRX_ERR_HIP, std::string("ncclBroadcast: ") + (R.GetErrorString ? R.GetErrorString(nr) : "error"));
            // read the broadcast result back from the LAST device (not the owner unless G = 1): proves the payload crossed the fabric
            Shard &sl = m->sh[(owner + G - 1) % G];
            for (int g = 0; g < G; g++) { (void)hipSetDevice(m->sh[g].device); if (hipStreamSynchronize(m->sh[g].stream) != hipSuccess) return frx::set_error(FRX_ERR_HIP, "frx_multi_optimize: exchange failed"); }
            (void)hipSetDevice(sl.device);
            if (hipMemcpy(pay.data(), sl.d_payload, sizeof(double) * payload, hipMemcpyDeviceToHost) != hipSuccess) return frx::set_error(FRX_ERR_HIP, "frx_multi_optimize: reading the winner");
        }
        m->last_comm = 1;
    } else {
        table = pair;                                                           // in-process communicator: the gathered table is the concatenation
        argmin(table, owner, wid, wval);
        if (owner >= 0) fill_payload(wid);
    }
    if (winner_id) *winner_id = wid;
    if (winner_objective) *winner_objective = wval;
    const int wn = owner >= 0 ? (int)pay[0] : 0;
    if (winner_n) *winner_n = wn;
    if (winner_C && wn > 0) std::memcpy(winner_C, &pay[1], sizeof(double) * 18 * wn);
    if (winner_T && wn > 0) std::memcpy(winner_T, &pay[1 + 18 * (size_t)m->maxN], sizeof(double) * wn);
    return FRX_OK;
}

int frx_multi_last_exchange(const frx_multi *m) { return m ? m->last_comm : 0; }

// Device form of frx_line_segment_dilate for a batch of segments against one obstacle cloud (frx_corridor_kernels.hpp): host
// buffers in and out, one workgroup per segment.  n_planes[s] = number of half-space records of segment s (tangent planes in the
// reference's order, then the six planes of the local box).
int frx_dilate_batch(int device, int n_seg, const double *p1, const double *p2, const double *bbox, int n_obs, const double *obs, double offset,
                     int cap_planes, int *n_planes, double *h_rec, double *ell_C, double *ell_d) {
    if (n_seg < 1 || !p1 || !p2 || !bbox || n_obs < 0 || (n_obs && !obs) || cap_planes < 6 || !n_planes || !h_rec)
        return frx::set_error(FRX_ERR_INVALID_ARG, "frx_dilate_batch: null or out-of-range argument");
    if (frx_device_count() < 1) return frx::set_error(FRX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    DeviceGuard restore_device;
    if (hipSetDevice(device) != hipSuccess) return frx::set_error(FRX_ERR_INVALID_ARG, "frx_dilate_batch: device ordinal out of range");
    const int pcap = 4096;                                                           // 4096 candidate points per cell: 96 KB of LDS + flags
    double *d_p1 = nullptr, *d_p2 = nullptr, *d_obs = nullptr, *d_h = nullptr, *d_C = nullptr, *d_d = nullptr; int *d_np = nullptr;
    auto cleanup = [&]() { for (void *q : {(void *)d_p1, (void *)d_p2, (void *)d_obs, (void *)d_h, (void *)d_C, (void *)d_d, (void *)d_np}) if (q) (void)hipFree(q); };
    const size_t hb = sizeof(double) * 6 * (size_t)cap_planes * n_seg;
    if (hipMalloc((void **)&d_p1, 24 * (size_t)n_seg) != hipSuccess || hipMalloc((void **)&d_p2, 24 * (size_t)n_seg) != hipSuccess ||
        hipMalloc((void **)&d_obs, 24 * (size_t)std::max(n_obs, 1)) != hipSuccess || hipMalloc((void **)&d_h, hb) != hipSuccess ||
        hipMalloc((void **)&d_C, 72 * (size_t)n_seg) != hipSuccess || hipMalloc((void **)&d_d, 24 * (size_t)n_seg) != hipSuccess || hipMalloc((void **)&d_np, 4 * (size_t)n_seg) != hipSuccess) {
        cleanup();
        return frx::set_error(FRX_ERR_ALLOC, "frx_dilate_batch: device buffers");
    }
    hipError_t e = hipMemcpy(d_p1, p1, 24 * (size_t)n_seg, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_p2, p2, 24 * (size_t)n_seg, hipMemcpyHostToDevice);
    if (e == hipSuccess && n_obs) e = hipMemcpy(d_obs, obs, 24 * (size_t)n_obs, hipMemcpyHostToDevice);
    frx::DilateLaunch L;
    L.p1 = d_p1; L.p2 = d_p2; L.obs = d_obs; L.bbox[0] = bbox[0]; L.bbox[1] = bbox[1]; L.bbox[2] = bbox[2]; L.offset = offset;
    L.S = n_seg; L.n_obs = n_obs; L.cap_planes = cap_planes; L.pcap = pcap; L.n_planes = d_np; L.h_rec = d_h; L.ell_C = d_C; L.ell_d = d_d;
    if (e == hipSuccess) e = (hipError_t)frx::launch_dilate(L, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(n_planes, d_np, 4 * (size_t)n_seg, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_rec, d_h, hb, hipMemcpyDeviceToHost);
    if (e == hipSuccess && ell_C) e = hipMemcpy(ell_C, d_C, 72 * (size_t)n_seg, hipMemcpyDeviceToHost);
    if (e == hipSuccess && ell_d) e = hipMemcpy(ell_d, d_d, 24 * (size_t)n_seg, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return frx::set_error(FRX_ERR_HIP, std::string("frx_dilate_batch: ") + hipGetErrorString(e));
    for (int s = 0; s < n_seg; s++)
        if (n_planes[s] < 0) return frx::set_error(FRX_ERR_CAPACITY, n_planes[s] == -1 ? "frx_dilate_batch: more than 4096 obstacle points inside one cell's local box"
                                                                                         : "frx_dilate_batch: more half-spaces than cap_planes");
    return FRX_OK;
}

} // extern "C"
