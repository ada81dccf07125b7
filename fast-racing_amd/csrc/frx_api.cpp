// C ABI (include/frx.h) of the MI355X back-end: problem set-up, batched device evaluation,
// batched host L-BFGS driver.  Host-only translation unit (g++); the kernels live in
// frx_device.hip.  No CPU fallback: without a HIP device create() fails.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <vector>

#include "../../include/frx.h"
#include "../../include/frx_debug.h"
#include "frx_device.hpp"
#include "frx_internal.hpp"
#include "frx_lbfgs.hpp"
#include "frx_host_pool.hpp"
#include "frx_host_setup.hpp"
#include "frx_compact.hpp"

namespace {

thread_local std::string g_err = "";
int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(FRX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

// a pair of HIP events that is destroyed on every exit path
struct HipEventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t create() { hipError_t e = hipEventCreate(&e0); return e != hipSuccess ? e : hipEventCreate(&e1); }
    ~HipEventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};

using clk = std::chrono::steady_clock;
inline double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    hipError_t alloc(size_t count) {
        release();
        const hipError_t e = hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) n = count; else p = nullptr;
        return e;
    }
    hipError_t upload(const std::vector<T> &h) {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};
template <class T> struct PinBuf {
    T *p = nullptr;
    size_t n = 0;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
    hipError_t alloc(size_t count) {
        release();
        hipError_t e = hipHostMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent);
        if (e == hipSuccess) { n = count; std::memset(p, 0, std::max<size_t>(count, 1) * sizeof(T)); } else p = nullptr;
        return e;
    }
};

} // namespace

namespace frx {
int set_error(int code, const std::string &msg) { return fail(code, msg); }

// ---- this plan's share of the host CPUs (the mailbox threads of a resident plan SPIN for its whole length) ----
static std::atomic<int> g_extra_plans{0};                 // plans of this process that run next to the caller's (frx_multi: shards - 1)
void concurrent_plans_hint(int delta) { g_extra_plans.fetch_add(delta, std::memory_order_relaxed); }
// CPUs the process may use: the affinity mask, cut by the cgroup's CPU quota (v2: cpu.max "quota period"; v1: cpu.cfs_quota_us / cpu.cfs_period_us)
double host_cpu_budget() {
    double cpus = (double)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0 && CPU_COUNT(&allowed) > 0) cpus = std::min(cpus, (double)CPU_COUNT(&allowed));
    auto read2 = [](const char *path, double &a, double &b) -> int {
        FILE *f = std::fopen(path, "r");
        if (!f) return 0;
        char t0[64] = {0}, t1[64] = {0};
        const int n = std::fscanf(f, "%63s %63s", t0, t1);
        std::fclose(f);
        if (n >= 1 && std::strcmp(t0, "max") == 0) return -1;            // no quota
        if (n >= 1) a = std::atof(t0);
        if (n >= 2) b = std::atof(t1);
        return n;
    };
    double q = 0.0, per = 0.0;
    const int n2 = read2("/sys/fs/cgroup/cpu.max", q, per);
    if (n2 == 2 && q > 0.0 && per > 0.0) cpus = std::min(cpus, q / per);
    else if (n2 == 0) {
        double q1 = 0.0, p1 = 0.0, dummy = 0.0;
        if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q1, dummy) >= 1 && read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", p1, dummy) >= 1 && q1 > 0.0 && p1 > 0.0)
            cpus = std::min(cpus, q1 / p1);
    }
    if (const char *e = std::getenv("FRX_HOST_CPUS")) { const double v = std::atof(e); if (v > 0.0) cpus = v; }   // (tests, experiments)
    return std::max(1.0, cpus);
}
int host_cpu_share();
// mailbox service threads of a resident plan with S clusters (the caller included): one per sixteen clusters, at most four, and at most share - 1
int mailbox_threads(int S) { return std::max(1, std::min(std::max(1, std::min(4, S / 16)), host_cpu_share() - 1)); }
//  The budget is divided by the node's ranks only when they SHARE it: a launcher that pins every rank to its own CPU subset (the affinity mask is
// smaller than the online set) has made the division already - dividing again left large batches with one mailbox thread where S / 16 were measured to help.
// FRX_LOCAL_RANKS states the divisor explicitly and is always honoured.
static int share_of(int extra_plans) {
    int ranks = 1;
    if (const char *e = std::getenv("FRX_LOCAL_RANKS")) ranks = std::max(1, std::atoi(e));
    else if (const char *e2 = std::getenv("LOCAL_WORLD_SIZE")) {
        ranks = std::max(1, std::atoi(e2));
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        const long online = sysconf(_SC_NPROCESSORS_ONLN);
        if (ranks > 1 && sched_getaffinity(0, sizeof(allowed), &allowed) == 0 && online > 0 && (long)CPU_COUNT(&allowed) * ranks <= online) ranks = 1;   // pinned per rank
    }
    const int plans = ranks * (1 + std::max(0, extra_plans));
    return std::max(1, (int)(host_cpu_budget() / plans));
}
int host_cpu_share() { return share_of(g_extra_plans.load(std::memory_order_relaxed)); }
int mailbox_threads_for(int S, int extra_plans) { return std::max(1, std::min(std::max(1, std::min(4, S / 16)), share_of(extra_plans) - 1)); }
int host_cpu_share_for(int extra_plans) { return share_of(extra_plans); }
} // namespace frx

static bool device_numa_cpus(int device, cpu_set_t *out);
// CPUs of the NUMA node the device hangs on (sysfs: the PCI function's numa_node, the node's cpulist), intersected with what the process may use.
// The mailbox threads of a resident plan are kept there: with the threads on the OTHER socket of a two-socket box every command the device reads
// and every result it writes crosses the socket interconnect and its cache-coherence traffic (measured, 32 candidates: 28.2-28.9 us per round with
// the process on the device's node, 30.6-32.5 on the other one; waits for the host's confirmation: p90 2.3 against 6.9 us - profiles/r04_numa.txt).
static bool device_numa_cpus_uncached(int device, cpu_set_t *out);
//  looked up once per device and process - two sysfs reads and a PCI query are not something to do while a resident kernel's clusters
// already spin; a change of the process's affinity mask after the first plan on a device is not followed
static bool device_numa_cpus(int device, cpu_set_t *out) {
    struct Entry { bool ok; cpu_set_t set; };
    static std::mutex lock;
    static std::map<int, Entry> cache;
    std::lock_guard<std::mutex> g(lock);
    auto it = cache.find(device);
    if (it == cache.end()) { Entry e; e.ok = device_numa_cpus_uncached(device, &e.set); it = cache.emplace(device, e).first; }
    if (it->second.ok) *out = it->second.set;
    return it->second.ok;
}
static bool device_numa_cpus_uncached(int device, cpu_set_t *out) {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), device) != hipSuccess) return false;
    for (char *c = bdf; *c; c++) *c = (char)std::tolower(*c);
    int node = -1;
    { const std::string p = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node"; if (FILE *f = std::fopen(p.c_str(), "r")) { if (std::fscanf(f, "%d", &node) != 1) node = -1; std::fclose(f); } }
    if (node < 0) return false;
    std::string list;
    { const std::string p = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist"; if (FILE *f = std::fopen(p.c_str(), "r")) { char buf[4096] = {0}; if (std::fgets(buf, sizeof(buf), f)) list = buf; std::fclose(f); } }
    if (list.empty()) return false;
    cpu_set_t allowed, want;
    CPU_ZERO(&allowed); CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    for (size_t i = 0; i < list.size();) {                                   // "0-63,128-191"
        if (!std::isdigit((unsigned char)list[i])) { i++; continue; }
        size_t j = i; int a = 0; while (j < list.size() && std::isdigit((unsigned char)list[j])) a = 10 * a + (list[j++] - '0');
        int b = a;
        if (j < list.size() && list[j] == '-') { j++; b = 0; while (j < list.size() && std::isdigit((unsigned char)list[j])) b = 10 * b + (list[j++] - '0'); }
        for (int c = a; c <= b && c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) CPU_SET(c, &want);
        i = j;
    }
    if (CPU_COUNT(&want) < 2) return false;                                 // nothing (or a single CPU) of that node is ours: leave the threads where they are
    *out = want;
    return true;
}

// The calling thread on the CPUs of the device's NUMA node for the length of a scope (FRX_NUMA=0: wherever it is).  Used around the allocation - and first touch - of
// the mailboxes in mapped host memory: pinned pages are taken from the node the allocating thread runs on, and a process whose main thread happened to sit on the other
// socket at that moment kept its mailboxes there for good - every command the device read and every result it wrote crossed the socket interconnect, and so did the
// service threads' scans from the right socket.  Round 6 found the two MODES of a process that round 5 put down to "the box's host" (profiles/r06_mode_probe.jsonl:
// 25.0-25.2 us per round with a mailbox scan of 0.07 us, 26.2-26.3 with 0.12-0.15, constant within a process, same clocks, same XCD placement; stretching the scan
// period itself to 3.2 us changes nothing, profiles/r06_ab_host_scan.jsonl).
struct NumaScope {
    cpu_set_t saved; bool active = false;
    explicit NumaScope(int device) {
        const char *e = std::getenv("FRX_NUMA"), *ea = std::getenv("FRX_NUMA_ALLOC");       // FRX_NUMA_ALLOC=0: allocations where the caller happens to run (round 5's behaviour, for A/B)
        cpu_set_t want;
        if ((e && e[0] == '0') || (ea && ea[0] == '0') || !device_numa_cpus(device, &want)) return;
        if (pthread_getaffinity_np(pthread_self(), sizeof(saved), &saved) != 0) return;
        active = pthread_setaffinity_np(pthread_self(), sizeof(want), &want) == 0;
    }
    ~NumaScope() { if (active) (void)pthread_setaffinity_np(pthread_self(), sizeof(saved), &saved); }
};

struct frx_problem {
    frx_config cfg;
    int device = 0, B = 0, P = 0, Pc = 0, NX = 0, Kmax = 0, maxN = 0, maxCN = 0, sumKfine = 0;
    bool softT = true;
    std::vector<frx::HostCand> cand;
    std::vector<int> poff, coff, xoff, boff, dimT;
    frx::DevProblem dp;
    hipStream_t stream = nullptr;
    // device-resident constants
    DevBuf<int> d_cvoff, d_poff, d_coff, d_xoff, d_boff, d_piece_hbeg, d_piece_K, d_piece_coarse, d_piece_iv, d_coarse_iv, d_coarse_fbeg, d_wp_vbeg,
        d_wp_nv, d_wp_xbeg;
    DevBuf<double> d_head, d_tail, d_hblk, d_vrec;
    // device work space
    DevBuf<double> d_x, d_f, d_g, d_T, d_C, d_band, d_out20;
    DevBuf<long long> d_stamps;
    DevBuf<double> d_pcrw;
    DevBuf<double> d_wq;                                    // [P][4]: per waypoint {|xi|^2, sum_a V_a xi_a^2}, forward map -> adjoint of the same evaluation
    // pinned staging
    PinBuf<double> h_x, h_f, h_g, h_T, h_C, h_out20;
    // device-vector L-BFGS state (allocated on first use)
    DevBuf<double> d_xp, d_gp, d_dir, d_S, d_Y, d_ys, d_gt;
    PinBuf<frx::DvCommand> h_cmd;
    PinBuf<frx::DvResult> h_res;
    int dv_mem = 0; size_t dv_hs = 0;
    double *pending_f = nullptr, *pending_g = nullptr;      // outputs of an frx_objective_eval_async awaiting frx_wait
    // line-search tap of k_backward_knot (set only while optimize_device_vectors runs)
    const double *tap_d = nullptr; const int *tap_flags = nullptr; void *tap_res = nullptr;
    unsigned *tap_arrive = nullptr; volatile unsigned *tap_flag = nullptr; unsigned tap_round = 0;
    DevBuf<unsigned> d_arrive; PinBuf<unsigned> h_flag; DevBuf<int> d_flags, d_pflags;
    // resident round kernel (frx_round_kernel.hpp): cluster exchange buffers and mapped mailboxes, allocated on first use
    DevBuf<double> d_pubsyg, d_part, d_upub, d_dpub, d_rdbg, d_out20ll;
    int dirlog_cap = 0, dirlog_cands = 0, dirlog_nxp = 0;   // direction log of the resident kernel (frx_debug_direction_log): records asked for / row geometry of the last plan
    std::vector<double> dirlog;                             // [B] counts, then dirlog_cands x dirlog_cap records of 4 NXP + 2 doubles
    DevBuf<unsigned long long> d_rprof;                     // FRX_RESIDENT_PROF: [B][G][16] per-segment ticks of the last resident launch
    std::vector<unsigned long long> rprof;
    DevBuf<unsigned> d_rwords;
    DevBuf<unsigned char> d_rargs;                          // the resident launch's arguments in device memory (k_round takes them by pointer)
    PinBuf<unsigned long long> h_rcmd, h_rres;              // [S] x 8 words each (S clusters: one mailbox per cluster)
    int rk_B = 0, rk_S = 0, rk_G = 0, rk_NXP = 0;
    // take-over of a per-stage batch's stragglers by the resident kernel (optimize_resident with a TakeOver): per cluster candidate index, last objective value,
    // newest slot and pair count of its history, and the dense state rebuilt from that history (frx_compact.hpp)
    DevBuf<int> d_rs_int; DevBuf<double> d_rs_f, d_rs_rinv, d_rs_yy, d_rs_vd;
    int taken_over = 0;                                     // candidates of the last plan that finished on the resident kernel after per-stage rounds
    long takeover_at = 0;                                   // tests (frx_debug_set_takeover_at): > 0 = every plan starts as per-stage rounds and hands over after this many, whatever the batch size
    int resident_clusters = 0;                              // clusters of the last resident launch (< B: the candidates went through the work queue)
    int resident_mode = 1;                                  // 1 = use the resident kernel when it applies (frx_problem_set_resident); 2 = also when the batch
                                                            // is larger than the chip holds at once, whatever its size (work queue); 0 = never
    int resident_retry = 0;                                 // 1 = re-run candidates that fail on the resident kernel on the per-stage rounds (diagnostic, frx_debug_set_resident_retry); the reference takes no second chance and neither does the default
    int resident_retried = 0;                               // candidates of the last plan re-run on the per-stage rounds after an L-BFGS error
    unsigned long long spec_counts[4] = {0, 0, 0, 0};       // last resident plan, summed over candidates: rounds started on a predicted ADVANCE / trial step, predictions redone, reserved
    int resident_failed = 0;                                // candidates of the last resident plan that ended with an L-BFGS error other than the iteration limit
    int resident_used = 0;                                  // diagnostics: 1 = the last frx_optimize ran on the resident kernel
    unsigned resident_status = 0;                           // device-side error code of the last resident launch (RK_ERR_*)
    std::vector<double> trace;                              // FRX_TRACE: per command of candidate 0 {flags, step, f, dg, dginit, xx, gg}
    // one launch per evaluation (frx_eval_kernel.hpp): granules and control words of its clusters, zeroed once; eval_fused: 1 = frx_objective_eval[_device] take it
    // (set at create when the geometry applies and the chip holds the whole batch at once; FRX_EVAL_FUSED=0 / frx_debug_set_eval_fused turn it off)
    DevBuf<unsigned long long> d_ev_ll; unsigned *d_ev_words = nullptr;   // one allocation: [78 P] granule words, then the [64 B + 1] control words
    int eval_fused = 0, eval_fused_G = 0, eval_fused_stamps = 0;
    unsigned long long eval_fused_ticks = 25000000ull;      // bound of every wait inside the launch, ticks of the 100 MHz counter: 250 ms (a healthy evaluation takes ~20 us; FRX_EVAL_TIMEOUT_MS)
    std::vector<unsigned char> ev_args, ev_args_up;         // the one-launch evaluation's constant arguments (frx::eval_cluster_args): as they should be / as the device copy holds them
    DevBuf<unsigned char> d_ev_args;
    PinBuf<unsigned> h_ev_status;                           // mapped host word: the code of an expired wait, written by the leader that saw it - launch_eval reads it without a synchronisation
    unsigned eval_fused_code = 0;                           // the code that retired the one-launch form on this handle (0: none)
    // one launch per evaluation for LARGE batches (frx_solo_kernel.hpp: one workgroup per candidate): 0 = never, 1 = from eval_solo_min_B candidates on (default), 2 = always
    int eval_solo = 0, eval_solo_min_B = 0, eval_solo_max_B = 0;
    frx::LaunchGeom geo;
    bool banded_ok = true;
    bool penalty_only = false;              // frx_penalty_problem_create: the inner boundary alone (no variables, no waypoint polytopes: only frx_penalty_eval[_device] apply)
    int lbfgs_mode = 0;                     // 0 = device vectors (default), 1 = host vectors
    double stats[4] = {0, 0, 0, 0};
};

namespace {

int launch_eval(frx_problem *p, const double *x_dev, double *f_dev, double *g_dev, void *st, bool backward) {
    // a plain evaluation of a batch the chip holds at once: ONE launch (clusters of workgroups, frx_eval_kernel.hpp); the optimiser's rounds (line-search tap,
    // skipped candidates) and the diagnostics that look at stage buffers keep the three stage kernels
    //  A wait inside an earlier one-launch evaluation expired: its leader left the code in mapped host memory, and from the first launch that sees it the
    // handle takes the stage kernels by itself - also on the capturable _device form and inside the host-vector L-BFGS, which have no status check of their own.
    // No HIP call here (the caller may be capturing): the device's sticky word is cleared at the next host-synchronous point (eval_cluster_status).
    if (p->eval_fused && p->h_ev_status.p && *(volatile unsigned *)p->h_ev_status.p != 0u) { p->eval_fused_code = *(volatile unsigned *)p->h_ev_status.p; p->eval_fused = 0; }
    const bool solo_ok = backward && p->geo.lds_solo && p->geo.solver == frx::SOLVER_KNOT_PCR && (!p->dp.stamps || p->eval_solo == 2);   // (cycle stamps of the stage kernels' phases: frx_profile_phases; the forced form carries them too)
    auto solo = [&]() { return frx::launch_eval_solo(p->dp, p->geo, x_dev, p->d_T.p, p->d_C.p, p->d_out20.p, f_dev, g_dev, st, p->tap_d, p->tap_flags, p->tap_res, p->tap_arrive, p->tap_flag, p->tap_round); };
    if (solo_ok && p->eval_solo == 2) return solo();                          // (forced: tests and measurements at batch sizes the other forms would take)
    if (backward && p->eval_fused && p->geo.solver == frx::SOLVER_KNOT_PCR && !p->tap_d && !p->dp.cand_active && (!p->dp.stamps || p->eval_fused_stamps)) {
        // the handle's constant arguments live in device memory (uploaded at create); they only change when a diagnostic switches the cycle stamps on or off -
        // a synchronous copy then (never inside somebody's capture: the diagnostics are blocking calls of their own)
        frx::eval_cluster_args(p->dp, p->geo, p->d_T.p, p->d_C.p, p->d_ev_ll.p, p->d_ev_words, p->ev_args.data());
        if (p->ev_args != p->ev_args_up) {
            if (hipMemcpy(p->d_ev_args.p, p->ev_args.data(), p->ev_args.size(), hipMemcpyHostToDevice) != hipSuccess) return (int)hipErrorUnknown;
            p->ev_args_up = p->ev_args;
        }
        return frx::launch_eval_cluster(p->geo, p->B, p->ev_args.data(), p->d_ev_args.p, x_dev, f_dev, g_dev, p->eval_fused_ticks, st, p->h_ev_status.p);
    }
    // batches beyond the clusters' reach: one workgroup per candidate runs the three stage bodies back to back - plain evaluations AND the optimiser's rounds (same
    // stage buffers, tap and skipped candidates as the three launches; bit-identical results)
    if (solo_ok && p->eval_solo == 1 && p->B >= p->eval_solo_min_B && p->B <= p->eval_solo_max_B) return solo();
    int e = frx::launch_forward(p->dp, p->geo, x_dev, p->d_T.p, p->d_C.p, backward ? p->d_band.p : (double *)nullptr, st);
    if (e || !backward) return e;
    if ((e = frx::launch_penalty(p->dp, p->geo, p->d_T.p, p->d_C.p, p->d_out20.p, st))) return e;
    return frx::launch_backward(p->dp, p->geo, x_dev, p->d_T.p, p->d_C.p, p->d_band.p, p->d_out20.p, f_dev, g_dev, st, p->tap_d, p->tap_flags, p->tap_res, p->tap_arrive, p->tap_flag, p->tap_round);
}

// The one-launch evaluation's sticky error word (a wait inside the launch expired: the f of the candidates concerned are NaN, so the word is only fetched when
// an objective value is not a number).  Read, cleared, reported; the handle then goes on with the stage kernels.  The stream has been synchronised by the caller.
// f == nullptr (frx_eval_status: the caller of the capturable form asks): the word is fetched whatever the objective values say.
int eval_cluster_status(frx_problem *p, const double *f) {
    if (!p->eval_fused_G || !p->d_ev_words) return FRX_OK;
    const unsigned seen = p->h_ev_status.p ? *(volatile unsigned *)p->h_ev_status.p : 0u;
    if (!p->eval_fused && !p->eval_fused_code && !seen) return FRX_OK;                  // switched off by the caller, never failed
    bool any_nan = f == nullptr || seen != 0u || p->eval_fused_code != 0u;
    for (int b = 0; f && b < p->B; b++) any_nan = any_nan || f[b] != f[b];
    if (!any_nan) return FRX_OK;
    unsigned st = 0;
    unsigned *w = p->d_ev_words + (size_t)64 * p->B;
    HIP_TRY(hipMemcpy(&st, w, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (st == 0 && seen == 0u && p->eval_fused_code == 0u) return FRX_OK;
    if (st == 0) st = seen ? seen : p->eval_fused_code;
    HIP_TRY(hipMemset(w, 0, sizeof(unsigned)));
    if (p->h_ev_status.p) *(volatile unsigned *)p->h_ev_status.p = 0u;
    p->eval_fused = 0; p->eval_fused_code = 0;
    return fail(FRX_ERR_TIMEOUT, "one-launch evaluation: a wait between the workgroups of a cluster expired (code " + std::to_string(st) + "); this handle continues with one launch per stage");
}

} // namespace

extern "C" {

int frx_version(void) { return FRX_VERSION; }
const char *frx_last_error(void) { return g_err.c_str(); }
int frx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
void frx_lbfgs_default_params(frx_lbfgs_params *p) { if (p) frx::lbfgs_defaults(*p); }
void frx_lbfgs_gcopter_params(frx_lbfgs_params *p, double rel_cost_tol) {
    if (!p) return;
    frx::lbfgs_defaults(*p);
    p->mem_size = 128; p->past = 3; p->g_epsilon = 1.0e-16; p->min_step = 1.0e-32; p->delta = rel_cost_tol;
}

int frx_problem_create(const frx_config *cfg, int device, int B, const int *coarse_n, const double *ini_state,
                       const double *fin_state, const int *h_off, const double *h_rec, const int *v_off, const double *v_rec,
                       frx_problem **out) {
    if (!cfg || !coarse_n || !ini_state || !fin_state || !h_off || !h_rec || !v_off || !v_rec || !out || B <= 0)
        return fail(FRX_ERR_INVALID_ARG, "frx_problem_create: null argument or B <= 0");
    if (cfg->qd_intervals < 1) return fail(FRX_ERR_INVALID_ARG, "qd_intervals must be >= 1");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FRX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(FRX_ERR_INVALID_ARG, "device ordinal out of range");
    if (hipSetDevice(device) != hipSuccess) return fail(FRX_ERR_NO_DEVICE, "hipSetDevice failed");

    const auto t_create0 = clk::now();
    frx_problem *p = new (std::nothrow) frx_problem();
    if (!p) return fail(FRX_ERR_ALLOC, "out of host memory");
    p->cfg = *cfg;
    p->device = device;
    p->B = B;
    p->softT = cfg->rho > 0;                                             // CPU.hpp:1097
    p->cand.resize(B);
    p->poff.assign(B + 1, 0); p->coff.assign(B + 1, 0); p->xoff.assign(B + 1, 0); p->boff.assign(B + 1, 0);
    p->dimT.assign(B, 0);

    std::vector<int> piece_hbeg, piece_K, piece_coarse, piece_iv, coarse_iv, coarse_fbeg, wp_vbeg, wp_nv, wp_xbeg, cvoff(B + 1, 0);
    std::vector<double> hrec, horg, vrec, head(ini_state, ini_state + 9 * (size_t)B), tail(fin_state, fin_state + 9 * (size_t)B);
    int hpoly = 0, vpoly = 0;                     // running polytope indices into h_off / v_off
    for (int b = 0; b < B; b++) {
        frx::HostCand &hc = p->cand[b];
        const int cN = coarse_n[b];
        if (cN < 1) { delete p; return fail(FRX_ERR_INVALID_ARG, "coarse_n[b] < 1"); }
        // host description: V-polytopes as [v0, v_r - v0], gridMesh, waypoint index map, clipped boundary speeds (frx_host_setup.hpp)
        if (frx::host_cand_init(hc, *cfg, p->softT, cN, ini_state + 9 * (size_t)b, fin_state + 9 * (size_t)b, v_off + vpoly, v_rec) != FRX_OK) {
            delete p; return fail(FRX_ERR_EMPTY_POLYTOPE, "a corridor polytope has no vertices");
        }
        cvoff[b] = (int)(vrec.size() / 3);
        // index maps (CPU.hpp:1129-1152), expanded into per-piece / per-waypoint device descriptors
        const int gp0 = p->poff[b], gc0 = p->coff[b];
        int offset = 0;
        int xcur = p->xoff[b] + hc.dimT;
        for (int i = 0; i < cN; i++) {
            const int hb = h_off[hpoly + i], K = h_off[hpoly + i + 1] - hb;
            if (K < 1) { delete p; return fail(FRX_ERR_INVALID_ARG, "an H-polytope has no half-spaces"); }
            const int hbeg_dev = (int)(hrec.size() / 4);
            const double *org = h_rec + 6 * (size_t)hb + 3;              // polytope origin = point of its first half-space
            horg.push_back(org[0]); horg.push_back(org[1]); horg.push_back(org[2]);
            for (int k = 0; k < K; k++) {                                // normalise outer normals, CPU.hpp:1116
                const double *rec = h_rec + 6 * (size_t)(hb + k);
                const double nn = std::sqrt(rec[0] * rec[0] + rec[1] * rec[1] + rec[2] * rec[2]);
                const double n0 = rec[0] / nn, n1 = rec[1] / nn, n2 = rec[2] / nn;
                hrec.push_back(n0); hrec.push_back(n1); hrec.push_back(n2);
                // n.(pos - p_k) + margin = n.(pos - org) - c   with   c = n.(p_k - org) - margin   (CPU.hpp:325,328)
                hrec.push_back(n0 * (rec[3] - org[0]) + n1 * (rec[4] - org[1]) + n2 * (rec[5] - org[2]) - cfg->safe_margin);
            }
            p->Kmax = std::max(p->Kmax, K);
            coarse_iv.push_back(hc.intervals[i]);
            coarse_fbeg.push_back(gp0 + offset);
            for (int j = 0; j < hc.intervals[i]; j++) {
                int vm = -1;
                if (j < hc.intervals[i] - 1) vm = 2 * i;
                else if (i < cN - 1) vm = 2 * i + 1;
                if (vm >= 0) {
                    const int nv = (int)(hc.cfgVs[vm].size() / 3);
                    wp_vbeg.push_back((int)(vrec.size() / 3)); wp_nv.push_back(nv); wp_xbeg.push_back(xcur);
                    vrec.insert(vrec.end(), hc.cfgVs[vm].begin(), hc.cfgVs[vm].end());   // waypoint order, contiguous per candidate
                    xcur += nv - 1;
                }
                piece_hbeg.push_back(hbeg_dev); piece_K.push_back(K); piece_coarse.push_back(gc0 + i); piece_iv.push_back(hc.intervals[i]);
                p->sumKfine += K;
                offset++;
            }
        }
        hpoly += cN;
        vpoly += 2 * cN - 1;
        p->dimT[b] = hc.dimT;
        p->poff[b + 1] = gp0 + hc.fineN;
        p->coff[b + 1] = gc0 + cN;
        p->xoff[b + 1] = p->xoff[b] + hc.dimT + hc.dimP;
        p->boff[b + 1] = p->boff[b] + 6 * hc.fineN * FRX_BAND_W;
        p->maxN = std::max(p->maxN, hc.fineN);
        p->maxCN = std::max(p->maxCN, cN);
    }
    cvoff[B] = (int)(vrec.size() / 3);
    p->P = p->poff[B]; p->Pc = p->coff[B]; p->NX = p->xoff[B];
    int maxXb = 1, maxVb = 3;
    for (int b = 0; b < B; b++) {
        maxXb = std::max(maxXb, p->xoff[b + 1] - p->xoff[b]);
        maxVb = std::max(maxVb, 3 * (cvoff[b + 1] - cvoff[b]));
    }

    // launch geometry + LDS budgets
    const int spp = cfg->qd_intervals + 1;
    frx::LaunchGeom &ge = p->geo;
    ge.maxN = p->maxN; ge.maxCN = p->maxCN; ge.Kmax = p->Kmax;
    ge.lpp = std::min(spp, 64);
    ge.ppw = 64 / ge.lpp;
    {   // stage kernel k_penalty: a workgroup of W waves owns floor(64 W / lpp) whole pieces; W in 1..4 for the best lane utilisation (the
        // fewest waves among equals) - kappa = 16: 51/64 lanes with one wave, 255/256 with four; stock kappa = 48: 49/64 and 245/256.
        // Measured (profiles/r03_penalty_waves_per_workgroup.jsonl, kappa = 16): 1024 candidates 40.3 -> 35.9 us, 4096 candidates 153 -> 132 us
        // (kappa = 48, 1024 candidates: 116 -> 95 us); but a launch that fits the chip's SIMDs in one go is ONE wave's latency, and there
        // one-wave workgroups (no barrier, twice the CUs in use) win: 4.79 against 5.17 us at the headline batch.  FRX_PENALTY_WAVES=1..4 overrides.
        int best_w = 1; double best_u = 0.0;
        for (int w = 1; w <= 4; w++) {
            const double u = (double)((64 * w) / ge.lpp * ge.lpp) / (64.0 * w);
            if (u > best_u + 1e-9) { best_u = u; best_w = w; }
        }
        int simds = 1024;
        { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess) simds = 4 * prop.multiProcessorCount; }
        if ((p->P + ge.ppw - 1) / ge.ppw <= simds) best_w = 1;
        if (const char *pw = std::getenv("FRX_PENALTY_WAVES")) { const int w = std::atoi(pw); if (w >= 1 && w <= 4) best_w = w; }
        //  W is chosen for lane utilisation, but the workgroup's LDS grows with it (ppg corridor blocks of Kmax + 1 records): step W down
        // until the block fits the CU - a corridor with many half-spaces runs on fewer waves instead of failing create
        auto pen_lds = [&](int w) { const int ppg = (64 * w) / ge.lpp; return sizeof(double) * ((size_t)ppg * 19 + (size_t)ppg * (p->Kmax + 1) * 4 + (size_t)64 * w * 21); };
        while (best_w > 1 && pen_lds(best_w) > (size_t)160 * 1024) best_w--;
        ge.pen_w = best_w; ge.ppg = (64 * best_w) / ge.lpp;
    }
    ge.lds_fwd = sizeof(double) * ((size_t)6 * p->maxN * (FRX_BAND_W + 3) + p->maxN + p->maxCN);
    ge.lds_bwd = sizeof(double) * ((size_t)6 * p->maxN * (FRX_BAND_W + 6) + 2 * (size_t)p->maxN + p->maxCN);
    ge.lds_pen = sizeof(double) * ((size_t)ge.ppg * 19 + (size_t)ge.ppg * (p->Kmax + 1) * 4 + (size_t)64 * ge.pen_w * 21);
    // large batches (four-wave workgroups), one sample per lane: the two-phase form of the integrator (k_penalty_lat2) - half the transpose buffer, four waves per SIMD
    ge.lds_pen2 = (ge.pen_w == 4 && spp <= 64) ? sizeof(double) * ((size_t)ge.ppg * 19 + (size_t)ge.ppg * (p->Kmax + 1) * 4 + (size_t)64 * ge.pen_w * 11) : 0;
    ge.solver = frx::SOLVER_KNOT_PCR;
    ge.knot_threads = 64 * ((p->maxN + 63) / 64);
    {
        const size_t nt = ge.knot_threads;
        ge.maxXb = maxXb; ge.maxVb = maxVb;
        ge.pcr_steps = 0;
        for (int s = 1; s < p->maxN - 1; s <<= 1) ge.pcr_steps++;
        ge.lds_kfwd = sizeof(double) * (36 * nt + 9 * (nt + 1) + nt + p->maxCN + maxXb + maxVb + (size_t)(ge.pcr_steps * 8 + 5) * nt);
        ge.pcr_steps = 0;
        for (int s = 1; s < p->maxN - 1; s <<= 1) ge.pcr_steps++;
        ge.lds_kbwd = sizeof(double) * (36 * nt + 9 * (nt + 1) + 2 * nt + p->maxCN + 2 * 4 + 2 + 2 * maxXb + maxVb + (size_t)(ge.pcr_steps * 8 + 5) * nt);
    }
    {   // one launch per evaluation when every cluster of the batch gets its CUs at once
        int cus = 256;
        { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount; }
        const char *ef = std::getenv("FRX_EVAL_FUSED");
        const int G = frx::eval_cluster_geometry(ge);
        //  ranks of one node that SHARE a device (more local ranks than devices): their grids would compete for the CUs a cluster's leader assumes - off
        // unless asked for (FRX_EVAL_FUSED=1); a lone process, or one rank per device, takes the form
        int ranks = 1;
        if (const char *e = std::getenv("FRX_LOCAL_RANKS")) ranks = std::max(1, std::atoi(e));
        else if (const char *e2 = std::getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, std::atoi(e2));
        const bool shared_device = ranks > ndev && !(ef && ef[0] == '1');
        if (!G || (long long)B * G > cus || (ef && ef[0] == '0') || shared_device) { ge.ev_G = 0; ge.lds_ev = 0; }
        p->eval_fused_G = ge.ev_G; p->eval_fused = ge.ev_G ? 1 : 0;
        // (the solo form: from the batch size on at which the stage launches' ramps and stage-buffer trips outweigh a lone workgroup's five penalty passes - measured,
        // profiles/NOTES.md round 6; FRX_EVAL_SOLO=0 never, =1 always, FRX_EVAL_SOLO_MIN_B=n moves the threshold)
        frx::eval_solo_geometry(ge, spp);
        p->eval_solo = ge.lds_solo ? 1 : 0; p->eval_solo_min_B = 384; p->eval_solo_max_B = 512;
        if (const char *es = std::getenv("FRX_EVAL_SOLO")) { if (es[0] == '0') p->eval_solo = 0; else if (es[0] == '1' && ge.lds_solo) p->eval_solo = 2; }
        if (const char *mb = std::getenv("FRX_EVAL_SOLO_MIN_B")) { const int v = std::atoi(mb); if (v > 0) p->eval_solo_min_B = v; }
        if (const char *mb = std::getenv("FRX_EVAL_SOLO_MAX_B")) { const int v = std::atoi(mb); if (v > 0) p->eval_solo_max_B = v; }
        if (const char *tm = std::getenv("FRX_EVAL_TIMEOUT_MS")) { const double v = std::atof(tm); if (v > 0.0) p->eval_fused_ticks = (unsigned long long)(v * 1e5); }
    }
    const size_t lds_cap = 160 * 1024;
    if (ge.knot_threads > 256 || ge.lds_kbwd > lds_cap) {
        delete p;
        return fail(FRX_ERR_CAPACITY, "a candidate's knot system does not fit one workgroup: " + std::to_string(ge.lds_kbwd / 1024) + " KB of LDS needed, 160 KB available (" +
                                          std::to_string(ge.knot_threads) + " knot threads, limit 256; about 128 pieces per candidate fit; the reference caps at 100, cuda_computer.cuh:24)");
    }
    // the banded-LU cross-check kernels need the whole band in LDS; fall back to "unavailable" instead of failing create
    p->banded_ok = !(ge.lds_bwd > lds_cap || ge.lds_fwd > lds_cap);
    if (!p->banded_ok) { ge.lds_fwd = ge.lds_bwd = 1024; }
    if (ge.lds_pen > lds_cap) {
        delete p;
        return fail(FRX_ERR_CAPACITY, "piece count / half-space count too large for the LDS-resident kernels (160 KiB per CU)");
    }

#define CR(expr)                                                                                        \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            std::string m_ = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
            if (p->stream) (void)hipStreamDestroy(p->stream);                                           \
            delete p;                                                                                   \
            return fail(e_ == hipErrorOutOfMemory ? FRX_ERR_ALLOC : FRX_ERR_HIP, m_);                   \
        }                                                                                               \
    } while (0)
    const double ms_host_build = ms_since(t_create0);
    CR(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    CR((hipError_t)frx::launch_set_limits(p->geo));
    CR((hipError_t)frx::eval_solo_raise_limit(p->geo));
    if (p->geo.lds_solo && frx::eval_solo_blocks_per_cu(p->geo) < 1) { p->geo.lds_solo = 0; p->eval_solo = 0; }
    if (p->geo.ev_G && frx::eval_cluster_blocks_per_cu(p->geo.lds_ev) < 1) { p->geo.ev_G = 0; p->geo.lds_ev = 0; p->eval_fused_G = 0; p->eval_fused = 0; }   // (the runtime's own occupancy answer: a CU must hold a workgroup of the cluster kernel)
    CR(p->d_cvoff.upload(cvoff)); CR(p->d_poff.upload(p->poff)); CR(p->d_coff.upload(p->coff)); CR(p->d_xoff.upload(p->xoff)); CR(p->d_boff.upload(p->boff));
    CR(p->d_piece_hbeg.upload(piece_hbeg)); CR(p->d_piece_K.upload(piece_K)); CR(p->d_piece_coarse.upload(piece_coarse)); CR(p->d_piece_iv.upload(piece_iv));
    CR(p->d_coarse_iv.upload(coarse_iv)); CR(p->d_coarse_fbeg.upload(coarse_fbeg));
    CR(p->d_wp_vbeg.upload(wp_vbeg)); CR(p->d_wp_nv.upload(wp_nv)); CR(p->d_wp_xbeg.upload(wp_xbeg));
    CR(p->d_head.upload(head)); CR(p->d_tail.upload(tail)); {
        // per-piece corridor blocks, padded to Kmax: one contiguous, index-free read per piece in k_penalty
        const int hs = (p->Kmax + 1) * 4;
        std::vector<double> hblk((size_t)p->P * hs, 0.0);
        for (int gp = 0; gp < p->P; gp++) {
            double *blk = &hblk[(size_t)gp * hs];
            const int gc = piece_coarse[gp], K = piece_K[gp];
            blk[0] = horg[3 * (size_t)gc]; blk[1] = horg[3 * (size_t)gc + 1]; blk[2] = horg[3 * (size_t)gc + 2]; blk[3] = (double)K;
            std::memcpy(blk + 4, &hrec[4 * (size_t)piece_hbeg[gp]], sizeof(double) * 4 * K);
        }
        CR(p->d_hblk.upload(hblk));
    } CR(p->d_vrec.upload(vrec));
    CR(p->d_x.alloc(p->NX)); CR(p->d_f.alloc(B)); CR(p->d_g.alloc(p->NX));
    CR(p->d_T.alloc(p->P)); CR(p->d_C.alloc((size_t)p->P * 18)); CR(p->d_band.alloc(p->boff[B])); CR(p->d_out20.alloc((size_t)p->P * 20));
    CR(p->d_pcrw.alloc((size_t)(p->geo.pcr_steps * 8 + 4) * p->P)); p->geo.pcrw = p->d_pcrw.p;
    CR(p->d_wq.alloc((size_t)4 * p->P)); CR(hipMemset(p->d_wq.p, 0, sizeof(double) * 4 * p->P));
    if (p->eval_fused_G) {
        const size_t n_ll = (size_t)78 * p->P, n_w64 = ((size_t)64 * B + 2) / 2;
        CR(p->d_ev_ll.alloc(n_ll + n_w64)); CR(hipMemset(p->d_ev_ll.p, 0, sizeof(unsigned long long) * (n_ll + n_w64)));   // (not on the handle's stream: the first evaluation may come on the caller's)
        p->d_ev_words = (unsigned *)(p->d_ev_ll.p + n_ll);
        { NumaScope numa_alloc(device); CR(p->h_ev_status.alloc(16)); }
        p->ev_args.assign(frx::eval_cluster_args_bytes(), 0); CR(p->d_ev_args.alloc(p->ev_args.size()));
    }
    {   // pinned staging buffers: pages from the device's NUMA node (NumaScope)
        NumaScope numa_alloc(device);
        CR(p->h_x.alloc(p->NX)); CR(p->h_f.alloc(B)); CR(p->h_g.alloc(p->NX));
        CR(p->h_T.alloc(p->P)); CR(p->h_C.alloc((size_t)p->P * 18)); CR(p->h_out20.alloc((size_t)p->P * 20));
    }
#undef CR

    frx::DevProblem &d = p->dp;
    d.B = B; d.P = p->P; d.kappa = cfg->qd_intervals; d.inv_kappa = 1.0 / cfg->qd_intervals; d.soft = p->softT ? 1 : 0; d.c2 = cfg->c2_diffeo ? 1 : 0;
    d.rho = p->softT ? cfg->rho : 0.0;                                   // CPU.hpp:1098-1107
    d.sumT = p->softT ? 1.0 : cfg->total_t;
    d.pc.ell[0] = cfg->horiz_half_len; d.pc.ell[1] = cfg->horiz_half_len; d.pc.ell[2] = cfg->vert_half_len;
    d.pc.safeMargin = cfg->safe_margin;
    d.pc.vMaxSqr = cfg->vel_max * cfg->vel_max;
    d.pc.thrMinSqr = cfg->thr_acc_min * cfg->thr_acc_min;
    d.pc.thrMaxSqr = cfg->thr_acc_max * cfg->thr_acc_max;
    d.pc.bdrMaxSqr = cfg->body_rate_max * cfg->body_rate_max;
    d.pc.gAcc = cfg->grav_acc;
    for (int q = 0; q < 4; q++) d.pc.chi[q] = cfg->penalty_pvtb[q];
    d.cvoff = p->d_cvoff.p; d.poff = p->d_poff.p; d.coff = p->d_coff.p; d.xoff = p->d_xoff.p; d.boff = p->d_boff.p;
    d.headPVA = p->d_head.p; d.tailPVA = p->d_tail.p;
    d.piece_hbeg = p->d_piece_hbeg.p; d.piece_K = p->d_piece_K.p; d.piece_coarse = p->d_piece_coarse.p; d.piece_iv = p->d_piece_iv.p;
    d.coarse_iv = p->d_coarse_iv.p; d.coarse_fbeg = p->d_coarse_fbeg.p;
    d.wp_vbeg = p->d_wp_vbeg.p; d.wp_nv = p->d_wp_nv.p; d.wp_xbeg = p->d_wp_xbeg.p;
    d.hblk = p->d_hblk.p; d.vrec = p->d_vrec.p; d.wq_glob = p->d_wq.p; d.stamps = nullptr; d.cand_active = nullptr; d.piece_active = nullptr;
    if (p->eval_fused_G) {                                                  // (the first evaluation may be captured: the device copy of its arguments is complete before create returns)
        frx::eval_cluster_args(p->dp, p->geo, p->d_T.p, p->d_C.p, p->d_ev_ll.p, p->d_ev_words, p->ev_args.data());
        const hipError_t e_ = hipMemcpy(p->d_ev_args.p, p->ev_args.data(), p->ev_args.size(), hipMemcpyHostToDevice);
        if (e_ != hipSuccess) { const std::string m_ = std::string("upload of the evaluation arguments: ") + hipGetErrorString(e_); (void)hipStreamDestroy(p->stream); delete p; return fail(FRX_ERR_HIP, m_); }
        p->ev_args_up = p->ev_args;
    }
    if (std::getenv("FRX_SETUP_TIMING")) fprintf(stderr, "[frx setup] frx_problem_create: B = %d, host descriptors %.3f ms, device allocations + uploads %.3f ms\n", B, ms_host_build, ms_since(t_create0) - ms_host_build);
    *out = p;
    return FRX_OK;
}

void frx_problem_destroy(frx_problem *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) { (void)hipStreamSynchronize(p->stream); (void)hipStreamDestroy(p->stream); }
    delete p;
}

// The inner boundary on its own (include/frx.h): what cuda_computer::compute receives per call (cc.cuh:118-134) - idxHs, cfgHs, the ellipsoid, margin, limits
// and weights - is everything the penalty integrator needs; the waypoint polytopes, end states and variables of the outer boundary do not exist on that side
// of the reference (MINCO_S3 knows none of them).  The handle is an ordinary one whose candidates have one coarse piece per fine piece (gridRes = inf) and
// two-vertex placeholder polytopes that no entry point of a penalty-only handle ever reads.
int frx_penalty_problem_create(const frx_config *cfg, int device, int B, const int *piece_n, const int *piece_poly, const int *h_off, const double *h_rec,
                               frx_problem **out) {
    if (!cfg || !piece_n || !piece_poly || !h_off || !h_rec || !out || B <= 0) return fail(FRX_ERR_INVALID_ARG, "frx_penalty_problem_create: null argument or B <= 0");
    *out = nullptr;
    size_t P = 0;
    for (int b = 0; b < B; b++) { if (piece_n[b] < 1) return fail(FRX_ERR_INVALID_ARG, "piece_n[b] < 1"); P += (size_t)piece_n[b]; }
    std::vector<int> hoff2(P + 1, 0), voff(1, 0);
    std::vector<double> hrec2, vrec, states((size_t)9 * B, 0.0);
    for (size_t gp = 0; gp < P; gp++) {
        const int m = piece_poly[gp];
        if (m < 0) return fail(FRX_ERR_INVALID_ARG, "piece_poly: negative polytope index");
        const int hb = h_off[m], K = h_off[m + 1] - hb;
        if (K < 1) return fail(FRX_ERR_INVALID_ARG, "an H-polytope has no half-spaces");
        hrec2.insert(hrec2.end(), h_rec + 6 * (size_t)hb, h_rec + 6 * (size_t)(hb + K));
        hoff2[gp + 1] = hoff2[gp] + K;
    }
    for (int b = 0; b < B; b++)
        for (int q = 0; q < 2 * piece_n[b] - 1; q++) {
            const double v2[6] = {0.0, 0.0, 0.0, 1.0, 0.0, 0.0};
            vrec.insert(vrec.end(), v2, v2 + 6);
            voff.push_back(voff.back() + 2);
        }
    frx_config c2 = *cfg;
    c2.grid_res = INFINITY;                                                 // one fine piece per polytope entry: idxHs is already expanded
    if (!(c2.rho > 0.0)) { c2.rho = 1.0; }                                  // (soft total time: the time layer is never evaluated on this handle, and it needs no TotalT)
    frx_problem *p = nullptr;
    const int rc = frx_problem_create(&c2, device, B, piece_n, states.data(), states.data(), hoff2.data(), hrec2.data(), voff.data(), vrec.data(), &p);
    if (rc != FRX_OK) return rc;
    p->penalty_only = true;
    *out = p;
    return FRX_OK;
}

int frx_problem_set_lbfgs_mode(frx_problem *p, int mode) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (mode != FRX_LBFGS_DEVICE_VECTORS && mode != FRX_LBFGS_HOST_VECTORS) return fail(FRX_ERR_INVALID_ARG, "unknown L-BFGS mode");
    p->lbfgs_mode = mode;
    return FRX_OK;
}

int frx_problem_set_solver(frx_problem *p, int solver) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (solver != FRX_SOLVER_KNOT_PCR && solver != FRX_SOLVER_BANDED_LU) return fail(FRX_ERR_INVALID_ARG, "unknown solver id");
    if (solver == FRX_SOLVER_BANDED_LU && !p->banded_ok)
        return fail(FRX_ERR_CAPACITY, "banded-LU kernels need the 6N x 13 band in LDS: too many pieces");
    p->geo.solver = solver;
    return FRX_OK;
}

// Diagnostic: one evaluation at x with s_memtime stamps at the phase boundaries of candidate 0's k_forward_knot
// (slots 0..6) and k_backward_knot (slots 16..24); out32 receives the raw shader-clock stamps.
int frx_profile_phases(frx_problem *p, const double *x, long long *out32) {
    if (!p || !x || !out32) return fail(FRX_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    if (!p->d_stamps.p) HIP_TRY(p->d_stamps.alloc(64));
    HIP_TRY(hipMemset(p->d_stamps.p, 0, 64 * sizeof(long long)));
    std::vector<double> f(p->B), g(p->NX);
    int rc = frx_objective_eval(p, x, f.data(), g.data());          // warm
    if (rc != FRX_OK) return rc;
    p->dp.stamps = p->d_stamps.p;
    rc = frx_objective_eval(p, x, f.data(), g.data());
    p->dp.stamps = nullptr;
    if (rc != FRX_OK) return rc;
    HIP_TRY(hipMemcpy(out32, p->d_stamps.p, 32 * sizeof(long long), hipMemcpyDeviceToHost));
    return FRX_OK;
}

// Diagnostic: one evaluation at x in the one-launch form with cycle stamps of cluster 0 (forward map 0..12 and adjoint 16..31 as frx_profile_phases; 40..43 the leader's entry,
// end of the forward map, end of the adjoint, end; 44..48 wave 0 of the first member: entry, gate seen, granules staged, samples done, partials out).
int frx_debug_profile_eval_cluster(frx_problem *p, const double *x, long long *out64) {
    if (!p || !x || !out64) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (!p->eval_fused) return fail(FRX_ERR_INVALID_ARG, "the one-launch evaluation does not apply to this handle");
    HIP_TRY(hipSetDevice(p->device));
    if (!p->d_stamps.p) HIP_TRY(p->d_stamps.alloc(64));
    HIP_TRY(hipMemset(p->d_stamps.p, 0, 64 * sizeof(long long)));
    std::vector<double> f(p->B), g(p->NX);
    int rc = frx_objective_eval(p, x, f.data(), g.data());          // warm
    if (rc != FRX_OK) return rc;
    p->dp.stamps = p->d_stamps.p; p->eval_fused_stamps = 1;
    rc = frx_objective_eval(p, x, f.data(), g.data());
    p->dp.stamps = nullptr; p->eval_fused_stamps = 0;
    if (rc != FRX_OK) return rc;
    HIP_TRY(hipMemcpy(out64, p->d_stamps.p, 64 * sizeof(long long), hipMemcpyDeviceToHost));
    return FRX_OK;
}

// Diagnostic (tests, tuning): drives k_lbfgs_pre alone.  `iters` successive accepted steps on random (x, g) sequences of
// length n for B candidates; after each the device search direction is compared with a plain host two-loop recursion over
// the same history (lbfgs.hpp:1381-1411, in the reference's order).  Returns the worst relative error (max-norm) and the
// mean duration in us of the last min(iters, 32) launches.  geom4 = {E, W, PF, BLK} or null for the default choice.
int frx_dv_selftest(int device, int n, int B, int m, int iters, const int *geom4, unsigned seed, double *max_rel_err, double *avg_us) {
    if (n < 1 || B < 1 || m < 1 || m > 512 || iters < 1 || !max_rel_err || !avg_us) return fail(FRX_ERR_INVALID_ARG, "bad selftest argument");
    HIP_TRY(hipSetDevice(device));
    int E = 0, W = 0, PF = 0, BLK = 4;
    frx::dv_geometry(n, &E, &W, &PF);
    if (geom4) { E = geom4[0]; W = geom4[1]; PF = geom4[2]; BLK = geom4[3]; }
    if (E == 0 || n > 64 * W * E) return fail(FRX_ERR_CAPACITY, "vector too long for k_lbfgs_pre");
    const char *tight_env = std::getenv("FRX_DV_TIGHT");                          // (read per call: the A/B of the row stride runs both in one process)
    const size_t HS = frx::dv_row_stride(n, E, W, !(tight_env && tight_env[0] == '0')), NX = (size_t)n * B;
    DevBuf<double> dx, dg, dxp, dgp, dd, dS, dY, dys, dgt; DevBuf<int> dxoff;
    PinBuf<frx::DvCommand> cmd; PinBuf<frx::DvResult> res;
    hipError_t e;
    if ((e = dx.alloc(NX)) || (e = dg.alloc(NX)) || (e = dxp.alloc(NX)) || (e = dgp.alloc(NX)) || (e = dd.alloc(NX)) || (e = dS.alloc(m * B * HS)) ||
        (e = dY.alloc(m * B * HS)) || (e = dys.alloc((size_t)B * m)) || (e = dgt.alloc((size_t)B * m * 4)) || (e = dxoff.alloc(B + 1)) || (e = cmd.alloc(B)) || (e = res.alloc(B)))
        return fail(FRX_ERR_ALLOC, hipGetErrorString(e));
    HIP_TRY(hipMemset(dS.p, 0, sizeof(double) * m * B * HS)); HIP_TRY(hipMemset(dY.p, 0, sizeof(double) * m * B * HS));
    HIP_TRY(hipMemset(dys.p, 0, sizeof(double) * B * m)); HIP_TRY(hipMemset(dgt.p, 0, sizeof(double) * B * m * 4));
    std::vector<int> xoff(B + 1);
    for (int b = 0; b <= B; b++) xoff[b] = b * n;
    HIP_TRY(hipMemcpy(dxoff.p, xoff.data(), sizeof(int) * (B + 1), hipMemcpyHostToDevice));
    frx::DvLaunch dv;
    dv.xoff = dxoff.p; dv.x = dx.p; dv.g = dg.p; dv.xp = dxp.p; dv.gp = dgp.p; dv.d = dd.p; dv.S = dS.p; dv.Y = dY.p; dv.ys = dys.p; dv.gt = dgt.p;
    dv.ld = NX; dv.hs = HS; dv.m = m; dv.B = B; dv.E = E; dv.W = W; dv.PF = PF; dv.BLK = BLK;
    unsigned long long st = 0x9E3779B97F4A7C15ull * (seed + 1);
    auto rnd = [&]() { st += 0x9E3779B97F4A7C15ull; unsigned long long z = st; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; return (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5; };
    std::vector<double> x(NX), g(NX), xp(NX), gp(NX), dref(n), ddev(NX);
    std::vector<std::vector<double>> S(B * (size_t)m, std::vector<double>(n)), Y(B * (size_t)m, std::vector<double>(n));
    std::vector<double> ysv((size_t)B * m), alpha(m);
    for (size_t i = 0; i < NX; i++) { x[i] = rnd(); g[i] = rnd(); }
    HIP_TRY(hipMemcpy(dx.p, x.data(), sizeof(double) * NX, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dg.p, g.data(), sizeof(double) * NX, hipMemcpyHostToDevice));
    for (int b = 0; b < B; b++) { cmd.p[b].flags = frx::DV_INIT; cmd.p[b].slot = cmd.p[b].bound = cmd.p[b].newest = 0; cmd.p[b].step = 0.0; }
    if (frx::launch_lbfgs_pre(dv, cmd.p, res.p, nullptr)) return fail(FRX_ERR_HIP, "k_lbfgs_pre launch (geometry not instantiated?)");
    HIP_TRY(hipDeviceSynchronize());
    xp = x; gp = g;
    HipEventPair evp; HIP_TRY(evp.create());
    const hipEvent_t e0 = evp.e0, e1 = evp.e1;
    double worst = 0.0, us_sum = 0.0; int us_n = 0, end = 0;
    for (int k = 1; k <= iters; k++) {
        // the new accepted point: x moves by a random step that correlates with the gradient change, so y.s > 0 mostly
        for (size_t i = 0; i < NX; i++) { const double sx = 0.1 * rnd(); x[i] = xp[i] + sx; g[i] = gp[i] + 3.0 * sx + 0.05 * rnd(); }
        HIP_TRY(hipMemcpy(dx.p, x.data(), sizeof(double) * NX, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dg.p, g.data(), sizeof(double) * NX, hipMemcpyHostToDevice));
        const int bound = m <= k ? m : k;
        for (int b = 0; b < B; b++) { cmd.p[b].flags = frx::DV_ADVANCE; cmd.p[b].slot = end; cmd.p[b].bound = bound; cmd.p[b].newest = end; cmd.p[b].step = 0.0; }
        HIP_TRY(hipEventRecord(e0, nullptr));
        if (frx::launch_lbfgs_pre(dv, cmd.p, res.p, nullptr)) return fail(FRX_ERR_HIP, "k_lbfgs_pre launch");
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (k > iters - 32) { us_sum += 1e3 * ms; us_n++; }
        HIP_TRY(hipMemcpy(ddev.data(), dd.p, sizeof(double) * NX, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; b++) {
            double *sx = S[(size_t)b * m + end].data(), *yx = Y[(size_t)b * m + end].data();
            double ys = 0.0, yy = 0.0;
            for (int i = 0; i < n; i++) { sx[i] = x[b * (size_t)n + i] - xp[b * (size_t)n + i]; yx[i] = g[b * (size_t)n + i] - gp[b * (size_t)n + i]; ys += yx[i] * sx[i]; yy += yx[i] * yx[i]; }
            ysv[(size_t)b * m + end] = ys;
            for (int i = 0; i < n; i++) dref[i] = -g[b * (size_t)n + i];
            int j = (end + 1) % m;
            for (int i = 0; i < bound; i++) {
                j = (j + m - 1) % m;
                const double *sj = S[(size_t)b * m + j].data(), *yj = Y[(size_t)b * m + j].data();
                double a = 0.0; for (int q = 0; q < n; q++) a += sj[q] * dref[q];
                a /= ysv[(size_t)b * m + j]; alpha[j] = a;
                for (int q = 0; q < n; q++) dref[q] -= a * yj[q];
            }
            for (int q = 0; q < n; q++) dref[q] *= ys / yy;
            for (int i = 0; i < bound; i++) {
                const double *sj = S[(size_t)b * m + j].data(), *yj = Y[(size_t)b * m + j].data();
                double be = 0.0; for (int q = 0; q < n; q++) be += yj[q] * dref[q];
                be /= ysv[(size_t)b * m + j];
                for (int q = 0; q < n; q++) dref[q] += (alpha[j] - be) * sj[q];
                j = (j + 1) % m;
            }
            double num = 0.0, den = 0.0, dgr = 0.0;
            for (int q = 0; q < n; q++) { num = std::max(num, std::fabs(dref[q] - ddev[b * (size_t)n + q])); den = std::max(den, std::fabs(dref[q])); dgr += dref[q] * g[b * (size_t)n + q]; }
            worst = std::max(worst, num / den);
            worst = std::max(worst, std::fabs(res.p[b].dginit - dgr) / std::max(std::fabs(dgr), 1e-300));
        }
        xp = x; gp = g; end = (end + 1) % m;
    }
    *max_rel_err = worst; *avg_us = us_n ? us_sum / us_n : 0.0;
    return FRX_OK;
}

// Diagnostic (bench): average duration of each stage kernel of an evaluation at x, HIP events on the handle's stream around `reps`
// back-to-back launches of ONE kernel at a time (the other stages run once before, so every kernel sees valid inputs).
int frx_eval_stage_times(frx_problem *p, const double *x, int reps, double *out3_us) {
    if (!p || !x || !out3_us || reps < 1) return fail(FRX_ERR_INVALID_ARG, "null argument or reps < 1");
    HIP_TRY(hipSetDevice(p->device));
    std::vector<double> f(p->B), g(p->NX);
    int rc;
    {   // (the stage kernels, not the one-launch form: their buffers are what the timed launches below read)
        struct Stage { frx_problem *q; int was, was_solo; ~Stage() { q->eval_fused = was; q->eval_solo = was_solo; } } stage{p, p->eval_fused, p->eval_solo};
        p->eval_fused = 0; p->eval_solo = 0;                             // (neither one-launch form leaves (T, C), the multipliers and the waypoint sums in global memory)
        rc = frx_objective_eval(p, x, f.data(), g.data());               // d_x, d_T, d_C, d_out20, pcrw all valid afterwards
    }
    if (rc != FRX_OK) return rc;
    HipEventPair evp; HIP_TRY(evp.create());
    const hipEvent_t e0 = evp.e0, e1 = evp.e1;
    auto time_it = [&](int which, double &us) -> int {
        for (int w = 0; w < 3; w++) {                                    // warm
            hipError_t e = hipSuccess;
            if (which == 0) e = (hipError_t)frx::launch_forward(p->dp, p->geo, p->d_x.p, p->d_T.p, p->d_C.p, p->d_band.p, p->stream);
            if (which == 1) e = (hipError_t)frx::launch_penalty(p->dp, p->geo, p->d_T.p, p->d_C.p, p->d_out20.p, p->stream);
            if (which == 2) e = (hipError_t)frx::launch_backward(p->dp, p->geo, p->d_x.p, p->d_T.p, p->d_C.p, p->d_band.p, p->d_out20.p, p->d_f.p, p->d_g.p, p->stream);
            if (e != hipSuccess) return fail(FRX_ERR_HIP, hipGetErrorString(e));
        }
        if (hipEventRecord(e0, p->stream) != hipSuccess) return fail(FRX_ERR_HIP, "hipEventRecord");
        for (int r = 0; r < reps; r++) {
            if (which == 0) (void)frx::launch_forward(p->dp, p->geo, p->d_x.p, p->d_T.p, p->d_C.p, p->d_band.p, p->stream);
            if (which == 1) (void)frx::launch_penalty(p->dp, p->geo, p->d_T.p, p->d_C.p, p->d_out20.p, p->stream);
            if (which == 2) (void)frx::launch_backward(p->dp, p->geo, p->d_x.p, p->d_T.p, p->d_C.p, p->d_band.p, p->d_out20.p, p->d_f.p, p->d_g.p, p->stream);
        }
        if (hipEventRecord(e1, p->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return fail(FRX_ERR_HIP, "hipEventSynchronize");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return fail(FRX_ERR_HIP, "hipEventElapsedTime");
        us = 1e3 * ms / reps;
        return FRX_OK;
    };
    for (int k = 0; k < 3; k++) if ((rc = time_it(k, out3_us[k])) != FRX_OK) break;
    return rc;
}

// Diagnostic (bench, tests): the one-launch evaluation.  set: 1 = use it where it applies (the default), 0 = always three stage launches; returns FRX_OK.
// frx_debug_eval_fused: workgroups per candidate of the form frx_objective_eval[_device] takes right now, 0 = one launch per stage.
int frx_debug_set_eval_fused(frx_problem *p, int enable) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (enable && p->eval_fused_G && (p->eval_fused_code || (p->h_ev_status.p && *(volatile unsigned *)p->h_ev_status.p))) (void)eval_cluster_status(p, nullptr);   // a retired form comes back clean
    p->eval_fused = (enable && p->eval_fused_G) ? 1 : 0;
    p->eval_fused_ticks = enable == 2 ? 1ull : 25000000ull;               // 2 (tests): the value 1 tells the launcher to drop the members and bound the leader's waits at 50 us - the failure path
    return FRX_OK;
}
int frx_debug_eval_fused(const frx_problem *p) { return (p && p->eval_fused) ? p->eval_fused_G : 0; }
// Diagnostic (bench, tests): the solo form of an evaluation (one workgroup per candidate, one launch).  mode: 0 = never, 1 = from the handle's threshold on (the
// default), 2 = at every batch size.  frx_debug_eval_solo: workgroups of the kernel a CU holds when the NEXT evaluation of this handle takes the form, 0 = it does not.
int frx_debug_set_eval_solo(frx_problem *p, int mode) {
    if (!p || mode < 0 || mode > 2) return fail(FRX_ERR_INVALID_ARG, "null argument or unknown mode");
    if (mode == 2 && !p->geo.lds_solo) return fail(FRX_ERR_INVALID_ARG, "the solo form does not apply to this handle (more than 64 pieces or samples per piece, or the banded solver)");   // (1 = where it applies: always accepted)
    p->eval_solo = mode;
    return FRX_OK;
}
// Diagnostic (bench): which penalty kernel a stage launch of this handle takes - 0 = k_penalty (FRX_PENALTY_FORM=thr), 1 = k_penalty_lat, 2 = k_penalty_lat2 (four-wave
// workgroups, one sample per lane: the large-batch form; FRX_PENALTY_TWOPHASE=0 keeps the one-phase launch).  Mirrors launch_penalty (frx_device.hip).
int frx_debug_penalty_kernel(const frx_problem *p) {
    if (!p) return -1;
    const char *f = std::getenv("FRX_PENALTY_FORM");
    if (f && f[0] != 'l') return 0;
    const char *tp = std::getenv("FRX_PENALTY_TWOPHASE");
    return (p->geo.lds_pen2 && !(tp && tp[0] == '0')) ? 2 : 1;
}
int frx_debug_eval_solo(const frx_problem *p) {
    if (!p || !p->geo.lds_solo || p->geo.solver != frx::SOLVER_KNOT_PCR) return 0;
    if (!(p->eval_solo == 2 || (p->eval_solo == 1 && p->B >= p->eval_solo_min_B && p->B <= p->eval_solo_max_B))) return 0;
    if (p->eval_fused && p->eval_solo != 2) return 0;                          // (a plain evaluation of a batch the chip holds as clusters takes that form first)
    return frx::eval_solo_blocks_per_cu(p->geo);
}
// Diagnostic (bench): average duration of one evaluation at x in the form frx_objective_eval_device takes, HIP events around `reps` back-to-back evaluations.
int frx_eval_launch_time(frx_problem *p, const double *x, int reps, double *out_us) {
    if (!p || !x || !out_us || reps < 1) return fail(FRX_ERR_INVALID_ARG, "null argument or reps < 1");
    HIP_TRY(hipSetDevice(p->device));
    std::vector<double> f(p->B), g(p->NX);
    int rc = frx_objective_eval(p, x, f.data(), g.data());
    if (rc != FRX_OK) return rc;
    HipEventPair evp; HIP_TRY(evp.create());
    for (int w = 0; w < 3; w++) HIP_TRY((hipError_t)launch_eval(p, p->d_x.p, p->d_f.p, p->d_g.p, p->stream, true));
    HIP_TRY(hipEventRecord(evp.e0, p->stream));
    for (int r = 0; r < reps; r++) (void)launch_eval(p, p->d_x.p, p->d_f.p, p->d_g.p, p->stream, true);
    HIP_TRY(hipEventRecord(evp.e1, p->stream));
    HIP_TRY(hipEventSynchronize(evp.e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, evp.e0, evp.e1));
    *out_us = 1e3 * ms / reps;
    HIP_TRY(hipMemcpy(f.data(), p->d_f.p, sizeof(double) * p->B, hipMemcpyDeviceToHost));
    return eval_cluster_status(p, f.data());
}

int frx_problem_totals(const frx_problem *p, int *out6) {
    if (!p || !out6) return fail(FRX_ERR_INVALID_ARG, "null argument");
    out6[0] = p->B; out6[1] = p->P; out6[2] = p->Pc; out6[3] = p->NX; out6[4] = p->Kmax; out6[5] = p->sumKfine;
    return FRX_OK;
}
int frx_problem_layout(const frx_problem *p, int *piece_off, int *coarse_off, int *x_off, int *dim_t) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (piece_off) std::copy(p->poff.begin(), p->poff.end(), piece_off);
    if (coarse_off) std::copy(p->coff.begin(), p->coff.end(), coarse_off);
    if (x_off) std::copy(p->xoff.begin(), p->xoff.end(), x_off);
    if (dim_t) std::copy(p->dimT.begin(), p->dimT.end(), dim_t);
    return FRX_OK;
}

int frx_initial_guess(frx_problem *p, double *x0) {
    if (!p || !x0) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (p->penalty_only) return fail(FRX_ERR_INVALID_ARG, "this handle serves frx_penalty_eval[_device] only (frx_penalty_problem_create)");
    frx::initial_guess_batch(p->cfg, p->softT, p->cand, p->xoff.data(), x0);     // one flat task list over (candidate, waypoint)
    return FRX_OK;
}

int frx_objective_eval_device(frx_problem *p, const double *x_dev, double *f_dev, double *g_dev, void *hip_stream) {
    if (!p || !x_dev || !f_dev || !g_dev) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (p->penalty_only) return fail(FRX_ERR_INVALID_ARG, "this handle serves frx_penalty_eval[_device] only (frx_penalty_problem_create)");
    HIP_TRY((hipError_t)launch_eval(p, x_dev, f_dev, g_dev, hip_stream, true));
    return FRX_OK;
}

int frx_eval_status(frx_problem *p) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    return eval_cluster_status(p, nullptr);
}

int frx_objective_eval(frx_problem *p, const double *x, double *f, double *g) {
    if (!p || !x || !f || !g) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (p->penalty_only) return fail(FRX_ERR_INVALID_ARG, "this handle serves frx_penalty_eval[_device] only (frx_penalty_problem_create)");
    HIP_TRY(hipSetDevice(p->device));
    std::memcpy(p->h_x.p, x, sizeof(double) * p->NX);
    HIP_TRY(hipMemcpyAsync(p->d_x.p, p->h_x.p, sizeof(double) * p->NX, hipMemcpyHostToDevice, p->stream));
    HIP_TRY((hipError_t)launch_eval(p, p->d_x.p, p->d_f.p, p->d_g.p, p->stream, true));
    HIP_TRY(hipMemcpyAsync(p->h_f.p, p->d_f.p, sizeof(double) * p->B, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(p->h_g.p, p->d_g.p, sizeof(double) * p->NX, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    std::memcpy(f, p->h_f.p, sizeof(double) * p->B);
    std::memcpy(g, p->h_g.p, sizeof(double) * p->NX);
    return eval_cluster_status(p, f);
}

// Asynchronous form of frx_objective_eval for host buffers: returns once the copies and kernels are enqueued on the handle's
// stream; frx_wait() completes it and fills f, g (which must stay valid until then).  One evaluation in flight per handle.
int frx_objective_eval_async(frx_problem *p, const double *x, double *f, double *g) {
    if (!p || !x || !f || !g) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (p->penalty_only) return fail(FRX_ERR_INVALID_ARG, "this handle serves frx_penalty_eval[_device] only (frx_penalty_problem_create)");
    if (p->pending_f) return fail(FRX_ERR_INVALID_ARG, "an asynchronous evaluation is already in flight on this handle (call frx_wait)");
    HIP_TRY(hipSetDevice(p->device));
    std::memcpy(p->h_x.p, x, sizeof(double) * p->NX);
    HIP_TRY(hipMemcpyAsync(p->d_x.p, p->h_x.p, sizeof(double) * p->NX, hipMemcpyHostToDevice, p->stream));
    HIP_TRY((hipError_t)launch_eval(p, p->d_x.p, p->d_f.p, p->d_g.p, p->stream, true));
    HIP_TRY(hipMemcpyAsync(p->h_f.p, p->d_f.p, sizeof(double) * p->B, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(p->h_g.p, p->d_g.p, sizeof(double) * p->NX, hipMemcpyDeviceToHost, p->stream));
    p->pending_f = f; p->pending_g = g;
    return FRX_OK;
}
int frx_wait(frx_problem *p) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->pending_f) {
        std::memcpy(p->pending_f, p->h_f.p, sizeof(double) * p->B);
        std::memcpy(p->pending_g, p->h_g.p, sizeof(double) * p->NX);
        const double *fdone = p->pending_f;
        p->pending_f = nullptr; p->pending_g = nullptr;
        return eval_cluster_status(p, fdone);
    }
    return FRX_OK;
}

int frx_penalty_eval_device(frx_problem *p, const double *T_dev, const double *C_dev, double *out_dev, void *hip_stream) {
    if (!p || !T_dev || !C_dev || !out_dev) return fail(FRX_ERR_INVALID_ARG, "null argument");
    HIP_TRY((hipError_t)frx::launch_penalty(p->dp, p->geo, T_dev, C_dev, out_dev, hip_stream));
    return FRX_OK;
}

int frx_penalty_eval(frx_problem *p, const double *T, const double *C, double *cost, double *gdT, double *gdC) {
    if (!p || !T || !C || !cost || !gdT || !gdC) return fail(FRX_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    std::memcpy(p->h_T.p, T, sizeof(double) * p->P);
    std::memcpy(p->h_C.p, C, sizeof(double) * 18 * (size_t)p->P);
    HIP_TRY(hipMemcpyAsync(p->d_T.p, p->h_T.p, sizeof(double) * p->P, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_C.p, p->h_C.p, sizeof(double) * 18 * (size_t)p->P, hipMemcpyHostToDevice, p->stream));
    int rc = frx_penalty_eval_device(p, p->d_T.p, p->d_C.p, p->d_out20.p, p->stream);
    if (rc != FRX_OK) return rc;
    HIP_TRY(hipMemcpyAsync(p->h_out20.p, p->d_out20.p, sizeof(double) * 20 * (size_t)p->P, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    // accumulate, like cuda_computer::compute (cc.cu:549-559)
    for (int b = 0; b < p->B; b++) {
        double s = 0.0;
        for (int gp = p->poff[b]; gp < p->poff[b + 1]; gp++) {
            const double *o = p->h_out20.p + 20 * (size_t)gp;
            s += o[0];
            gdT[gp] += o[1];
            for (int v = 0; v < 18; v++) gdC[18 * (size_t)gp + v] += o[2 + v];
        }
        cost[b] += s;
    }
    return FRX_OK;
}

int frx_forward(frx_problem *p, const double *x, double *T, double *C) {
    if (!p || !x) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (p->penalty_only) return fail(FRX_ERR_INVALID_ARG, "this handle serves frx_penalty_eval[_device] only (frx_penalty_problem_create)");
    HIP_TRY(hipSetDevice(p->device));
    std::memcpy(p->h_x.p, x, sizeof(double) * p->NX);
    HIP_TRY(hipMemcpyAsync(p->d_x.p, p->h_x.p, sizeof(double) * p->NX, hipMemcpyHostToDevice, p->stream));
    HIP_TRY((hipError_t)launch_eval(p, p->d_x.p, nullptr, nullptr, p->stream, false));
    HIP_TRY(hipMemcpyAsync(p->h_T.p, p->d_T.p, sizeof(double) * p->P, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(p->h_C.p, p->d_C.p, sizeof(double) * 18 * (size_t)p->P, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (T) std::memcpy(T, p->h_T.p, sizeof(double) * p->P);
    if (C) std::memcpy(C, p->h_C.p, sizeof(double) * 18 * (size_t)p->P);
    return FRX_OK;
}

} // extern "C"

// The batched driver shared by frx_optimize (device evaluator) and frx_lbfgs_minimize_batch (callback evaluator).
template <class EvalAll>
static int drive_batch(int count, const int *x_off, double *x, double *g, double *f, const frx_lbfgs_params &pm, int n_threads,
                       int *status, int *iters, int *evals, double *f_out, double *stats, EvalAll &&eval_all) {
    std::vector<frx::Solver> sv(count);
    frx::SpinPool pool(std::min(n_threads, count));
    // allocation + first touch of each solver's history by its owning (pinned) worker
    pool.run(count, [&](int i) { sv[i].start(x_off[i + 1] - x_off[i], x + x_off[i], g + x_off[i], pm); });
    std::vector<int> active;
    active.reserve(count);
    double t_eval = 0.0, t_host = 0.0;
    long rounds = 0;
    const auto t0 = clk::now();
    for (;;) {
        active.clear();
        for (int i = 0; i < count; i++)
            if (!sv[i].done()) active.push_back(i);
        if (active.empty()) break;
        auto te = clk::now();
        int rc = eval_all((int)active.size(), active.data());
        if (rc != FRX_OK) return rc;
        t_eval += ms_since(te);
        auto th = clk::now();
        pool.run(count, [&](int i) { if (!sv[i].done()) sv[i].feed(f[i]); });
        t_host += ms_since(th);
        rounds++;
    }
    if (stats) { stats[0] = ms_since(t0); stats[1] = t_eval; stats[2] = t_host; stats[3] = (double)rounds; }
    for (int i = 0; i < count; i++) {
        if (status) status[i] = sv[i].status();
        if (iters) iters[i] = sv[i].iterations();
        if (evals) evals[i] = sv[i].evaluations();
        if (f_out) f_out[i] = sv[i].value();
    }
    return FRX_OK;
}

static int finish_optimize(frx_problem *p, const double *x, double *C, double *T, double *jerk_cost);

// Device-vector optimisation: the host runs one SolverDV (scalars + decisions) per candidate, the device owns every
// vector.  A round = k_lbfgs_pre (history update + two-loop recursion for candidates that just accepted a step, then the
// trial point) -> k_forward -> k_penalty -> k_backward -> k_lbfgs_post; 32 B of commands go down and 40 B of results
// come back per candidate, through mapped host memory.
// `only` (optional, [B]): candidates with only[b] == 0 take no part - their x, status, counters and objective are left untouched
// (used to re-run, on this path, candidates that failed on the resident kernel).
// Stragglers of a per-stage batch that continue on the resident round kernel.  A per-stage round costs four launches whose first is a
// one-CU recursion over the candidate's history (~85 us per round however few candidates are left), a resident round ~25 us - and a batch waits for its
// slowest candidate (one infeasible scenario that wanders 16 673 evaluations held 511 finished ones hostage for a second, profiles/r04_bench_montecarlo4096.json).
// When no more than `capacity` candidates are still running, optimize_device_vectors stops and hands them over AS THEY ARE: their solver state machines
// (the host's side of the plan: line search, stop tests, counters), their pending commands, and where their history stands on the device.
struct TakeOver {
    int capacity = 0;                     // in: hand over when at most this many candidates are still running (0: never)
    long not_before = 1;                  // in: ... and this many rounds have run (FRX_MIGRATE_AT: tests force an early hand-over)
    std::vector<int> cand;                // out: the candidates, in cluster order
    std::vector<frx::SolverDV> sv;        // out: their solvers (pending command inside)
    std::vector<int> newest, bound;       // out: newest slot / number of pairs of their history rows on the device
    std::vector<double> f_last;           // out: objective value of their last evaluation
    double ms = 0.0, ms_dev = 0.0, ms_host = 0.0; long rounds = 0;   // out: what the per-stage part took
};

static int optimize_device_vectors(frx_problem *p, const frx_lbfgs_params &pm, double *x, int *status, int *iters, int *evals,
                                   double *objective, const std::vector<char> *only = nullptr, TakeOver *take = nullptr) {
    const int B = p->B, m = pm.mem_size;
    // k_lbfgs_pre geometry (frx::dv_geometry).  FRX_DV_GEOM=E,W,PF,BLK overrides (experiments).
    int E = 0, W = 0, PF = 0, BLK = 4;
    frx::dv_geometry(p->geo.maxXb, &E, &W, &PF);
    if (const char *gs = std::getenv("FRX_DV_GEOM")) {
        int e, w, pf, blk;
        if (std::sscanf(gs, "%d,%d,%d,%d", &e, &w, &pf, &blk) == 4 && p->geo.maxXb <= 64 * w * e) { E = e; W = w; PF = pf; BLK = blk; }
    }
    if (E == 0 || m > 512 || m < 1) return 1;                                  // caller falls back to host vectors
    // rows as long as the longest candidate's vector needs (FRX_DV_TIGHT=0: the full 64 W E of round 5 - A/B)
    const char *tight_env = std::getenv("FRX_DV_TIGHT");
    const size_t HS = frx::dv_row_stride(p->geo.maxXb, E, W, !(tight_env && tight_env[0] == '0'));
    hipError_t e;
    if (p->dv_mem != m || p->dv_hs != HS) {
        p->dv_mem = 0; p->dv_hs = 0;
        const size_t hist = (size_t)m * B * HS;
        auto need = [](auto &buf, size_t count) -> hipError_t { return buf.n >= count && buf.p ? hipSuccess : buf.alloc(count); };   // keep what is large enough
        NumaScope numa_alloc(p->device);                                          // command / result mailboxes of the per-stage rounds: pages from the device's NUMA node
        if ((e = need(p->d_xp, p->NX)) != hipSuccess || (e = need(p->d_gp, p->NX)) != hipSuccess || (e = need(p->d_dir, p->NX)) != hipSuccess ||
            (e = need(p->d_S, hist)) != hipSuccess || (e = need(p->d_Y, hist)) != hipSuccess ||
            (e = need(p->d_ys, (size_t)B * m)) != hipSuccess || (e = need(p->d_gt, (size_t)B * m * 4)) != hipSuccess || (e = need(p->h_cmd, B)) != hipSuccess || (e = need(p->h_res, B)) != hipSuccess) {
            (void)hipGetLastError();
            if (e == hipErrorOutOfMemory) {                                       // the history does not fit: the host-vector solver needs no device history
                p->d_S.release(); p->d_Y.release();
                return 1;
            }
            return fail(FRX_ERR_ALLOC, std::string("device-vector L-BFGS buffers: ") + hipGetErrorString(e));
        }
        // zero padding of the history slices is relied upon by the unconditional 16-byte loads of k_lbfgs_pre; on the handle's
        // stream, like every consumer (the stream is non-blocking: the null stream would not order against it)
        if ((e = hipMemsetAsync(p->d_S.p, 0, sizeof(double) * hist, p->stream)) != hipSuccess || (e = hipMemsetAsync(p->d_Y.p, 0, sizeof(double) * hist, p->stream)) != hipSuccess)
            return fail(FRX_ERR_HIP, hipGetErrorString(e));
        p->dv_mem = m; p->dv_hs = HS;
    }
    frx::DvLaunch dv;
    dv.xoff = p->d_xoff.p; dv.x = p->d_x.p; dv.g = p->d_g.p; dv.xp = p->d_xp.p; dv.gp = p->d_gp.p; dv.d = p->d_dir.p;
    dv.S = p->d_S.p; dv.Y = p->d_Y.p; dv.ys = p->d_ys.p; dv.gt = p->d_gt.p; dv.ld = (size_t)p->NX; dv.hs = HS; dv.m = m; dv.B = B; dv.E = E; dv.W = W; dv.PF = PF; dv.BLK = BLK;
    std::memcpy(p->h_x.p, x, sizeof(double) * p->NX);
    HIP_TRY(hipMemcpyAsync(p->d_x.p, p->h_x.p, sizeof(double) * p->NX, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemsetAsync(p->d_ys.p, 0, sizeof(double) * (size_t)B * m, p->stream));
    HIP_TRY(hipMemsetAsync(p->d_gt.p, 0, sizeof(double) * (size_t)B * m * 4, p->stream));
    std::vector<frx::SolverDV> sv(B);
    for (int b = 0; b < B; b++) {
        sv[b].start(p->xoff[b + 1] - p->xoff[b], pm, p->h_cmd.p + b);
        if (only && !(*only)[b]) sv[b].give_up(0);                              // not part of this run
    }
    // every wait is bounded ("status codes, bounded waits"): FRX_ROUND_TIMEOUT_MS, default 5 s per round
    const double round_timeout_ms = [] { const char *e = std::getenv("FRX_ROUND_TIMEOUT_MS"); const double v = e ? std::atof(e) : 0.0; return v > 0.0 ? v : 5000.0; }();
    auto wait_stream = [&]() -> int {
        const auto tw = clk::now();
        for (unsigned spins = 1;; spins++) {
            hipError_t q = hipStreamQuery(p->stream);
            if (q == hipSuccess) return FRX_OK;
            if (q != hipErrorNotReady) return fail(FRX_ERR_HIP, std::string("device round failed: ") + hipGetErrorString(q));
            if ((spins & 0x3FF) == 0 && ms_since(tw) > round_timeout_ms) return fail(FRX_ERR_TIMEOUT, "device round did not complete within FRX_ROUND_TIMEOUT_MS");
            __builtin_ia32_pause();
        }
    };
    // with the knot/PCR kernels the reductions the line search needs ride on k_backward_knot (LineSearchTap); the banded-LU
    // kernels keep the separate k_lbfgs_post
    const bool fused_post = p->geo.solver == frx::SOLVER_KNOT_PCR;
    struct TapGuard { frx_problem *q; ~TapGuard() { q->tap_d = nullptr; q->tap_flags = nullptr; q->tap_res = nullptr; q->dp.cand_active = nullptr; q->dp.piece_active = nullptr; q->tap_arrive = nullptr; q->tap_flag = nullptr; } } tap_guard{p};
    const char *mb_env = std::getenv("FRX_MAILBOX");
    const bool mailbox = fused_post && !(mb_env && mb_env[0] == '0');
    if (fused_post) {
        if (!p->d_flags.p && (e = p->d_flags.alloc(B)) != hipSuccess) return fail(FRX_ERR_ALLOC, hipGetErrorString(e));
        dv.dflags = p->d_flags.p;
        p->tap_d = p->d_dir.p; p->tap_flags = p->d_flags.p; p->tap_res = p->h_res.p;
        // large batches: the objective kernels skip candidates that are not being evaluated this round (finished ones above all)
        const char *sk = std::getenv("FRX_SKIP_INACTIVE");
        if (sk ? sk[0] != '0' : B > 64) {
            if (!p->d_pflags.p && (e = p->d_pflags.alloc(p->P)) != hipSuccess) return fail(FRX_ERR_ALLOC, hipGetErrorString(e));
            dv.pflags = p->d_pflags.p; dv.poff = p->d_poff.p;
            p->dp.cand_active = p->d_flags.p; p->dp.piece_active = p->d_pflags.p;
        }
    }
    if (mailbox) {
        NumaScope numa_alloc(p->device);
        if (!p->d_arrive.p && ((e = p->d_arrive.alloc(1)) != hipSuccess || (e = p->h_flag.alloc(1)) != hipSuccess)) return fail(FRX_ERR_ALLOC, hipGetErrorString(e));
        HIP_TRY(hipMemsetAsync(p->d_arrive.p, 0, sizeof(unsigned), p->stream));
        p->h_flag.p[0] = 0;
        p->tap_arrive = p->d_arrive.p; p->tap_flag = p->h_flag.p; p->tap_round = 0;
    }
    // completion of a round: the last workgroup of k_backward_knot posts the round number in mapped memory (LineSearchTap).
    // The spin is bounded: a stream that drains without the post (arrival counter out of step, e.g. a launch that never
    // happened) is reported at once, anything else after the deadline; the counters are re-initialised by the next call.
    auto wait_round = [&](bool evaluated) -> int {
        if (!mailbox || !evaluated) return wait_stream();
        const auto tw = clk::now();
        for (unsigned spins = 1;; spins++) {
            if (*(volatile unsigned *)p->h_flag.p == p->tap_round) return FRX_OK;
            if ((spins & 0x3FFF) == 0) {
                const hipError_t q = hipStreamQuery(p->stream);
                if (q == hipSuccess) {
                    if (*(volatile unsigned *)p->h_flag.p == p->tap_round) return FRX_OK;
                    return fail(FRX_ERR_TIMEOUT, "round " + std::to_string(p->tap_round) + ": the stream drained but completion was never posted (arrival counter out of step)");
                }
                if (q != hipErrorNotReady) return fail(FRX_ERR_HIP, std::string("device round failed: ") + hipGetErrorString(q));
                if (ms_since(tw) > round_timeout_ms) return fail(FRX_ERR_TIMEOUT, "round " + std::to_string(p->tap_round) + " did not complete within FRX_ROUND_TIMEOUT_MS");
            }
            __builtin_ia32_pause();
        }
    };
    // fault injection (tests): FRX_DEBUG_DROP_ROUND=k leaves k_backward_knot out of round k, so that round's completion is never posted
    const long drop_round = [] { const char *e = std::getenv("FRX_DEBUG_DROP_ROUND"); return e ? std::atol(e) : -1L; }();
    double t_dev = 0.0, t_host = 0.0;
    long rounds = 0;
    const bool tracing = std::getenv("FRX_TRACE") != nullptr;
    p->trace.clear();
    std::vector<int> h_newest(B, 0), h_bound(B, 0);                            // where every candidate's history stands on the device (for a hand-over)
    std::vector<double> h_flast(B, 0.0);
    bool handed_over = false;
    const auto t0 = clk::now();
    for (;;) {
        bool any_eval = false, any_cmd = false;
        int running = 0;
        for (int b = 0; b < B; b++) { any_eval |= (p->h_cmd.p[b].flags & frx::DV_EVAL) != 0; any_cmd |= p->h_cmd.p[b].flags != 0; running += p->h_cmd.p[b].flags != 0; }
        if (!any_cmd) break;
        if (take && take->capacity > 0 && running <= take->capacity && rounds >= take->not_before) { handed_over = true; break; }   // the rest continues on the resident kernel
        for (int b = 0; b < B; b++)
            if (p->h_cmd.p[b].flags & frx::DV_ADVANCE) { h_newest[b] = p->h_cmd.p[b].slot; h_bound[b] = p->h_cmd.p[b].bound; }    // the pair this round stores
        auto td = clk::now();
        HIP_TRY((hipError_t)frx::launch_lbfgs_pre(dv, p->h_cmd.p, p->h_res.p, p->stream));
        if (any_eval) {
            if (mailbox) p->tap_round++;
            if (rounds == drop_round) HIP_TRY((hipError_t)launch_eval(p, p->d_x.p, p->d_f.p, p->d_g.p, p->stream, false));
            else HIP_TRY((hipError_t)launch_eval(p, p->d_x.p, p->d_f.p, p->d_g.p, p->stream, true));
            if (!fused_post) HIP_TRY((hipError_t)frx::launch_lbfgs_post(dv, p->d_f.p, p->h_cmd.p, p->h_res.p, p->stream));
        }
        { const int wrc = wait_round(any_eval); if (wrc != FRX_OK) { const std::string keep = g_err; (void)wait_stream(); g_err = keep; return wrc; } }   // bounded drain, first error kept
        t_dev += ms_since(td);
        auto th = clk::now();
        for (int b = 0; b < B; b++) {
            frx::DvCommand &c = p->h_cmd.p[b];
            const bool evaluated = (c.flags & frx::DV_EVAL) != 0;
            if (!evaluated) { c.flags = 0; continue; }                         // a RESTORE has been executed
            if (tracing && b == 0) { const frx::DvResult &r = p->h_res.p[b]; const double row[7] = {(double)c.flags, c.step, r.f, r.dg, r.dginit, r.xx, r.gg}; p->trace.insert(p->trace.end(), row, row + 7); }
            h_flast[b] = p->h_res.p[b].f;
            if (sv[b].saw_nonfinite(p->h_res.p[b].f)) sv[b].give_up(frx::LBERR_ROUNDING); else sv[b].feed(p->h_res.p[b]);
        }
        t_host += ms_since(th);
        rounds++;
    }
    p->stats[0] = ms_since(t0); p->stats[1] = t_dev; p->stats[2] = t_host; p->stats[3] = (double)rounds;
    HIP_TRY(hipMemcpyAsync(p->h_x.p, p->d_x.p, sizeof(double) * p->NX, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (handed_over) { take->ms = p->stats[0]; take->ms_dev = t_dev; take->ms_host = t_host; take->rounds = rounds; }
    for (int b = 0; b < B; b++) {
        if (only && !(*only)[b]) continue;
        if (handed_over && p->h_cmd.p[b].flags != 0) {                          // still running: the solver, its pending command and the history's position travel
            take->cand.push_back(b); take->sv.push_back(sv[b]); take->newest.push_back(h_newest[b]); take->bound.push_back(h_bound[b]); take->f_last.push_back(h_flast[b]);
            continue;
        }
        std::memcpy(x + p->xoff[b], p->h_x.p + p->xoff[b], sizeof(double) * (p->xoff[b + 1] - p->xoff[b]));
        status[b] = sv[b].status();
        if (iters) iters[b] = sv[b].iterations();
        if (evals) evals[b] = sv[b].evaluations();
        if (objective) objective[b] = sv[b].value();
    }
    return handed_over ? 2 : FRX_OK;
}


// Resident optimisation (frx_round_kernel.hpp): ONE launch per plan.  The host runs the same SolverDV state machines as
// optimize_device_vectors, but talks to each candidate's cluster through its own mailbox and never waits for the batch: a candidate's
// next command is posted the moment its result arrives.  Returns FRX_OK, a negative frx_status, or 1 = not applicable / the device
// gave up (resident_status says why): the caller then runs the one-launch-per-stage path from the untouched x.
// A resident launch needs ALL of its workgroups on the chip at once (one per CU: ~154 KB of LDS each).  Two launches that share a
// device - two shards of frx_multi on one GPU, two user threads - could each get a partially resident grid, and neither census would
// ever complete.  Launches of this process are therefore serialised per device: the second plan waits for the first one's kernel to
// leave (another PROCESS on the same device is caught by the census bound instead: 250 ms, then the per-stage path).
static std::mutex &resident_device_lock(int device) {
    static std::mutex table_lock;
    static std::map<int, std::unique_ptr<std::mutex>> table;
    std::lock_guard<std::mutex> g(table_lock);
    std::unique_ptr<std::mutex> &m = table[device];
    if (!m) m.reset(new std::mutex());
    return *m;
}

// `take` (optional): not a plan from its start but the continuation of the candidates a per-stage batch handed over (TakeOver above): cluster k runs candidate
// take->cand[k] from its pending command on, with the per-stage path's vectors and a dense state rebuilt from its history (frx_compact.hpp); everybody
// else's results are left alone.
static int optimize_resident(frx_problem *p, const frx_lbfgs_params &pm, double *x, int *status, int *iters, int *evals, double *objective, TakeOver *take = nullptr) {
    const int B = p->B, m = pm.mem_size;
    const int Bsel = take ? (int)take->cand.size() : B;                              // candidates that run here
    int E = frx::ROUND_E;
    p->resident_used = 0; p->resident_status = 0; p->resident_failed = 0;
    if (p->geo.solver != frx::SOLVER_KNOT_PCR || m < 1 || m > 128 || Bsel < 1) return 1;
    if (take && (p->geo.knot_threads != 64 || m != 128 || p->dv_mem != m || !p->d_S.p)) return 1;    // the take-over instantiation: <= 64 pieces, the stock history length
    // The compact representation inverts R = S^T Y (triangular part); with fewer variables than twice the history length the pairs
    // become linearly dependent and R^-1 loses its digits (n = 1: entries grow like 2^k), where the two-loop recursion of
    // k_lbfgs_pre stays stable.  Such problems (a few pieces) are latency-trivial anyway and take the per-stage rounds.
    for (int q = 0; q < Bsel; q++) {
        const int b = take ? take->cand[q] : q;
        if (p->xoff[b + 1] - p->xoff[b] < 2 * m) return 1;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    {   // Half the history elements per thread - twice the history workgroups - when the chip has room for them (up to 16 candidates at the headline
        // size; the reference plans ONE): no history register lives in an AGPR then and both history loops are half as long (round 3 measured it
        // on a branch, 31.1 -> 30.3 us per round with one candidate; merged in round 4).  The summation order over the history workgroups, and with
        // it the last bits of a plan, depend on the size class of the batch.  FRX_RESIDENT_E=56 keeps the large chunks.
        const char *ee = std::getenv("FRX_RESIDENT_E");
        const int Es = frx::ROUND_E_SMALL, Gs = 2 + std::max(1, (p->geo.maxXb + 2 * Es - 1) / (2 * Es));
        if (!(ee && std::atoi(ee) == frx::ROUND_E) && Gs <= 16 && (long)8 * Gs * ((Bsel + 7) / 8) <= cus) E = Es;
    }
    int G = 2 + std::max(1, (p->geo.maxXb + 2 * E - 1) / (2 * E));                   // leader + history workgroups (2 E elements of every pair each) + dense
    const int G_min = std::max(G, 3);
    {   // more workgroups per candidate when the chip has room: the penalty integrand of a candidate is spread over the history workgroups, four wave-tasks each, and
        // the leader - last in line - should get none: it has the adjoint to prepare.  Until round 5 the formula counted the leader as a worker and stopped at 16: the
        // reference's own operating point (ONE candidate, kappa = 48: 64 wave-tasks) ran its penalty share in two passes on 60 waves.  With 18 workgroups (16 x 4 waves,
        // one pass, leader free) a round takes 24.4 instead of 26.4 us, bit-identical plans (scripts/r05/plumbing_g_probe.py: 17 -> 25.6, 20 -> 24.5, 24 -> 25.7).
        const int tasks = (p->geo.maxN + p->geo.ppw - 1) / p->geo.ppw;
        const int want = std::min({(tasks + 3) / 4 + 2, 18, cus / std::max(Bsel, 1), cus / (8 * ((Bsel + 7) / 8))});   // (the last bound: the grid is 8 G ceil(B / 8) blocks - a wish that does not fit is not made)
        G = std::max(G, want);
    }
    if (const char *ge = std::getenv("FRX_RESIDENT_G")) G = std::max(G, std::atoi(ge));
    G = std::max(G, 3);
    // Every workgroup must be resident at once (one per CU; grid = 8 G ceil(S / 8)).  A batch the chip cannot hold at once runs on
    // S < B clusters that stay on the chip and take the remaining candidates one after the other (work queue: a cluster whose candidate is
    // finished gets the next one with its DV_NEXT command instead of DV_QUIT).  The clusters run independently, so the batch takes
    // ~sum(evaluations) / S rounds where the per-stage path - every candidate in every launch - takes max(evaluations) rounds of a
    // launch that grows with B: measured (profiles/r03_queue_sizes.jsonl) the queue wins up to a few times the chip's capacity, the
    // per-stage path beyond; resident_mode 2 / FRX_RESIDENT_QUEUE=1 force the queue, FRX_RESIDENT_QUEUE=0 forbids it.
    int S = Bsel;
    const char *ce = take ? nullptr : std::getenv("FRX_RESIDENT_CLUSTERS");          // experiments / tests: no more than this many clusters
    if ((long)8 * G * ((Bsel + 7) / 8) > cus) { G = G_min; S = std::min(Bsel, 8 * (cus / (8 * G))); }
    if (ce) S = std::min(S, std::max(1, std::atoi(ce)));
    if (S < 1) return 1;
    if (take && S < Bsel) return 1;                                                   // a take-over has one cluster per straggler (no queue)
    if (S < B && !take) {
        const char *qe = std::getenv("FRX_RESIDENT_QUEUE");
        const int queue_max = [] { const char *e = std::getenv("FRX_RESIDENT_QUEUE_MAX"); return e ? std::atoi(e) : 3; }();   // x capacity (measured: queue 484 ms against 498 ms per stage at 96 = 3 x 32 candidates, 626 against 514 at 128)
        const bool allowed = qe ? qe[0] != '0' : (ce || p->resident_mode == 2 || B <= queue_max * S);
        if (!allowed) return 1;
    }
    const size_t lds = frx::round_lds_bytes(p->geo, m, E);
    if (lds == 0 || lds > 160 * 1024) return 1;
    const int NXP = (G - 2) * 2 * E;
    const size_t n_words = (size_t)frx::ROUND_WORDS_PER_CAND * S + 2 + (size_t)S * G + 4 * (size_t)B;
    hipError_t e = hipSuccess;
    if (p->rk_B != B || p->rk_S != S || p->rk_G != G || p->rk_NXP != NXP) {
        p->rk_B = 0;
        auto need = [](auto &buf, size_t count) -> hipError_t { return buf.n >= count && buf.p ? hipSuccess : buf.alloc(count); };
        NumaScope numa_alloc(p->device);                                                  // the mailboxes' pages come from the device's NUMA node
        if ((e = p->d_pubsyg.alloc((size_t)S * (3 * NXP + 2))) != hipSuccess || (e = p->d_part.alloc((size_t)S * G * 512)) != hipSuccess ||
            (e = p->d_upub.alloc((size_t)S * 258)) != hipSuccess || (e = p->d_dpub.alloc((size_t)S * 2 * NXP)) != hipSuccess ||
            (e = p->d_out20ll.alloc((size_t)p->P * 40)) != hipSuccess || (e = p->d_rwords.alloc(n_words)) != hipSuccess || (e = p->h_rcmd.alloc((size_t)8 * S)) != hipSuccess || (e = p->h_rres.alloc((size_t)8 * S)) != hipSuccess ||
            (e = need(p->d_xp, p->NX)) != hipSuccess || (e = need(p->d_gp, p->NX)) != hipSuccess || (e = need(p->d_dir, p->NX)) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        p->rk_B = B; p->rk_S = S; p->rk_G = G; p->rk_NXP = NXP;
    }
    // direction log (frx_debug.h): [B] record counts, then dirlog_cands x dirlog_cap records {s, y, g, d, slot, pair count}
    const int log_cands = std::min(p->dirlog_cands, B);
    const bool want_dbg = p->dirlog_cap > 0 && log_cands > 0;
    const size_t log_rec = 4 * (size_t)NXP + 2, log_doubles = want_dbg ? (size_t)B + (size_t)log_cands * p->dirlog_cap * log_rec : 0;
    p->dirlog.clear(); p->dirlog_nxp = NXP;
    if (want_dbg) {
        if ((!p->d_rdbg.p || p->d_rdbg.n < log_doubles) && p->d_rdbg.alloc(log_doubles) != hipSuccess) return fail(FRX_ERR_ALLOC, "direction log does not fit the device");
        HIP_TRY(hipMemsetAsync(p->d_rdbg.p, 0, sizeof(double) * log_doubles, p->stream));
    }
    const bool want_prof = !take && std::getenv("FRX_RESIDENT_PROF") != nullptr;
    if (want_prof) {
        if ((!p->d_rprof.p || p->d_rprof.n < (size_t)S * (G + 1) * 16) && p->d_rprof.alloc((size_t)S * (G + 1) * 16) != hipSuccess) return 1;
        HIP_TRY(hipMemsetAsync(p->d_rprof.p, 0, sizeof(unsigned long long) * (size_t)S * (G + 1) * 16, p->stream));
    }
    p->rprof.clear();
    const double timeout_ms = [] { const char *ev = std::getenv("FRX_ROUND_TIMEOUT_MS"); const double v = ev ? std::atof(ev) : 0.0; return v > 0.0 ? v : 5000.0; }();
    // state of this launch: all polled words zero, mailboxes empty
    HIP_TRY(hipMemsetAsync(p->d_rwords.p, 0, sizeof(unsigned) * n_words, p->stream));
    HIP_TRY(hipMemsetAsync(p->d_out20ll.p, 0, sizeof(double) * (size_t)p->P * 40, p->stream));
    HIP_TRY(hipMemsetAsync(p->d_dpub.p, 0, sizeof(double) * (size_t)S * 2 * NXP, p->stream));           // granules of the cluster's hand-offs: tag 0 = nothing yet
    HIP_TRY(hipMemsetAsync(p->d_pubsyg.p, 0, sizeof(double) * (size_t)S * (3 * NXP + 2), p->stream));   // the point and gradient the cluster reads: zero beyond n (padding of s and y)
    std::memset(p->h_rcmd.p, 0, sizeof(unsigned long long) * 8 * S);
    std::memset(p->h_rres.p, 0, sizeof(unsigned long long) * 8 * S);
    if (!take) {
        std::memcpy(p->h_x.p, x, sizeof(double) * p->NX);
        HIP_TRY(hipMemcpyAsync(p->d_x.p, p->h_x.p, sizeof(double) * p->NX, hipMemcpyHostToDevice, p->stream));
    }
    frx::RoundLaunch rl;
    if (take) {
        // The dense state of every straggler from its history rows: read back (1.5 MB per candidate), R = S^T Y (triangular part), its inverse, Y^T Y and D on the
        // set-up pool's threads (frx_compact.hpp: 2 x 128 x 128 dot products of length n per candidate), uploaded as the clusters' take-over arrays.
        const size_t HS = p->dv_hs, rowblk = (size_t)m * HS;
        std::vector<double> hS((size_t)S * rowblk), hY((size_t)S * rowblk), rinv((size_t)S * 128 * 129), yy((size_t)S * 128 * 128), vd((size_t)S * 128);
        for (int k = 0; k < S; k++) {
            const int b = take->cand[k];
            HIP_TRY(hipMemcpyAsync(hS.data() + (size_t)k * rowblk, p->d_S.p + (size_t)b * rowblk, sizeof(double) * rowblk, hipMemcpyDeviceToHost, p->stream));
            HIP_TRY(hipMemcpyAsync(hY.data() + (size_t)k * rowblk, p->d_Y.p + (size_t)b * rowblk, sizeof(double) * rowblk, hipMemcpyDeviceToHost, p->stream));
        }
        HIP_TRY(hipStreamSynchronize(p->stream));
        frx::TaskPool::get().run(S, frx::setup_threads(S, 8000.0), [&](int k, int) {
            const int b = take->cand[k];
            frx::compact_from_history(m, p->xoff[b + 1] - p->xoff[b], HS, take->bound[k], take->newest[k], hS.data() + (size_t)k * rowblk, hY.data() + (size_t)k * rowblk, 129,
                                      rinv.data() + (size_t)k * 128 * 129, yy.data() + (size_t)k * 128 * 128, vd.data() + (size_t)k * 128);
        });
        std::vector<int> ints((size_t)3 * S);
        for (int k = 0; k < S; k++) { ints[k] = take->cand[k]; ints[S + k] = take->newest[k]; ints[2 * S + k] = take->bound[k]; }
        auto need = [](auto &buf, size_t count) -> hipError_t { return buf.n >= count && buf.p ? hipSuccess : buf.alloc(count); };
        if (need(p->d_rs_int, ints.size()) != hipSuccess || need(p->d_rs_f, (size_t)S) != hipSuccess || need(p->d_rs_rinv, rinv.size()) != hipSuccess ||
            need(p->d_rs_yy, yy.size()) != hipSuccess || need(p->d_rs_vd, vd.size()) != hipSuccess) { (void)hipGetLastError(); return 1; }
        HIP_TRY(hipMemcpy(p->d_rs_int.p, ints.data(), sizeof(int) * ints.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_rs_f.p, take->f_last.data(), sizeof(double) * S, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_rs_rinv.p, rinv.data(), sizeof(double) * rinv.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_rs_yy.p, yy.data(), sizeof(double) * yy.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_rs_vd.p, vd.data(), sizeof(double) * vd.size(), hipMemcpyHostToDevice));
        rl.rs_cand = p->d_rs_int.p; rl.rs_newest = p->d_rs_int.p + S; rl.rs_bound = p->d_rs_int.p + 2 * S;
        rl.rs_f = p->d_rs_f.p; rl.rs_S = p->d_S.p; rl.rs_Y = p->d_Y.p; rl.rs_hs = HS; rl.rs_rinv = p->d_rs_rinv.p; rl.rs_yy = p->d_rs_yy.p; rl.rs_vd = p->d_rs_vd.p;
    }
    rl.x = p->d_x.p; rl.g = p->d_g.p; rl.xp = p->d_xp.p; rl.gp = p->d_gp.p; rl.d = p->d_dir.p; rl.f = p->d_f.p; rl.T = p->d_T.p; rl.C = p->d_C.p; rl.out20 = p->d_out20.p;
    rl.pubsyg = p->d_pubsyg.p; rl.part = p->d_part.p; rl.upub = p->d_upub.p; rl.dpub = p->d_dpub.p; rl.out20ll = std::getenv("FRX_RESIDENT_NO_LL20") ? nullptr : (unsigned long long *)p->d_out20ll.p; rl.dbg = want_dbg ? p->d_rdbg.p : nullptr; rl.dbg_cap = want_dbg ? p->dirlog_cap : 0; rl.dbg_cands = want_dbg ? log_cands : 0;
    rl.words = p->d_rwords.p; rl.h_cmd = p->h_rcmd.p; rl.h_res = p->h_rres.p;
    if (!p->d_rargs.p && p->d_rargs.alloc(frx::round_args_bytes()) != hipSuccess) { (void)hipGetLastError(); p->d_rargs.p = nullptr; }
    rl.args_dev = p->d_rargs.p;
    rl.timeout_ticks = (unsigned long long)(timeout_ms * 1e5);                        // wall_clock64: 100 MHz
    rl.B = B; rl.S = S; rl.G = G; rl.m = m; rl.E = E; rl.NXP = NXP;
    rl.prof = want_prof ? p->d_rprof.p : nullptr;
    // FRX_RESIDENT_TRACE="lo,hi[,file]" (with FRX_RESIDENT_PROF): timeline of cluster 0 for the phases lo <= n < hi - every workgroup's thread 0 logs
    // (segment id, phase, 100 MHz wall clock) - written as text to `file` (default frx_round_trace.txt) after the plan (scripts/r04/round_timeline.py)
    DevBuf<unsigned long long> d_trace;
    const int trace_cap = 2048;
    const char *tr_env = want_prof ? std::getenv("FRX_RESIDENT_TRACE") : nullptr;
    std::string trace_file = "frx_round_trace.txt";
    if (tr_env) {
        unsigned lo = 0, hi = 0; char fbuf[512] = {0};
        const int nf = std::sscanf(tr_env, "%u,%u,%511s", &lo, &hi, fbuf);
        if (nf >= 2 && hi > lo && d_trace.alloc((size_t)G * trace_cap) == hipSuccess) {
            HIP_TRY(hipMemsetAsync(d_trace.p, 0, sizeof(unsigned long long) * (size_t)G * trace_cap, p->stream));
            rl.trace = d_trace.p; rl.trace_cap = trace_cap; rl.trace_lo = lo; rl.trace_hi = hi;
            if (nf == 3) trace_file = fbuf;
        }
    }
    { const char *fc = std::getenv("FRX_RESIDENT_FAST_CONTROL"); rl.fast_control = fc && fc[0] == '0' ? 0 : 1; }
    { const char *tr = std::getenv("FRX_RESIDENT_TIMED_READ"); if (tr && tr[0] == '1') rl.fast_control |= 2; }
    { const char *er = std::getenv("FRX_RESIDENT_EARLY_READ"); if (er && er[0] == '0') rl.fast_control |= 4; }
    { const char *wt = std::getenv("FRX_RESIDENT_WRITE_THROUGH"); if (wt && wt[0] == '1') rl.fast_control |= 8; }   // no cluster trusts its XCD census: write-through stores everywhere (what a cluster spread over XCDs does)
    { const char *sr = std::getenv("FRX_RESIDENT_STAMP_ROUND"); rl.stamp_round = want_prof && sr ? std::max(0, std::atoi(sr)) : 0; }
    rl.ls_ftol = pm.f_dec_coeff; rl.ls_gtol = pm.s_curv_coeff; rl.ls_min_step = pm.min_step; rl.ls_max_step = pm.max_step; rl.ls_xtol = pm.xtol; rl.ls_max_linesearch = pm.max_linesearch;
    {   // What the leader expects of the host (frx_round_kernel.hpp): 0 nothing, it waits for every command; 1 the acceptance of a trial (ADVANCE, next
        // slot, step 1: lbfgs.hpp:1418); 2 also the next trial step of a running search (default).  Every expectation is checked against the host's
        // command word before the next result is posted.  Measured, production kernel, us per round: 32 candidates 31.5 (round 2's rule: first-trial
        // acceptances only) -> 30.7 (level 2); one candidate 30.3 -> 29.9.  FRX_RESIDENT_SPECULATE=0|1|2 overrides.
        const char *sp = std::getenv("FRX_RESIDENT_SPECULATE");
        int level = 2;
        if (sp) level = std::max(0, std::min(2, std::atoi(sp)));
        rl.speculate = (pm.min_step <= 1.0 && 1.0 <= pm.max_step) ? level : 0;         // the predicted ADVANCE carries step 1 (lbfgs.hpp:1418)
    }
    struct StampGuard { frx_problem *q; ~StampGuard() { q->dp.stamps = nullptr; } } stamp_guard{p};
    const bool want_stamps = want_prof && std::getenv("FRX_RESIDENT_PROF")[0] != '2';   // FRX_RESIDENT_PROF=2: segment times only (the stamps' stores perturb what they measure)
    if (want_prof && !want_stamps && p->d_stamps.p) HIP_TRY(hipMemsetAsync(p->d_stamps.p, 0, 64 * sizeof(long long), p->stream));
    if (want_stamps) {                                                                // phase stamps of candidate 0's forward / adjoint bodies (last evaluation)
        if (!p->d_stamps.p) HIP_TRY(p->d_stamps.alloc(64));
        HIP_TRY(hipMemsetAsync(p->d_stamps.p, 0, 64 * sizeof(long long), p->stream));
        p->dp.stamps = p->d_stamps.p;
    }

    // Host state per CLUSTER, one cache-line-aligned slot each and a CONTIGUOUS range of slots per service thread: with the state in
    // parallel arrays and candidate b served by thread b % nsrv (round 2), every line of `seq`, `waiting`, `cmd` and the solvers was
    // written by all threads - two service threads answered SLOWER than one (leader's wait for a command at 32 candidates: median
    // 8-16 us with two threads, 4-8 us with one, 2-4 us at 8 candidates; profiles/r03_hostwait_probe.jsonl) and more threads bought nothing.
    // A slot holds the solver of the candidate its cluster is working on; with more candidates than clusters (S < B) a slot whose
    // candidate is finished takes the next one of the batch (`next_cand`) and tells the cluster with DV_NEXT.
    struct alignas(128) Slot { frx::SolverDV sv; frx::DvCommand cmd; unsigned long long seq = 0; long ncmd = 0; int cand = -1; char waiting = 0, quit_sent = 0, switching = 0; };
    std::vector<Slot> slot_(S);
    std::atomic<int> next_cand{S};
    volatile unsigned long long *hc = p->h_rcmd.p, *hr = p->h_rres.p;
    { const char *cs = std::getenv("FRX_RESIDENT_CMD_STRIDE"); rl.cmd_stride = cs && std::atoi(cs) == 1 ? 1 : 4; }   // 4: one cache line per cluster
    const size_t cs2 = 2 * (size_t)rl.cmd_stride;
    auto post = [&](int k, int flags, int slot, int bound, double step) {
        std::memcpy((void *)(hc + cs2 * k + 1), &step, sizeof(double));
        std::atomic_thread_fence(std::memory_order_release);
        ++slot_[k].seq;
        if (step == 1.0 && !(flags & (128 | 32))) flags |= 64;                         // DV_STEP_IS_ONE: lets the leader confirm a predicted command from this word alone
        if ((flags & frx::DV_TRIAL) && !(flags & frx::DV_ADVANCE)) {                   // a trial inside a search: slot and pair count mean nothing here, a fold of the step's bits rides in their place
            const unsigned h = frx::dv_step_hash(step);
            slot = (int)(h & 0xFFFu); bound = (int)(h >> 12);
        }
        hc[cs2 * k] = (slot_[k].seq << 32) | ((unsigned long long)(bound & 0xFFF) << 20) | ((unsigned long long)(slot & 0xFFF) << 8) | (unsigned long long)(flags & 0xFF);
        slot_[k].ncmd++;
    };
    const bool tracing = std::getenv("FRX_TRACE") != nullptr;
    if (!take) p->trace.clear();                                                      // (a take-over continues the per-stage part's trace)
    auto record = [&](int k) {                                                       // the finished plan of slot k's candidate (the point itself is read from the device at the end)
        const int b = slot_[k].cand;
        status[b] = slot_[k].sv.status();
        if (iters) iters[b] = slot_[k].sv.iterations();
        if (evals) evals[b] = slot_[k].sv.evaluations();
        if (objective) objective[b] = slot_[k].sv.value();
    };
    // slot k's candidate is finished (or had nothing to run): the next candidate of the batch for its cluster, or DV_QUIT.  Returns true
    // while the slot has work.
    auto hand_over = [&](int k) -> bool {
        record(k);
        const int nb = next_cand.load(std::memory_order_relaxed) < B ? next_cand.fetch_add(1) : B;
        if (nb >= B) { post(k, 128, 0, 0, 0.0); slot_[k].quit_sent = 1; slot_[k].waiting = 0; return false; }   // this cluster leaves the chip
        slot_[k].cand = nb;
        slot_[k].sv.start(p->xoff[nb + 1] - p->xoff[nb], pm, &slot_[k].cmd);
        post(k, 32, nb & 0xFFF, nb >> 12, 0.0);                                      // DV_NEXT: the cluster re-binds and answers with the command's sequence number
        slot_[k].switching = 1; slot_[k].waiting = 1;
        return true;
    };
    if (!take) {
        for (int b = 0; b < B; b++) status[b] = 0;
        for (int k = 0; k < S; k++) { slot_[k].cand = k; slot_[k].sv.start(p->xoff[k + 1] - p->xoff[k], pm, &slot_[k].cmd); }
    } else {
        next_cand.store(B);                                                           // nobody is handed a further candidate
        for (int k = 0; k < S; k++) { slot_[k].cand = take->cand[k]; slot_[k].sv = take->sv[k]; slot_[k].sv.rebind(&slot_[k].cmd); }   // the pending command comes along
    }
    // the threads that talk to the device stay on the device's NUMA node for the length of the plan (FRX_NUMA=0: wherever the scheduler puts them);
    // the caller's own affinity is put back when the plan is over.  Looked up (cached per device) before the launch: the clusters spin from then on
    cpu_set_t numa_set, caller_set;
    const bool numa_pin = !([] { const char *e = std::getenv("FRX_NUMA"); return e && e[0] == '0'; }()) && device_numa_cpus(p->device, &numa_set);
    std::unique_lock<std::mutex> device_slot(resident_device_lock(p->device));        // one resident grid per device at a time (see above)
    const auto t0 = clk::now();
    HIP_TRY((hipError_t)frx::launch_round(p->dp, p->geo, rl, p->stream));
    for (int k = 0; k < S; k++) {
        if (slot_[k].cmd.flags != 0) { post(k, slot_[k].cmd.flags, slot_[k].cmd.slot, slot_[k].cmd.bound, slot_[k].cmd.step); slot_[k].waiting = 1; }
        else hand_over(k);                                                           // invalid parameters: nothing to run
    }
    // Mailbox service: thread tid serves the clusters [S tid / nsrv, S (tid + 1) / nsrv) (the caller is thread 0).  The clusters of a batch
    // run in near lock-step, so their results arrive together and a thread's k-th mailbox waits for the k - 1 before it.
    // Sixteen clusters per thread, at most four threads (round 4; four per thread, at most eight, until then): 1, 2, 4 and 8 threads measure the same round
    // (25.2-25.4 us at 32 clusters, profiles/r04_host_threads.txt) - a scan of 16 mailboxes and the line-search step of those that answered take a few
    // microseconds of a 25 us round - and every one of them SPINS for the length of the plan: under a CPU quota (the GPU boxes of this project run the
    // process with 16 CPUs' worth) eight ranks of a node with nine spinning threads each are throttled to a crawl, eight ranks with two are not.
    // ... and no more of them than this plan's share of the CPUs the process may use: cgroup quota and affinity mask, divided by
    // the plans that spin next to this one - the other ranks of the node (LOCAL_WORLD_SIZE, set by torch.distributed.run; FRX_LOCAL_RANKS overrides)
    // and the other shards of a frx_multi job in this process - minus one CPU for the rank's main (interpreter) thread.  Eight ranks under the GPU
    // boxes' quota of 16 CPUs: the caller alone serves the mailboxes (one thread measures the same round as eight, profiles/r04_host_threads.txt).
    int nsrv = frx::mailbox_threads(S);
    if (const char *se = std::getenv("FRX_RESIDENT_HOST_THREADS")) nsrv = std::max(1, std::min(std::atoi(se), S));
    std::atomic<int> abort_code{0};                                                  // 1 = device gave up, 2 = host deadline
    const int scan_pause = [] { const char *e = std::getenv("FRX_RESIDENT_SCAN_PAUSE"); return e ? std::max(0, std::atoi(e)) : 0; }();   // extra pauses between two scans of a thread's mailboxes (experiments)
    std::vector<double> t_host_thr(nsrv, 0.0);
    std::vector<long> scans(nsrv, 0);
    const bool caller_saved = numa_pin && pthread_getaffinity_np(pthread_self(), sizeof(caller_set), &caller_set) == 0;
    if (caller_saved) (void)pthread_setaffinity_np(pthread_self(), sizeof(numa_set), &numa_set);
    auto serve = [&](int tid) {
        if (numa_pin && tid > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof(numa_set), &numa_set);
        int mine = 0;
        long nscan = 0;
        const int lo = (int)((long)S * tid / nsrv), hi = (int)((long)S * (tid + 1) / nsrv);      // this thread's clusters
        for (int k = lo; k < hi; k++) mine += slot_[k].waiting ? 1 : 0;
        auto t_last = clk::now();
        while (mine > 0 && abort_code.load(std::memory_order_relaxed) == 0) {
            bool progress = false;
            nscan++;
            for (int k = lo; k < hi; k++) {
                if (!slot_[k].waiting) continue;
                const unsigned long long rs = hr[8 * k + 7];
                if (rs == ~0ull) { abort_code.store(1); break; }                        // the device gave up on this cluster
                if (rs != slot_[k].seq) continue;
                std::atomic_thread_fence(std::memory_order_acquire);
                progress = true;
                const auto th = clk::now();
                frx::DvCommand &c = slot_[k].cmd;
                if (slot_[k].switching) slot_[k].switching = 0;                         // the cluster is on its new candidate: c holds the first command of that plan (SolverDV::start)
                else if (c.flags & frx::DV_EVAL) {
                    frx::DvResult r;
                    std::memcpy(&r, (const void *)(hr + 8 * k), 5 * sizeof(double));
                    if (tracing && slot_[k].cand == 0) { const double row[7] = {(double)c.flags, c.step, r.f, r.dg, r.dginit, r.xx, r.gg}; p->trace.insert(p->trace.end(), row, row + 7); }
                    if (slot_[k].sv.saw_nonfinite(r.f)) slot_[k].sv.give_up(frx::LBERR_ROUNDING); else slot_[k].sv.feed(r);
                } else c.flags = 0;                                                     // a RESTORE has been executed
                if (c.flags != 0) post(k, c.flags, c.slot, c.bound, c.step);
                else if (!hand_over(k)) mine--;
                t_host_thr[tid] += ms_since(th);
            }
            if (progress) t_last = clk::now();
            else if (ms_since(t_last) > timeout_ms) abort_code.store(2);
            else {
                // a kernel that has LEFT while clusters still wait (start-up census failed: the grid did not fit next to another
                // process's work; a device fault) will never answer: notice it from the stream instead of sitting out the whole timeout
                if (tid == 0 && (nscan & 0x3FFF) == 0 && hipStreamQuery(p->stream) != hipErrorNotReady) {
                    bool answered = true;
                    for (int k = 0; k < S; k++) if (slot_[k].waiting && hr[8 * k + 7] != slot_[k].seq) answered = false;
                    if (!answered) abort_code.store(1);
                }
                for (int q = 0; q <= scan_pause; q++) __builtin_ia32_pause();
            }
        }
        scans[tid] = nscan;
    };
    {
        std::vector<std::thread> helpers;
        for (int tid = 1; tid < nsrv; tid++) helpers.emplace_back(serve, tid);
        serve(0);
        for (auto &h : helpers) h.join();
    }
    if (caller_saved) (void)pthread_setaffinity_np(pthread_self(), sizeof(caller_set), &caller_set);
    if (std::getenv("FRX_RESIDENT_HOST_STATS")) {                                     // diagnostic: how often a service thread looks at each of its mailboxes
        const double wall_us = 1e3 * ms_since(t0);
        for (int tid = 0; tid < nsrv; tid++) std::fprintf(stderr, "[frx] mailbox thread %d: %ld scans of %d mailboxes in %.0f us = %.3f us per scan, busy %.1f ms\n", tid, scans[tid], (int)((long)S * (tid + 1) / nsrv) - (int)((long)S * tid / nsrv), wall_us, wall_us / std::max(1L, scans[tid]), t_host_thr[tid]);
    }
    int rc = FRX_OK;
    double t_host = 0.0;
    for (double v : t_host_thr) t_host = std::max(t_host, v);
    if (abort_code.load() == 1) rc = 1;
    else if (abort_code.load() == 2) rc = fail(FRX_ERR_TIMEOUT, "resident round kernel: no result within FRX_ROUND_TIMEOUT_MS");
    for (int k = 0; k < S; k++) if (!slot_[k].quit_sent) post(k, 128, 0, 0, 0.0);
    {   // bounded drain: the kernel's own spins expire after the same timeout
        const auto tw = clk::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(p->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return fail(FRX_ERR_HIP, std::string("resident round kernel failed: ") + hipGetErrorString(q));
            if (ms_since(tw) > 3.0 * timeout_ms + 1000.0) return fail(FRX_ERR_TIMEOUT, "resident round kernel did not exit");
            __builtin_ia32_pause();
        }
    }
    device_slot.unlock();                                                             // the kernel has left the chip
    unsigned st[2] = {0, 0};
    HIP_TRY(hipMemcpy(st, p->d_rwords.p + (size_t)frx::ROUND_WORDS_PER_CAND * S, sizeof(st), hipMemcpyDeviceToHost));
    p->resident_status = st[1];
    {
        std::vector<unsigned> sc(4 * (size_t)B, 0u);
        HIP_TRY(hipMemcpy(sc.data(), p->d_rwords.p + (size_t)frx::ROUND_WORDS_PER_CAND * S + 2 + (size_t)S * G, sizeof(unsigned) * sc.size(), hipMemcpyDeviceToHost));
        for (int q = 0; q < 4; q++) { p->spec_counts[q] = 0; for (int b = 0; b < B; b++) p->spec_counts[q] += sc[4 * (size_t)b + q]; }
    }
    if (std::getenv("FRX_RESIDENT_HOST_STATS")) {                                     // diagnostic: where the workgroups of every cluster ran (XCC id of each)
        std::vector<unsigned> xc((size_t)S * G, 0u);
        HIP_TRY(hipMemcpy(xc.data(), p->d_rwords.p + (size_t)frx::ROUND_WORDS_PER_CAND * S + 2, sizeof(unsigned) * xc.size(), hipMemcpyDeviceToHost));
        int one_xcd = 0;
        std::string line;
        for (int k = 0; k < S; k++) {
            bool same = true;
            for (int w = 1; w < G; w++) same = same && (xc[(size_t)k * G + w] & 0xFFu) == (xc[(size_t)k * G] & 0xFFu);
            one_xcd += same ? 1 : 0;
            line += " [";
            for (int w = 0; w < G; w++) {                                // XCD : shader engine . shader array . CU of every workgroup (leader first, dense last)
                const unsigned e = xc[(size_t)k * G + w];
                line += std::to_string((int)(e & 0xFFu) - 1) + ":" + std::to_string((e >> 13) & 7u) + "." + std::to_string((e >> 12) & 1u) + "." + std::to_string((e >> 8) & 15u) + (w + 1 < G ? " " : "");
            }
            line += "]";
        }
        std::fprintf(stderr, "[frx] clusters on one XCD: %d of %d;%s\n", one_xcd, S, line.c_str());
    }
    if (want_dbg) {
        p->dirlog.resize(log_doubles);
        HIP_TRY(hipMemcpy(p->dirlog.data(), p->d_rdbg.p, sizeof(double) * log_doubles, hipMemcpyDeviceToHost));
    }
    if (rc < 0) return rc;
    if (rc != FRX_OK || st[1] != 0) return 1;
    long rounds = 0;
    for (int k = 0; k < S; k++) rounds = std::max(rounds, slot_[k].ncmd);                 // commands of the busiest cluster (S < B: over all the candidates it took)
    p->stats[0] = ms_since(t0); p->stats[1] = p->stats[0] - t_host; p->stats[2] = t_host; p->stats[3] = (double)rounds;
    if (!take) HIP_TRY(hipMemcpy(x, p->d_x.p, sizeof(double) * p->NX, hipMemcpyDeviceToHost));
    else                                                                              //  a take-over's result: the stragglers' points only - everybody else's x is final already
        for (int b : take->cand) HIP_TRY(hipMemcpy(x + p->xoff[b], p->d_x.p + p->xoff[b], sizeof(double) * (size_t)(p->xoff[b + 1] - p->xoff[b]), hipMemcpyDeviceToHost));
    if (rl.trace) {
        std::vector<unsigned long long> tr((size_t)G * trace_cap);
        HIP_TRY(hipMemcpy(tr.data(), d_trace.p, sizeof(unsigned long long) * tr.size(), hipMemcpyDeviceToHost));
        if (FILE *f = std::fopen(trace_file.c_str(), "w")) {
            std::fprintf(f, "# workgroup segment phase ticks(100MHz, 40 bits)   G=%d\n", G);
            for (int w = 0; w < G; w++)
                for (int i = 0; i < trace_cap; i++) {
                    const unsigned long long e = tr[(size_t)w * trace_cap + i];
                    if (!e) break;
                    std::fprintf(f, "%d %llu %llu %llu\n", w, e >> 56, (e >> 40) & 0xFFFFull, e & 0xFFFFFFFFFFull);
                }
            std::fclose(f);
        }
    }
    if (want_prof) {
        p->rprof.resize((size_t)S * (G + 1) * 16 + 32);                                // [S][G][16] segment sums, [S][16] host-wait histogram; the last 32 words: the bodies' cycle stamps
        HIP_TRY(hipMemcpy(p->rprof.data(), p->d_rprof.p, sizeof(unsigned long long) * (size_t)S * (G + 1) * 16, hipMemcpyDeviceToHost));
        if (!p->d_stamps.p) { HIP_TRY(p->d_stamps.alloc(64)); HIP_TRY(hipMemset(p->d_stamps.p, 0, 64 * sizeof(long long))); }
        HIP_TRY(hipMemcpy(p->rprof.data() + (size_t)S * (G + 1) * 16, p->d_stamps.p + (rl.stamp_round > 0 ? 32 : 0), 32 * sizeof(long long), hipMemcpyDeviceToHost));   // the last evaluation's stamps, or those of evaluation FRX_RESIDENT_STAMP_ROUND
    }
    for (int q = 0; q < Bsel; q++) {                                                  // (a take-over counts its own candidates, not the per-stage part's failures)
        const int b = take ? take->cand[q] : q;
        if (status[b] < 0 && status[b] != frx::LBERR_MAXIMUMITERATION) p->resident_failed++;
    }
    p->resident_used = G; p->resident_clusters = S;
    return FRX_OK;
}

// final generate (CPU.hpp:1258-1263) and the reference's return value (CPU.hpp:1267)
static int finish_optimize(frx_problem *p, const double *x, double *C, double *T, double *jerk_cost) {
    int rc;
    std::vector<double> Tt(p->P), Cc((size_t)p->P * 18);
    rc = frx_forward(p, x, Tt.data(), Cc.data());
    if (rc != FRX_OK) return rc;
    if (T) std::copy(Tt.begin(), Tt.end(), T);
    if (C) std::copy(Cc.begin(), Cc.end(), C);
    if (jerk_cost) {
        for (int b = 0; b < p->B; b++) {
            double obj = 0.0;                                            // getTrajJerkCost, CPU.hpp:507-520
            for (int gp = p->poff[b]; gp < p->poff[b + 1]; gp++) {
                const double *c3 = &Cc[18 * (size_t)gp + 9], *c4 = c3 + 3, *c5 = c3 + 6;
                const double t1 = Tt[gp], t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
                obj += 36.0 * frx::dot3(c3, c3) * t1 + 144.0 * frx::dot3(c4, c3) * t2 + 192.0 * frx::dot3(c4, c4) * t3 +
                       240.0 * frx::dot3(c5, c3) * t3 + 720.0 * frx::dot3(c5, c4) * t4 + 720.0 * frx::dot3(c5, c5) * t5;
            }
            jerk_cost[b] = obj;
        }
    }
    return FRX_OK;
}


extern "C" {

int frx_optimize(frx_problem *p, const frx_lbfgs_params *params, double *x, double *C, double *T, double *jerk_cost,
                 double *objective, int *status, int *iters, int *evals) {
    if (!p || !params || !x || !status) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (p->penalty_only) return fail(FRX_ERR_INVALID_ARG, "this handle serves frx_penalty_eval[_device] only (frx_penalty_problem_create)");
    HIP_TRY(hipSetDevice(p->device));
    // FRX_LBFGS=host keeps every vector on the host (bit-identical to the reference solver given identical f, g);
    // the default keeps the vectors on the device and only the line-search decisions on the host.
    const char *lb_env = std::getenv("FRX_LBFGS");
    const bool want_device_vectors = !(lb_env && lb_env[0] == 'h') && p->lbfgs_mode != 1;
    int rc_dv = 1;
    p->taken_over = 0;
    if (want_device_vectors) {
        // FRX_RESIDENT=0 keeps the one-launch-per-stage rounds; the default is the resident round kernel whenever the batch fits the
        // chip (one workgroup per CU, B x G of them), with the per-stage path as fallback when it does not or when the device gives up
        const char *rs_env = std::getenv("FRX_RESIDENT");
        p->resident_used = 0; p->resident_retried = 0;
        p->resident_failed = 0; p->resident_clusters = 0; for (auto &q : p->spec_counts) q = 0;      //  diagnostics of THIS plan, whichever path it takes
        const std::vector<double> x_start(x, x + p->NX);
        // Candidates whose resident plan ended with LBFGSERR_INCREASEGRADIENT are planned again on the per-stage rounds (see below, where the resident plan returns).
        // `scope` (optional): only these candidates are looked at - the stragglers of a take-over, whose dense state was rebuilt on the host by plain back-substitution
        // over up to 128 pairs (the long-running, worst-conditioned candidates of a batch; without this their g.d >= 0 would be final on that path alone).
        // The resident result of a candidate that is planned again is kept: when the per-stage path cannot run (rc2 > 0: its buffers do not fit) or fails
        // on it as well, it keeps the resident verdict, point and objective instead of ending at its start point.
        const char *rt_env = std::getenv("FRX_RESIDENT_RETRY");
        const int retry_mode = rt_env ? (rt_env[0] == '0' ? 0 : rt_env[0] == 't' ? 2 : 1) : (p->resident_retry != 0 ? 1 : 2);   // 2 = targeted (default)
        auto wants_retry = [&](int st) { return st < 0 && st != frx::LBERR_MAXIMUMITERATION && (retry_mode == 1 || (retry_mode == 2 && st == frx::LBERR_INCREASEGRADIENT)); };
        auto retry_increase_gradient = [&](const std::vector<int> *scope) -> int {
            std::vector<char> again(p->B, 0);
            int n_again = 0;
            if (scope) { for (int b : *scope) if (wants_retry(status[b])) { again[b] = 1; n_again++; } }
            else for (int b = 0; b < p->B; b++) if (wants_retry(status[b])) { again[b] = 1; n_again++; }
            if (n_again == 0) return 0;
            const std::vector<double> x_res(x, x + p->NX);
            std::vector<int> st_res(status, status + p->B), it_res, ev_res;
            std::vector<double> obj_res;
            if (iters) it_res.assign(iters, iters + p->B);
            if (evals) ev_res.assign(evals, evals + p->B);
            if (objective) obj_res.assign(objective, objective + p->B);
            for (int b = 0; b < p->B; b++) if (again[b]) std::copy(x_start.begin() + p->xoff[b], x_start.begin() + p->xoff[b + 1], x + p->xoff[b]);
            const int used = p->resident_used, taken = p->taken_over;
            const double t_res[4] = {p->stats[0], p->stats[1], p->stats[2], p->stats[3]};
            const int rc2 = optimize_device_vectors(p, *params, x, status, iters, evals, objective, &again);
            if (rc2 < 0) return rc2;
            if (rc2 != 0) {
                std::copy(x_res.begin(), x_res.end(), x);
                std::copy(st_res.begin(), st_res.end(), status);
                if (iters) std::copy(it_res.begin(), it_res.end(), iters);
                if (evals) std::copy(ev_res.begin(), ev_res.end(), evals);
                if (objective) std::copy(obj_res.begin(), obj_res.end(), objective);
                n_again = 0;
            }
            p->resident_used = used; p->taken_over = taken; p->resident_retried += n_again;
            if (rc2 == 0) p->stats[0] += t_res[0]; else p->stats[0] = t_res[0];
            if (scope) { for (int q = 1; q < 4; q++) p->stats[q] = (rc2 == 0 ? p->stats[q] : 0.0) + t_res[q]; }
            return 0;
        };
        bool resident_gave_up = false;
        if (!(rs_env && rs_env[0] == '0') && p->resident_mode != 0 && p->takeover_at <= 0) {      // (frx_debug_set_takeover_at, tests: per-stage rounds first, whatever the batch size)
            rc_dv = optimize_resident(p, *params, x, status, iters, evals, objective);
            resident_gave_up = rc_dv != 0 && p->resident_status != 0;
            if (rc_dv < 0) return rc_dv;
            if (rc_dv != 0) std::copy(x_start.begin(), x_start.end(), x);
            else {
                // A candidate that ends with an L-BFGS error on the resident kernel keeps that verdict, as it would in the reference
                // (lbfgs_optimize returns the code, SE3GCOPTER::optimize ignores it, CPU.hpp:1249): on the Monte-Carlo and perturbation
                // shares the resident kernel fails on no more candidates than the CPU oracle does (tests/test_gpu_configs.py) - with ONE
                // exception.  The resident kernel forms d = -H g in the compact representation (R^-1 maintained incrementally), the
                // reference by the two-loop recursion; in exact arithmetic H is positive definite in both and d is a descent direction.  A
                // search that starts with g.d >= 0 (LBFGSERR_INCREASEGRADIENT, lbfgs.hpp:766) can therefore only be the work of rounding in
                // the direction itself - an ill-conditioned R - and says nothing about the candidate: such candidates, and only those, are
                // planned again on the per-stage rounds, whose direction kernel IS the two-loop recursion.  Line searches that give up
                // (-1005, -1007: infeasible corridors, the CPU oracle's verdict as well) are final.  frx_debug_set_resident_retry(p, 1) /
                // FRX_RESIDENT_RETRY=1 re-run every failed candidate (diagnostic), FRX_RESIDENT_RETRY=0 none.
                const int rc_retry = retry_increase_gradient(nullptr);
                if (rc_retry < 0) return rc_retry;
            }
        }
        if (rc_dv != 0) {
            // A batch too large for the resident kernel's clusters runs as per-stage rounds - until no more candidates are left than the chip has clusters for:
            // those continue on the resident kernel (TakeOver; FRX_TAKEOVER=0 keeps the per-stage rounds to the end).  Not when the caller switched the
            // resident kernel or its queue off, and not when the resident kernel has just given up on this device.
            TakeOver take;
            {
                const char *te = std::getenv("FRX_TAKEOVER"), *qe = std::getenv("FRX_RESIDENT_QUEUE");
                const bool wanted = !(te && te[0] == '0') && !(rs_env && rs_env[0] == '0') && !(qe && qe[0] == '0') && p->resident_mode != 0 && !resident_gave_up;
                hipDeviceProp_t prop;
                if (wanted && p->geo.solver == frx::SOLVER_KNOT_PCR && p->geo.knot_threads == 64 && params->mem_size == 128 && hipGetDeviceProperties(&prop, p->device) == hipSuccess) {
                    bool fits = true;
                    for (int b = 0; b < p->B; b++) fits = fits && p->xoff[b + 1] - p->xoff[b] >= 2 * params->mem_size;
                    const int G = std::max(3, 2 + std::max(1, (p->geo.maxXb + 2 * frx::ROUND_E - 1) / (2 * frx::ROUND_E)));
                    const int cap = 8 * (prop.multiProcessorCount / (8 * G));
                    if (fits && cap >= 1 && p->B > cap) take.capacity = cap;
                    if (p->takeover_at > 0) { take.not_before = p->takeover_at; if (fits && cap >= 1 && p->B <= cap) take.capacity = cap; }   // (tests: a small batch hands over after so many rounds)
                }
            }
            rc_dv = optimize_device_vectors(p, *params, x, status, iters, evals, objective, nullptr, take.capacity > 0 ? &take : nullptr);
            if (rc_dv < 0) return rc_dv;
            if (rc_dv == 2) {
                const double ms_ps = take.ms, dev_ps = take.ms_dev, host_ps = take.ms_host;
                const long rounds_ps = take.rounds;
                const int rc3 = optimize_resident(p, *params, x, status, iters, evals, objective, &take);
                if (rc3 < 0) return rc3;
                if (rc3 == 0) {
                    p->taken_over = (int)take.cand.size();
                    p->stats[0] += ms_ps; p->stats[1] += dev_ps; p->stats[2] += host_ps; p->stats[3] += (double)rounds_ps;
                    const int rc_retry = retry_increase_gradient(&take.cand);             //  the stragglers get the resident path's targeted second chance
                    if (rc_retry < 0) return rc_retry;
                } else {
                    // the resident kernel could not take them (another process on the device, a geometry it does not cover): their vectors on the device are no
                    // longer the per-stage path's, so they are planned again from their start points, per stage - slower, never wrong
                    std::vector<char> again(p->B, 0);
                    for (int c : take.cand) { again[c] = 1; std::copy(x_start.begin() + p->xoff[c], x_start.begin() + p->xoff[c + 1], x + p->xoff[c]); }
                    const int rc4 = optimize_device_vectors(p, *params, x, status, iters, evals, objective, &again);
                    if (rc4 != 0) return rc4 < 0 ? rc4 : fail(FRX_ERR_HIP, "per-stage path unavailable after a failed take-over");
                    p->resident_used = 0;
                    p->stats[0] += ms_ps; p->stats[1] += dev_ps; p->stats[2] += host_ps; p->stats[3] += (double)rounds_ps;
                }
                rc_dv = 0;
            }
        }
    }
    if (rc_dv == 0) return finish_optimize(p, x, C, T, jerk_cost);
    std::memcpy(p->h_x.p, x, sizeof(double) * p->NX);
    int nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), (unsigned)p->B));
    if (const char *ht = std::getenv("FRX_HOST_THREADS")) nt = std::max(1, std::min(std::atoi(ht), p->B));
    int hip_rc = FRX_OK;
    // Zero-copy round trip: the kernels read x from and write (f, grad) to mapped, coherent host memory over PCIe
    // (205 KB each way at the headline size), so a round is three launches and one completion poll — no
    // hipMemcpyAsync calls (each costs more host time than the kernels it feeds).  FRX_ZERO_COPY=0 restores copies.
    const char *zc_env = std::getenv("FRX_ZERO_COPY");
    const bool zero_copy = !(zc_env && zc_env[0] == '0');
    auto wait_stream = [&]() -> hipError_t {
        for (;;) {
            hipError_t q = hipStreamQuery(p->stream);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
            __builtin_ia32_pause();
        }
    };
    auto eval_all = [&](int, const int *) -> int {
        hipError_t e;
        //  an expired wait of the one-launch form must not end the plan as a silent LBFGSERR_ROUNDING (NaN objectives): the evaluation is repeated
        // once with the stage kernels, which the handle keeps from then on (eval_cluster_status retires the form)
        for (int attempt = 0; attempt < 2; attempt++) {
            const bool fused = p->eval_fused != 0;
            if (zero_copy) {
                if ((e = (hipError_t)launch_eval(p, p->h_x.p, p->h_f.p, p->h_g.p, p->stream, true)) != hipSuccess) goto bad;
                if ((e = wait_stream()) != hipSuccess) goto bad;
            } else {
                if ((e = hipMemcpyAsync(p->d_x.p, p->h_x.p, sizeof(double) * p->NX, hipMemcpyHostToDevice, p->stream)) != hipSuccess) goto bad;
                if ((e = (hipError_t)launch_eval(p, p->d_x.p, p->d_f.p, p->d_g.p, p->stream, true)) != hipSuccess) goto bad;
                if ((e = hipMemcpyAsync(p->h_f.p, p->d_f.p, sizeof(double) * p->B, hipMemcpyDeviceToHost, p->stream)) != hipSuccess) goto bad;
                if ((e = hipMemcpyAsync(p->h_g.p, p->d_g.p, sizeof(double) * p->NX, hipMemcpyDeviceToHost, p->stream)) != hipSuccess) goto bad;
                if ((e = hipStreamSynchronize(p->stream)) != hipSuccess) goto bad;
            }
            if (!fused || eval_cluster_status(p, p->h_f.p) == FRX_OK) return FRX_OK;
        }
        return FRX_OK;
    bad:
        hip_rc = fail(FRX_ERR_HIP, std::string("device evaluation failed: ") + hipGetErrorString(e));
        return hip_rc;
    };
    int rc = drive_batch(p->B, p->xoff.data(), p->h_x.p, p->h_g.p, p->h_f.p, *params, nt, status, iters, evals, objective, p->stats,
                         eval_all);
    if (rc != FRX_OK) return rc;
    std::memcpy(x, p->h_x.p, sizeof(double) * p->NX);
    return finish_optimize(p, x, C, T, jerk_cost);
}

int frx_problem_set_resident(frx_problem *p, int enable) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    p->resident_mode = enable == 0 ? 0 : enable == 2 ? 2 : 1;
    return FRX_OK;
}
int frx_optimize_path(const frx_problem *p, int *resident_used, unsigned *device_status) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (resident_used) *resident_used = p->resident_used;
    if (device_status) *device_status = p->resident_status;
    return FRX_OK;
}
int frx_resident_profile(const frx_problem *p, unsigned long long *out, int cap_words) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    const int n = (int)p->rprof.size();
    if (out) std::memcpy(out, p->rprof.data(), sizeof(unsigned long long) * (size_t)std::min(n, std::max(cap_words, 0)));
    return n;
}
int frx_debug_direction_log(frx_problem *p, int cap_steps, int n_cands) {
    if (!p || cap_steps < 0 || n_cands < 0) return fail(FRX_ERR_INVALID_ARG, "null handle or negative size");
    p->dirlog_cap = cap_steps; p->dirlog_cands = cap_steps > 0 ? n_cands : 0;
    p->dirlog.clear();
    return FRX_OK;
}
int frx_debug_direction_log_read(const frx_problem *p, int cand, double *out, int cap_rows, int *rows, int *row_doubles) {
    if (!p || cand < 0) return fail(FRX_ERR_INVALID_ARG, "null handle or negative candidate");
    const size_t rec = 4 * (size_t)p->dirlog_nxp + 2;
    int n = 0;
    if (!p->dirlog.empty() && cand < std::min(p->dirlog_cands, p->B)) n = std::min((int)p->dirlog[cand], p->dirlog_cap);
    if (rows) *rows = n;
    if (row_doubles) *row_doubles = (int)rec;
    if (out && n > 0) std::memcpy(out, p->dirlog.data() + p->B + (size_t)cand * p->dirlog_cap * rec, sizeof(double) * rec * (size_t)std::min(n, std::max(cap_rows, 0)));
    return FRX_OK;
}
int frx_debug_set_resident_retry(frx_problem *p, int enable) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    p->resident_retry = enable ? 1 : 0;
    return FRX_OK;
}
int frx_debug_resident_predictions(const frx_problem *p, unsigned long long *out3) {
    if (!p || !out3) return fail(FRX_ERR_INVALID_ARG, "null argument");
    for (int q = 0; q < 3; q++) out3[q] = p->spec_counts[q];
    return FRX_OK;
}
int frx_debug_resident_clusters(const frx_problem *p, int *clusters) {
    if (!p || !clusters) return fail(FRX_ERR_INVALID_ARG, "null argument");
    *clusters = p->resident_clusters;
    return FRX_OK;
}
int frx_debug_resident_counts(const frx_problem *p, int *failed, int *retried) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    if (failed) *failed = p->resident_failed;
    if (retried) *retried = p->resident_retried;
    return FRX_OK;
}
int frx_debug_trace(const frx_problem *p, double *out, int cap_rows) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    const int rows = (int)(p->trace.size() / 7);
    if (out) std::memcpy(out, p->trace.data(), sizeof(double) * 7 * (size_t)std::min(rows, std::max(cap_rows, 0)));
    return rows;
}

int frx_debug_set_takeover_at(frx_problem *p, long rounds) {
    if (!p) return fail(FRX_ERR_INVALID_ARG, "null argument");
    p->takeover_at = rounds > 0 ? rounds : 0;
    return FRX_OK;
}
// Diagnostic (bench, scripts/r06/mode_probe.py): where the resident plan's mailboxes live.  out4 = {NUMA node of the command mailbox's first page, of the result
// mailbox's first page (move_pages query; -1: no resident plan yet or not available), NUMA node the device hangs on (sysfs; -1: unknown), CPU the caller runs on}.
int frx_debug_mailbox_numa(const frx_problem *p, int *out4) {
    if (!p || !out4) return fail(FRX_ERR_INVALID_ARG, "null argument");
    auto node_of = [](const void *ptr) -> int {
        if (!ptr) return -1;
        void *page = (void *)((uintptr_t)ptr & ~(uintptr_t)4095);
        int status = -1;
        const long rc = syscall(SYS_move_pages, 0, 1UL, &page, (const int *)nullptr, &status, 0);
        return rc == 0 ? status : -1;
    };
    out4[0] = node_of(p->h_rcmd.p); out4[1] = node_of(p->h_rres.p);
    out4[2] = -1;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), p->device) == hipSuccess) {
        for (char *c = bdf; *c; c++) *c = (char)std::tolower(*c);
        const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
        if (FILE *f = std::fopen(path.c_str(), "r")) { int node = -1; if (std::fscanf(f, "%d", &node) == 1) out4[2] = node; std::fclose(f); }
    }
    out4[3] = sched_getcpu();
    return FRX_OK;
}

int frx_debug_shader_clock(int device, double ms, double *mhz_min, double *mhz_mean, double *mhz_max) {
    if (ms <= 0.0 || ms > 100.0) return fail(FRX_ERR_INVALID_ARG, "frx_debug_shader_clock: 0 < ms <= 100");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int blocks = std::max(1, prop.multiProcessorCount);
    DevBuf<double> out; DevBuf<unsigned long long> st;
    if (out.alloc(1) != hipSuccess || st.alloc((size_t)2 * blocks) != hipSuccess) return fail(FRX_ERR_ALLOC, "clock probe buffers");
    HIP_TRY((hipError_t)frx::launch_clock_probe(out.p, st.p, blocks, (unsigned long long)(ms * 1e5), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)2 * blocks);
    HIP_TRY(hipMemcpy(h.data(), st.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    double lo = 1e30, hi = 0.0, sum = 0.0;
    for (int b = 0; b < blocks; b++) { const double mhz = 100.0 * (double)h[2 * b] / (double)std::max<unsigned long long>(1, h[2 * b + 1]); lo = std::min(lo, mhz); hi = std::max(hi, mhz); sum += mhz; }
    if (mhz_min) *mhz_min = lo;
    if (mhz_mean) *mhz_mean = sum / blocks;
    if (mhz_max) *mhz_max = hi;
    return FRX_OK;
}
int frx_debug_taken_over(const frx_problem *p, int *candidates) {
    if (!p || !candidates) return fail(FRX_ERR_INVALID_ARG, "null argument");
    *candidates = p->taken_over;
    return FRX_OK;
}
int frx_debug_compact_from_history(int m, int n, int hs, int bound, int newest, const double *S, const double *Y, double *rinv129, double *yy, double *vd) {
    if (!S || !Y || !rinv129 || !yy || !vd || m < 1 || m > 128 || n < 1 || hs < n || bound < 0 || bound > m || newest < 0 || newest >= m) return fail(FRX_ERR_INVALID_ARG, "frx_debug_compact_from_history: argument out of range");
    frx::compact_from_history(m, n, (size_t)hs, bound, newest, S, Y, 129, rinv129, yy, vd);
    return FRX_OK;
}
int frx_debug_host_cpu_share(int clusters, int extra_plans, double *budget, int *share, int *mailbox_threads) {
    // (computed from the arguments - the process-wide count of concurrent plans is not touched while real plans may be running)
    if (budget) *budget = frx::host_cpu_budget();
    if (share) *share = frx::host_cpu_share_for(extra_plans);
    if (mailbox_threads) *mailbox_threads = frx::mailbox_threads_for(std::max(1, clusters), extra_plans);
    return FRX_OK;
}

int frx_optimize_stats(const frx_problem *p, double *out4) {
    if (!p || !out4) return fail(FRX_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < 4; i++) out4[i] = p->stats[i];
    return FRX_OK;
}

int frx_lbfgs_minimize_batch(int count, const int *x_off, double *x, double *f_out, int *status, int *iters, int *evals,
                             const frx_lbfgs_params *params, frx_batch_eval_fn eval, void *instance, int n_threads) {
    if (count <= 0 || !x_off || !x || !params || !eval || !status) return fail(FRX_ERR_INVALID_ARG, "null argument or count <= 0");
    const int NX = x_off[count];
    std::vector<double> g(NX, 0.0), f(count, 0.0);
    auto eval_all = [&](int na, const int *ids) -> int {
        eval(instance, na, ids, x, f.data(), g.data());
        return FRX_OK;
    };
    return drive_batch(count, x_off, x, g.data(), f.data(), *params, std::max(1, n_threads), status, iters, evals, f_out, nullptr,
                       eval_all);
}

} // extern "C"
