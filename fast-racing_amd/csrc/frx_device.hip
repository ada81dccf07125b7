// hipcc translation unit: the gfx950 stage kernels, the L-BFGS vector kernels, the corridor kernel and their launchers.
// (The resident round kernel and the one-launch evaluation have translation units of their own - frx_device_round.hip, frx_device_eval.hip - so that the three
// compile side by side: k_round's six instantiations alone take longer than everything else together.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "frx_kernels.hpp"
#include "frx_lbfgs_kernels.hpp"
#include "frx_corridor_kernels.hpp"

namespace frx {

// The dynamic-LDS limit of a kernel is a property of the FUNCTION, not of a handle: handles of different geometry live side by side (tests, a planner with several
// corridors), so the limit only ever grows - a handle created later with a smaller need must not lower it under an older handle's launches.
static int raise_lds_limit(const void *fn, size_t bytes, size_t &held) {
    if (bytes <= held) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) held = bytes;
    return (int)e;
}
int launch_set_limits(const LaunchGeom &g) {
    static std::mutex mu;
    static size_t held_all[64][13] = {};                             // per device: a function's attributes belong to the device that is current
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    size_t *held = held_all[dev];
    int e;
    if ((e = raise_lds_limit((const void *)k_forward, g.lds_fwd, held[0]))) return e;
    if ((e = raise_lds_limit((const void *)k_backward, g.lds_bwd, held[1]))) return e;
    if ((e = raise_lds_limit((const void *)k_penalty, g.lds_pen, held[2]))) return e;
    if ((e = raise_lds_limit((const void *)k_penalty_lat, g.lds_pen, held[3]))) return e;
    if (g.lds_pen2) {                                                  // the instantiation launch_penalty takes for this geometry
        const void *fn = g.lpp == 17 ? (const void *)k_penalty_lat2<17> : g.lpp == 49 ? (const void *)k_penalty_lat2<49> : (const void *)k_penalty_lat2<0>;
        if ((e = raise_lds_limit(fn, g.lds_pen2, held[g.lpp == 17 ? 4 : g.lpp == 49 ? 10 : 11]))) return e;
        if ((e = raise_lds_limit((const void *)k_penalty_lat2_r5, g.lds_pen2, held[12]))) return e;
    }
    if ((e = raise_lds_limit((const void *)k_forward_knot, g.lds_kfwd, held[5]))) return e;
    if ((e = raise_lds_limit((const void *)k_backward_knot, g.lds_kbwd, held[6]))) return e;
    if ((e = raise_lds_limit((const void *)k_forward_knot64, g.lds_kfwd, held[7]))) return e;
    if ((e = raise_lds_limit((const void *)k_backward_knot64, g.lds_kbwd, held[8]))) return e;
    if (g.ev_G && (e = eval_cluster_raise_limit(g.lds_ev))) return e;                   // (frx_device_eval.hip: the kernel lives in that translation unit)
    return 0;
}
int launch_forward(const DevProblem &dp, const LaunchGeom &g, const double *x, double *T, double *C, double *band, void *stream) {
    if (g.solver == SOLVER_KNOT_PCR)
        if (g.knot_threads == 64) hipLaunchKernelGGL(k_forward_knot64, dim3(dp.B), dim3(256), g.lds_kfwd, (hipStream_t)stream, dp, x, T, C, g.maxCN, g.maxXb, g.maxVb, g.pcrw, g.pcr_steps);
        else hipLaunchKernelGGL(k_forward_knot, dim3(dp.B), dim3(256), g.lds_kfwd, (hipStream_t)stream, dp, x, T, C, g.maxCN, g.maxXb, g.maxVb, g.knot_threads, g.pcrw, g.pcr_steps);
    else
        hipLaunchKernelGGL(k_forward, dim3(dp.B), dim3(64), g.lds_fwd, (hipStream_t)stream, dp, x, T, C, band, g.maxN, g.maxCN);
    return (int)hipGetLastError();
}
int launch_penalty(const DevProblem &dp, const LaunchGeom &g, const double *T, const double *C, double *out20, void *stream) {
    // default: the latency form (148 VGPRs, 3 waves per SIMD, phases interleaved by the scheduler) - measured faster at the headline batch
    // (4.97 vs 5.23 us) AND at 1024 candidates (39.8 vs 42.3 us) than the throughput form (126 VGPRs, 4 waves per SIMD); FRX_PENALTY_FORM=thr selects that one
    static const int forced = [] { const char *e = std::getenv("FRX_PENALTY_FORM"); return !e ? 0 : e[0] == 'l' ? 1 : 2; }();
    const bool lat = forced != 2;
    const int nwg = (dp.P + g.ppg - 1) / g.ppg;
    const char *tp_env = std::getenv("FRX_PENALTY_TWOPHASE");           // (read per launch: the test toggles it inside one process)
    const bool two_phase = !(tp_env && tp_env[0] == '0');
    if (lat && g.lds_pen2 && tp_env && tp_env[0] == '5') hipLaunchKernelGGL(k_penalty_lat2_r5, dim3(nwg), dim3(256), g.lds_pen2, (hipStream_t)stream, dp, T, C, out20, g.lpp, g.ppg, g.Kmax);   // (A/B: the round-5 form)
    else if (lat && two_phase && g.lds_pen2) {                         // (pen_w == 4: 256 threads; the two boundary resolutions kappa = 16 / 48 have instantiations of their own)
        if (g.lpp == 17) hipLaunchKernelGGL(k_penalty_lat2<17>, dim3(nwg), dim3(256), g.lds_pen2, (hipStream_t)stream, dp, T, C, out20, g.lpp, g.ppg, g.Kmax);
        else if (g.lpp == 49) hipLaunchKernelGGL(k_penalty_lat2<49>, dim3(nwg), dim3(256), g.lds_pen2, (hipStream_t)stream, dp, T, C, out20, g.lpp, g.ppg, g.Kmax);
        else hipLaunchKernelGGL(k_penalty_lat2<0>, dim3(nwg), dim3(256), g.lds_pen2, (hipStream_t)stream, dp, T, C, out20, g.lpp, g.ppg, g.Kmax);
    }
    else if (lat) hipLaunchKernelGGL(k_penalty_lat, dim3(nwg), dim3(64 * g.pen_w), g.lds_pen, (hipStream_t)stream, dp, T, C, out20, g.lpp, g.ppg, g.Kmax);
    else hipLaunchKernelGGL(k_penalty, dim3(nwg), dim3(64 * g.pen_w), g.lds_pen, (hipStream_t)stream, dp, T, C, out20, g.lpp, g.ppg, g.Kmax);
    return (int)hipGetLastError();
}
int launch_backward(const DevProblem &dp, const LaunchGeom &g, const double *x, const double *T, const double *C,
                    const double *band, const double *out20, double *f, double *grad, void *stream, const double *tap_d, const int *tap_flags,
                    void *tap_res, unsigned *tap_arrive, volatile unsigned *tap_flag, unsigned tap_round) {
    if (g.solver == SOLVER_KNOT_PCR && g.knot_threads == 64)
        hipLaunchKernelGGL(k_backward_knot64, dim3(dp.B), dim3(256), g.lds_kbwd, (hipStream_t)stream, dp, x, T, C, out20, f,
                           grad, g.maxCN, g.maxXb, g.maxVb, g.pcrw, g.pcr_steps,
                           LineSearchTap{tap_d, tap_flags, (DvResult *)tap_res, tap_arrive, tap_flag, tap_round});
    else if (g.solver == SOLVER_KNOT_PCR)
        hipLaunchKernelGGL(k_backward_knot, dim3(dp.B), dim3(256), g.lds_kbwd, (hipStream_t)stream, dp, x, T, C, out20, f,
                           grad, g.maxCN, g.maxXb, g.maxVb, g.knot_threads, g.pcrw, g.pcr_steps,
                           LineSearchTap{tap_d, tap_flags, (DvResult *)tap_res, tap_arrive, tap_flag, tap_round});
    else
        hipLaunchKernelGGL(k_backward, dim3(dp.B), dim3(64), g.lds_bwd, (hipStream_t)stream, dp, x, T, C, band, out20, f, grad, g.maxN,
                           g.maxCN);
    return (int)hipGetLastError();
}


static DvBuffers to_buffers(const DvLaunch &dv) {
    DvBuffers b;
    b.xoff = dv.xoff; b.x = dv.x; b.g = dv.g; b.xp = dv.xp; b.gp = dv.gp; b.d = dv.d; b.S = dv.S; b.Y = dv.Y; b.ys = dv.ys; b.gt = dv.gt; b.dflags = dv.dflags; b.pflags = dv.pflags; b.poff = dv.poff; b.m = dv.m; b.B = dv.B; b.hs = dv.hs ? (int)dv.hs : 64 * dv.W * dv.E;
    return b;
}
int launch_lbfgs_pre(const DvLaunch &dv, const void *cmd, void *res, void *stream) {
    const DvBuffers b = to_buffers(dv);
    const DvCommand *c = (const DvCommand *)cmd; DvResult *r = (DvResult *)res;
    hipStream_t st = (hipStream_t)stream;
#define FRX_PRE(E_, W_, PF_, BLK_) case ((E_ * 16 + W_) * 64 + PF_) * 8 + BLK_: hipLaunchKernelGGL((k_lbfgs_pre<E_, W_, PF_, BLK_>), dim3(dv.B), dim3(64 * W_), 0, st, b, c, r); break;
    switch (((dv.E * 16 + dv.W) * 64 + dv.PF) * 8 + dv.BLK) {
    FRX_PRE(2, 1, 16, 4) FRX_PRE(2, 2, 16, 4) FRX_PRE(2, 3, 16, 4) FRX_PRE(2, 4, 16, 4) FRX_PRE(2, 5, 16, 4) FRX_PRE(2, 6, 16, 4) FRX_PRE(2, 7, 16, 4) FRX_PRE(2, 8, 16, 4)
    FRX_PRE(4, 1, 8, 4) FRX_PRE(4, 2, 8, 4) FRX_PRE(4, 3, 8, 4) FRX_PRE(4, 4, 8, 4) FRX_PRE(4, 5, 8, 4) FRX_PRE(4, 6, 8, 4) FRX_PRE(4, 7, 8, 4) FRX_PRE(4, 8, 8, 4)
    FRX_PRE(6, 1, 8, 4) FRX_PRE(6, 2, 8, 4) FRX_PRE(6, 3, 8, 4) FRX_PRE(6, 4, 8, 4) FRX_PRE(6, 5, 8, 4) FRX_PRE(6, 6, 8, 4) FRX_PRE(6, 7, 8, 4) FRX_PRE(6, 8, 8, 4)
    FRX_PRE(8, 1, 4, 4) FRX_PRE(8, 2, 4, 4) FRX_PRE(8, 3, 4, 4) FRX_PRE(8, 4, 4, 4) FRX_PRE(8, 5, 4, 4) FRX_PRE(8, 6, 4, 4) FRX_PRE(8, 7, 4, 4) FRX_PRE(8, 8, 4, 4)
    FRX_PRE(4, 3, 8, 1) FRX_PRE(4, 3, 8, 2) FRX_PRE(2, 6, 16, 1)     // experiments (FRX_DV_GEOM)
    default: return (int)hipErrorInvalidValue;
    }
#undef FRX_PRE
    return (int)hipGetLastError();
}
int launch_lbfgs_post(const DvLaunch &dv, const double *f, const void *cmd, void *res, void *stream) {
    hipLaunchKernelGGL(k_lbfgs_post, dim3(dv.B), dim3(64), 0, (hipStream_t)stream, to_buffers(dv), f, (const DvCommand *)cmd, (DvResult *)res);
    return (int)hipGetLastError();
}


// Diagnostic (bench): the shader clock the device SUSTAINS under a latency-bound FP64 load like the round kernel's - one lone wave per CU on every CU running a
// dependent FMA chain for ~`ms` milliseconds - as shader cycles (s_memtime) per tick of the constant 100 MHz counter.  Boxes of the pool differ by 3-6 % per round
// with the SAME code object (profiles/NOTES.md); a round is a chain of dependent instructions, so its time is cycles / clock, and this is the clock.
__global__ __launch_bounds__(64) void k_clock_probe(double *out, unsigned long long *stamps, unsigned long long ticks) {
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999999;
    unsigned long long w1 = w0;
    while (w1 - w0 < ticks) {
#pragma unroll
        for (int i = 0; i < 256; i++) a = __builtin_fma(a, b, 1e-12);
        w1 = wall_clock64();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
    if (a == 12345.678) out[0] = a;                                      // (keeps the chain)
}
int launch_clock_probe(double *out, unsigned long long *stamps, int blocks, unsigned long long ticks, void *stream) {
    hipLaunchKernelGGL(k_clock_probe, dim3(blocks), dim3(64), 0, (hipStream_t)stream, out, stamps, ticks);
    return (int)hipGetLastError();
}

size_t dilate_lds_bytes(int pcap) { return sizeof(double) * ((size_t)3 * pcap + 32 + 36 + 16) + sizeof(int) * ((size_t)2 * pcap + 257 + 3); }
int launch_dilate(const DilateLaunch &d, void *stream) {
    DilateArgs a;
    a.p1 = d.p1; a.p2 = d.p2; a.obs = d.obs; a.bbox[0] = d.bbox[0]; a.bbox[1] = d.bbox[1]; a.bbox[2] = d.bbox[2]; a.offset = d.offset;
    a.S = d.S; a.n_obs = d.n_obs; a.cap_planes = d.cap_planes; a.pcap = d.pcap;
    a.n_planes = d.n_planes; a.h_rec = d.h_rec; a.ell_C = d.ell_C; a.ell_d = d.ell_d;
    const size_t lds = dilate_lds_bytes(d.pcap);
    hipError_t e = hipFuncSetAttribute((const void *)k_dilate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_dilate, dim3(d.S), dim3(256), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

} // namespace frx
