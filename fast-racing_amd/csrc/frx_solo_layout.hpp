// LDS layout of the solo launch (frx_solo_kernel.hpp), shared by the kernel, the launcher's size computation and the CPU-side layout test (tests/test_hostcheck.py):
// plain C++, no HIP.
#pragma once

#if !defined(__HIPCC__) && !defined(__host__)
#define __host__
#define __device__
#define FRX_SOLO_LAYOUT_DEFINED_QUALIFIERS
#endif

namespace frx {

enum { SOLO_RB = 24 * 64 + 4 };              // row buffer of the bodies: (D^-1, L) rows of two buffers + the matrix wave's progress words (forward_knot_body<.., RB>)
enum { SOLO_QV = 10, SOLO_QS = SOLO_QV + 1 }; // partials per transpose half, row stride of the transpose square

// LDS of a workgroup (doubles): (C, T) copy | x | multipliers | waypoint sums | polytopes | scratch of the bodies; the last two = the penalty phase's corridor blocks and transpose square
struct SoloLds { int ctl, xs, vs, pw, wq, ev, total; };
__host__ __device__ inline SoloLds solo_lds(int maxN, int maxXb, int maxVb, int maxCN, int nsteps, int ppg, int Kmax) {
    SoloLds L;
    int o = 0;
    L.ctl = o; o += (maxN * 19 + 1) & ~1;
    L.xs = o; o += (maxXb + 1) & ~1;
    L.pw = o; o += ((nsteps * 8 + 5) * 64 + 1) & ~1;
    L.wq = o; o += 4 * 64;
    L.vs = o; o += (maxVb + 1) & ~1;                                          // (the polytopes LAST in front of the scratch: the penalty phase takes both, the polytopes are staged again behind it)
    L.ev = o;
    const int body = SOLO_RB + 9 * 65 + 2 * 64 + maxCN + 16;                  // rows | knot arrays | Tf, gT | gCo | cross-wave partials (the adjoint's layout is the larger one)
    o += (body + 1) & ~1;
    const int pen = (ppg < maxN ? ppg : maxN) * (Kmax + 1) * 4 + 256 * SOLO_QS + 2;   // one pass of corridor blocks | [256][11] transpose square, from L.vs on
    if (o - L.vs < pen) o = L.vs + ((pen + 1) & ~1);
    L.total = o;
    return L;
}


} // namespace frx

#ifdef FRX_SOLO_LAYOUT_DEFINED_QUALIFIERS
#undef __host__
#undef __device__
#undef FRX_SOLO_LAYOUT_DEFINED_QUALIFIERS
#endif
