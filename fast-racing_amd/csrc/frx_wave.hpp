// Wave-level reductions on gfx950 (wave64) through DPP row operations and the LDS crossbar - no LDS memory, no barrier.
// Every function sums in a FIXED order, so results are reproducible run to run.
#pragma once
#include <hip/hip_runtime.h>

namespace frx {

template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_add(double v) {
    // With every row enabled and bound_ctrl set the "old" operand is dead and the compiler emits a bare v_mov_b32_dpp; with old = 0
    // and bound_ctrl clear it first zeroes the destination (one extra v_mov per half, 24 per reduction).  The two row-broadcast
    // stages write only some rows and need the zero for the others.
    constexpr bool full = ROW_MASK == 0xf;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, full);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, full);
    return v + __hiloint2double(hi, lo);
}
// Sum over the 64 lanes, result broadcast to every lane (as a wave-uniform value).
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = dpp_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);     // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);     // row_mirror        -> every lane holds the sum of its row of 16
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3 -> row 3 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Four sums over the 64 lanes at once, results in every lane.  Separate wave_sum_dpp calls cost 4 x (12 DPP moves + 6 adds + 2
// readlanes) and measured ~900 cycles per block of the recursion; here the values are PACKED while they are reduced: after the
// xor-1 and xor-2 exchanges each lane carries one value (class = lane & 3) summed over its quad, two row rotations sum the quads of
// a row, the rows are combined through the LDS crossbar (ds_swizzle xor 16, ds_bpermute xor 32; no LDS memory involved), and four
// quad broadcasts hand every lane all four totals.  Fixed order, so deterministic.
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// `spread` = false leaves total k in lane k (k < 4) only - enough when the totals go to LDS anyway.
template <bool spread> __device__ __forceinline__ double wave_sum4_packed(double (&v)[4]) {
    const int lane = threadIdx.x & 63;
    const bool odd = lane & 1, up = lane & 2;
    const double x = odd ? v[1] : v[0], px = odd ? v[0] : v[1];
    const double y = odd ? v[3] : v[2], py = odd ? v[2] : v[3];
    const double r0 = x + dpp_mov<0xB1>(px);                       // even lanes: v0 of the pair, odd lanes: v1
    const double r1 = y + dpp_mov<0xB1>(py);                       // even lanes: v2, odd lanes: v3
    const double z = up ? r1 : r0, pz = up ? r0 : r1;
    double q = z + dpp_mov<0x4E>(pz);                              // class lane & 3 summed over the quad
    q += dpp_mov<0x124>(q);                                        // row_ror:4
    q += dpp_mov<0x128>(q);                                        // row_ror:8  -> summed over the row of 16
    {
        const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(q), 0x401F), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(q), 0x401F);   // lane ^ 16
        q += __hiloint2double(hi, lo);
    }
    {
        const int addr = ((lane ^ 32) << 2);
        const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(q)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(q));      // lane ^ 32
        q += __hiloint2double(hi, lo);
    }
    if (spread) { v[0] = dpp_mov<0x00>(q); v[1] = dpp_mov<0x55>(q); v[2] = dpp_mov<0xAA>(q); v[3] = dpp_mov<0xFF>(q); }
    return q;
}


// Neighbour exchange by ONE lane as a DPP move (wave_shr:1 / wave_shl:1, 2 instructions per double) instead of __shfl_up / __shfl_down, which
// go through the LDS crossbar (address arithmetic + 2 ds_bpermute + a wait for their round trip).  Same semantics: the lane without a
// neighbour (0 resp. 63) keeps its own value.
__device__ __forceinline__ double lane_up1(double v) {            // lane i <- lane i - 1
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_down1(double v) {          // lane i <- lane i + 1
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// sum over the 4 lanes of a quad, result in every lane of the quad (two quad permutes; replaces two ds_bpermute round trips)
__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);          // quad_perm [2,3,0,1]
    return v;
}

} // namespace frx
