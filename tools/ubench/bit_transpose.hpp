// bit_transpose.hpp — 64 x 64 bit-matrix transpose across the lanes of a wave (gfx950).
// Before: lane c holds the 64-bit word {lo, hi} whose bit r is M[c][r].  After: lane r holds the word whose bit c is M[c][r].
// Recursive block swap: the two off-diagonal 32 x 32 blocks change places with one v_permlane32_swap (gfx950), then each
// 32 x 32 block is transposed in place inside its register and its 32 lanes by five stages s = 16, 8, 4, 2, 1: a lane
// rotates its word so that the half-blocks it gives away sit where the partner (lane ^ s) wants them (v_alignbit with a
// per-lane amount), the words cross by ds_swizzle, and a v_bfi keeps the lane's own half-blocks.  23 VALU operations
// and 10 swizzles per 4096 bits, against one ballot + two v_writelane per row (192 half-rate operations) before.
#pragma once
#include <hip/hip_runtime.h>

template <int S>
__device__ __forceinline__ unsigned dg_bt_stage(unsigned x, int lane)
{
    constexpr unsigned M = S == 16 ? 0x0000ffffu : S == 8 ? 0x00ff00ffu : S == 4 ? 0x0f0f0f0fu : S == 2 ? 0x33333333u : 0x55555555u;
    const bool up = (lane & S) != 0;
    // up: keeps the bits with (bit & S) != 0 and hands its other bits over shifted LEFT by S (rotate right by 32 - S);
    // down: keeps the bits with (bit & S) == 0 and hands the others over shifted RIGHT by S.  What a rotation wraps around
    // lands in positions the receiver masks off.
    const unsigned amt = up ? 32 - S : S;
    const unsigned send = __builtin_amdgcn_alignbit(x, x, amt);
    const unsigned recv = (unsigned)__builtin_amdgcn_ds_swizzle((int)send, (S << 10) | 0x1f);     // lane ^ S
    const unsigned keep = up ? ~M : M;
    return (x & keep) | (recv & ~keep);
}

__device__ __forceinline__ void dg_bit_transpose64(unsigned &lo, unsigned &hi, int lane)
{
    // lanes 0..31 give their hi (rows 32..63 of columns 0..31) for the lo of lanes 32..63 (rows 0..31 of columns 32..63)
    auto sw = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);      // vdst = lo: its lanes 32..63 <-> src0 = hi: its lanes 0..31
    lo = sw[0];
    hi = sw[1];
    // now {lo of lanes 0..31, hi of lanes 32..63} are the diagonal blocks and {hi of 0..31, lo of 32..63} hold the swapped ones;
    lo = dg_bt_stage<16>(lo, lane); hi = dg_bt_stage<16>(hi, lane);
    lo = dg_bt_stage<8>(lo, lane);  hi = dg_bt_stage<8>(hi, lane);
    lo = dg_bt_stage<4>(lo, lane);  hi = dg_bt_stage<4>(hi, lane);
    lo = dg_bt_stage<2>(lo, lane);  hi = dg_bt_stage<2>(hi, lane);
    lo = dg_bt_stage<1>(lo, lane);  hi = dg_bt_stage<1>(hi, lane);
}
