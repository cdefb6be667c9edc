// ksw_reg.h -- cross-lane helpers of the register-resident kswcpp kernels (ksw_pk.h: exact kernel, two cells per
// lane; ksw_ext.h: the pipeline's extension kernel) and the statement + proof of the early stop both use.
// (The first register-resident kernel of this project lived here: one diagonal cell per lane in a ring of 64-lane
// slots, DPP wave shifts for the t-1 neighbours, v_readlane broadcasts; ksw_pk.h is its two-cells-per-lane successor.)
#pragma once
#include "ksw_wave.h"

#if defined( __HIPCC__ )
namespace ma
{
// ---- cross-lane helpers (gfx950) ------------------------------------------------------------------
__device__ __forceinline__ i32 dpp_wave_shr1( i32 x ) // lane i <- lane i-1 (lane 0 patched by the caller)
{
    return __builtin_amdgcn_update_dpp( 0, x, 0x138, 0xf, 0xf, false );
}
__device__ __forceinline__ i32 dpp_wave_ror1( i32 x ) // lane i <- lane i-1, lane 0 <- lane 63
{
    return __builtin_amdgcn_mov_dpp( x, 0x13C, 0xf, 0xf, false ); // every lane is written: no `old` to initialise
}
template <int CTRL> __device__ __forceinline__ i32 dpp_ctrl( i32 x )
{
    return __builtin_amdgcn_update_dpp( 0, x, CTRL, 0xf, 0xf, true ); // folds into the consuming VALU op
}
__device__ __forceinline__ i32 lane_bcast( i32 x, int srcLane ) // wave-uniform source lane
{
    return __builtin_amdgcn_readlane( x, srcLane );
}
// (h desc, k asc) combine
__device__ __forceinline__ void best_pair( i32& h, i32& k, i32 oh, i32 ok )
{
    if( oh > h || ( oh == h && ok < k ) )
    {
        h = oh;
        k = ok;
    }
}

__device__ __forceinline__ i32 wave_max_i32( i32 v )
{
    v = max( v, dpp_ctrl<0x121>( v ) ); // row_ror:1
    v = max( v, dpp_ctrl<0x122>( v ) ); // row_ror:2
    v = max( v, dpp_ctrl<0x124>( v ) ); // row_ror:4
    v = max( v, dpp_ctrl<0x128>( v ) ); // row_ror:8
    return max( max( lane_bcast( v, 0 ), lane_bcast( v, 16 ) ), max( lane_bcast( v, 32 ), lane_bcast( v, 48 ) ) );
}

// EARLY (pipeline mode, extension jobs only): stop as soon as no later diagonal can raise ez.max.
// The callers of the extension (NeedlemanWunsch::dynPrg / ksw_dual_ext, needlemanWunsch.cpp:499-622,
// 392-497) read only max_q, max_t and the cigar traced back from (max_t, max_q), and kswcpp moves that
// cell only when a diagonal's maximum is strictly greater than ez.max (kswcpp_core.h:22-44).  Every cell
// obeys H(i,j) = H(i-1,j-1) + z with z <= match (kswcpp_core.h:703, the min with sc_mch_), and with
// qlen <= w+1 the lower band edge is the last query row, so the diagonal predecessor chain of any later
// cell stays inside the band until it reaches diagonal r or r-1 or the first-row boundary H(i,-1).
// Hence for r >= qlen:  later H <= max( B_r, B_{r-1}, H(r-1,-1) + match*qlen ),
// B_d = max over cells (t, d-t) of diagonal d of  H + match * min(qlen-1-(d-t), tlen-1-t).
// When that bound is <= ez.max the remaining diagonals cannot change max/max_q/max_t; the other ez
// fields (mqe, mte, score, zdropped) are then unspecified, which is why ma_ksw_batch never uses EARLY.
//
// Bands that cut the rectangle (qlen > w+1; ksw_pk.h only).  The band [st0, en0] = [(r-w+1)>>1, (r+w)>>1] (clipped to the
// rectangle) is a set of DP diagonals: st0(r) - 1 = st0(r-2) and en0(r) - 1 = en0(r-2), so the diagonal predecessor
// (t-1, r-2) of an in-band cell (t, r) is in the band, or is a first-row / first-column boundary cell.  The H kswcpp
// tracks for the cells of [st0, en0] -- H(t,r) = H(t,r-1) + v(t,r), and H(en0,r) = H(en0-1,r-1) + u(en0,r) for the cell
// that enters -- satisfies H(t,r) = H(t-1,r-2) + z(t,r) there whatever stale values the cells at the band edge read from
// outside the band (those only enter z through a, b, a2, b2, and z is clipped to <= match afterwards): with
// u(t,r) = z - v(t-1,r-1) and v(t,r) = z - u(t,r-1), H(t-1,r-1) - v(t-1,r-1) = H(t-1,r-2) = H(t,r-1) - u(t,r-1), because
// the v / u on both sides are the stored values of in-band cells.  A chain that starts at a boundary cell has offset
// |t - q| <= w, i.e. its first cell lies on a diagonal <= w + 1.  Hence for r >= w + 3 every later in-band cell has its
// chain pass through diagonal r or r-1 and  later H <= max( B_r, B_{r-1} ).  The row q = qlen-1 is made of later cells
// as well, so mqe cannot exceed ez.max either and the back-trace starts at (max_t, max_q) as it does when the band runs
// out (zdropped = 1, kswcpp_core.h:541-559).
} // namespace ma
#endif
