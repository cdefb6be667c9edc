// ksw_reg.h -- register-resident "ring" variant of the kswcpp wavefront kernel (see ksw_wave.h for the
// contract and the bit-exactness rules; this file changes WHERE the state lives, not what is computed).
//
// One wavefront per DP job.  Diagonal lane t is owned by wave lane (t mod 64) in register slot
// ((t / 64) mod R): the 64*R lanes [st, st + 64*R) that the reference can touch on a diagonal form a
// ring that rotates with the 16-aligned window start `st`; when st advances, the 16 lanes that fell
// out of the window are recycled (re-initialised exactly like the reference's freshly cleared scratch)
// for the 16 lanes that enter at the top.  All per-cell state -- the int8 difference vectors
// u,v,x,y,x2,y2, the score profile s, the target base and the exact score H -- lives in VGPRs with
// static register indices; x[t-1], v[t-1], x2[t-1], H[t-1] come from the neighbouring lane through a
// DPP wave shift (lane 0 takes the previous slot's lane 63, snapshotted with v_readlane before any
// update); the 8-/4-lane calcMaxScore reduction uses DPP row rotates; scalars are broadcast with
// v_readlane.  Only the reversed query sits in LDS; there is no barrier and no LDS round trip in the
// diagonal loop.  Direction bytes stream to an HBM row per diagonal exactly as in ksw_wave.h and the
// back-trace is shared.
#pragma once
#include "ksw_wave.h"

#if defined( __HIPCC__ )
namespace ma
{
// register slots a job needs: everything the reference touches on a diagonal lies in [st, st + m + 29]
// with m = min(qlen, tlen, w+1) (aligned band + up to 15 lanes of score-profile overshoot)
MA_HD i32 ksw_need_slots( i32 qlen, i32 tlen, i32 w )
{
    if( w < 0 )
        w = tlen > qlen ? tlen : qlen;
    i64 m = qlen < tlen ? qlen : tlen;
    if( (i64)w + 1 < m )
        m = (i64)w + 1;
    return (i32)( ( m + 30 + 63 ) / 64 );
}

// ---- cross-lane helpers (gfx950) ------------------------------------------------------------------
__device__ __forceinline__ i32 dpp_wave_shr1( i32 x ) // lane i <- lane i-1 (lane 0 patched by the caller)
{
    return __builtin_amdgcn_update_dpp( 0, x, 0x138, 0xf, 0xf, false );
}
__device__ __forceinline__ i32 dpp_wave_ror1( i32 x ) // lane i <- lane i-1, lane 0 <- lane 63
{
    return __builtin_amdgcn_update_dpp( 0, x, 0x13C, 0xf, 0xf, false );
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
template <int R, typename TH, int HL, bool EARLY, typename QF, typename TF>
__device__ void ksw_reg_core( const KswScoring& SC, const KswJobView& J, QF qbase, TF tbase, uint8_t* qr /*LDS*/,
                              uint8_t* P /*HBM direction bytes*/, u32* cig, KswEz& ez, u32& nCigar, u64& cells,
                              u64& pathSteps, u32 ldsBytes )
{
    const int lane = threadIdx.x & 63;
    const i32 qlen = J.qlen, tlen = J.tlen;
    ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
    ez.max = 0;
    ez.score = ez.mqe = ez.mte = (i32)0x80000000;
    ez.zdropped = 0;
    ez.reach_end = 0;
    nCigar = 0;
    cells = 0;
    pathSteps = 0;
    if( qlen <= 0 || tlen <= 0 )
        return;
    int8_t q = (int8_t)SC.q, e = (int8_t)SC.e, q2 = (int8_t)SC.q2, e2 = (int8_t)SC.e2;
    const i32 sc_mch = (int8_t)( SC.match < 0 ? -SC.match : SC.match );
    const i32 sc_mis = (int8_t)( SC.mismatch > 0 ? -SC.mismatch : SC.mismatch );
    const i32 qe0 = q + e; // q+e before the swap (H[0] on the first diagonal)
    if( q2 + e2 < q + e )
    {
        int8_t t = q;
        q = q2;
        q2 = t;
        t = e;
        e = e2;
        e2 = t;
    }
    i32 w = J.w;
    if( w < 0 )
        w = tlen > qlen ? tlen : qlen;
    {
        const i32 min_sc = sc_mis < 0 ? sc_mis : 0;
        if( -min_sc > 2 * ( q + e ) )
            return;
    }
    const i32 n_col = (i32)ksw_ncol( qlen, tlen, J.w ) * 16;
    i32 long_thres = e != e2 ? ( q2 - q ) / ( e - e2 ) - 1 : 0;
    if( q2 + e2 + long_thres * e2 > q + e + long_thres * e )
        ++long_thres;
    const i32 long_diff = long_thres * ( e - e2 ) - ( q2 - q ) - e2;
    const i32 L = ( ( tlen + 15 ) / 16 ) * 16;
    const i32 qrBytes = ( ( qlen + 15 ) / 16 ) * 16 + 32;
    const i32 NEG = sizeof( TH ) == 2 ? -32768 : (i32)0x80000000;
    const i32 cQE = (int8_t)( -q - e ), cQE2 = (int8_t)( -q2 - e2 );
    const i32 vQ = q, vQ2 = q2, vQE = q + e, vQE2 = q2 + e2, vNE2 = (int8_t)( -e2 );

    for( i32 t = lane; t < qrBytes; t += 64 )
        qr[ t ] = t < qlen ? (uint8_t)qbase( qlen - 1 - t ) : (uint8_t)0;
    __syncthreads( );

    // target byte as the reference's contiguous scratch sees it: sf[t] for t < L, then the qr region
    auto tgtAt = [ & ]( i32 tt ) -> i32 {
        if( tt < tlen )
            return (i32)tbase( tt );
        if( tt < L )
            return 0;
        const i32 k = tt - L;
        return k < qrBytes ? (i32)qr[ k ] : 0;
    };

    i32 U[ R ], V[ R ], X[ R ], Y[ R ], X2[ R ], Y2[ R ], Sp[ R ], T[ R ], H[ R ], TT[ R ];
#pragma unroll
    for( int s = 0; s < R; s++ )
    {
        TT[ s ] = 64 * s + lane;
        U[ s ] = V[ s ] = X[ s ] = Y[ s ] = cQE;
        X2[ s ] = Y2[ s ] = cQE2;
        Sp[ s ] = 0;
        T[ s ] = tgtAt( TT[ s ] );
        H[ s ] = NEG;
    }
    const bool left = !( J.flag & KSW_EZ_RIGHT );
    i32 last_st = -1, last_en = -1, cur_st = 0;
    i32 hBelow = NEG; // H[st-1]: the only recycled lane that is read again (as H[en0-1] when en0 == st)
    const i32 nDiag = qlen + tlen - 1;
    bool stop = false;
    const bool early = EARLY && ( J.flag & KSW_EZ_EXTZ_ONLY ) && qlen <= w + 1;
    i32 topH = 0, boundPrev = 0x7fffffff; // H(r-1,-1) of the first-row boundary; B_{r-1}
    for( i32 r = 0; r < nDiag && !stop; ++r )
    {
        // ---- bounds (kswcpp_core.h:541-559); 32 bit is enough since r < 2^31
        i32 st0 = 0, en0 = tlen - 1;
        st0 = max( st0, r - qlen + 1 );
        en0 = min( en0, r );
        st0 = max( st0, ( r - w + 1 ) >> 1 );
        en0 = min( en0, ( r + w ) >> 1 );
        if( st0 > en0 )
        {
            ez.zdropped = 1;
            break;
        }
        const i32 st = st0 & ~15, en = en0 | 15;
        // ---- carry-in (kswcpp_core.h:562-579) and ring rotation; st advances by 16 at most
        i32 x1 = cQE, x21 = cQE2, v1 = cQE;
        if( st == 0 )
            v1 = (int8_t)( r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2 );
        if( st != cur_st )
        {
            const int jOld = ( cur_st >> 6 ) % R; // slot that holds [cur_st, cur_st + 16)
            const int src = ( st - 1 ) & 63;
            const bool useOld = st - 1 >= last_st && st - 1 <= last_en;
#pragma unroll
            for( int s = 0; s < R; s++ )
                if( s == jOld )
                {
                    if( useOld )
                    {
                        x1 = lane_bcast( X[ s ], src );
                        x21 = lane_bcast( X2[ s ], src );
                        v1 = lane_bcast( V[ s ], src );
                    }
                    hBelow = lane_bcast( H[ s ], src );
                    // recycle the 16 lanes that left the window for the 16 lanes entering at the top
                    if( TT[ s ] < st )
                    {
                        TT[ s ] += 64 * R;
                        U[ s ] = V[ s ] = X[ s ] = Y[ s ] = cQE;
                        X2[ s ] = Y2[ s ] = cQE2;
                        Sp[ s ] = 0;
                        H[ s ] = NEG;
                        T[ s ] = tgtAt( TT[ s ] );
                    }
                }
            cur_st = st;
        }
        const i32 uInit = (int8_t)( r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2 );
        const bool initRow = en >= r; // kswcpp_core.h:580-585
        const i32 pEnd = st0 + ( ( en0 - st0 ) / 16 + 1 ) * 16; // score profile refreshes [st0, pEnd)
        const i32 qoff = qlen - 1 - r;
        uint8_t* pr = P + (size_t)r * (size_t)n_col - st;
        cells += (u64)( en - st + 1 );
        const i32 hi = max( en, pEnd - 1 );
        const i32 en1 = st0 + ( ( en0 - st0 ) / HL ) * HL;
        const int b0 = st >> 6, j0 = b0 % R;
        // old lane-63 values of every slot: lane 0 of slot s continues lane 63 of slot s-1
        i32 l63x[ R ], l63v[ R ], l63x2[ R ], l63h[ R ];
        if( R > 1 )
        {
#pragma unroll
            for( int s = 0; s < R; s++ )
            {
                l63x[ s ] = lane_bcast( X[ s ], 63 );
                l63v[ s ] = lane_bcast( V[ s ], 63 );
                l63x2[ s ] = lane_bcast( X2[ s ], 63 );
                l63h[ s ] = lane_bcast( H[ s ], 63 );
            }
        }
        i32 hEn0c = 0, hSt0c = 0; // owner-lane candidates
        i32 laneMax = (i32)0x80000000; // largest new H of this lane's cells in [st0, en0)
#pragma unroll
        for( int s = 0; s < R; s++ )
        {
            // wave-uniform skip of slots whose lanes are all above the touched range
            int dj = s - j0;
            if( dj < 0 )
                dj += R;
            if( dj != 0 && ( ( b0 + dj ) << 6 ) > hi )
                continue;
            const int sp = s == 0 ? R - 1 : s - 1;
            const i32 tt = TT[ s ];
            // neighbours t-1 (values of the previous diagonal)
            i32 xt1, vt1, x2t1, hup;
            if( R == 1 )
            {
                xt1 = dpp_wave_ror1( X[ s ] );
                vt1 = dpp_wave_ror1( V[ s ] );
                x2t1 = dpp_wave_ror1( X2[ s ] );
                hup = dpp_wave_ror1( H[ s ] );
            }
            else
            {
                xt1 = dpp_wave_shr1( X[ s ] );
                vt1 = dpp_wave_shr1( V[ s ] );
                x2t1 = dpp_wave_shr1( X2[ s ] );
                hup = dpp_wave_shr1( H[ s ] );
                if( lane == 0 )
                    xt1 = l63x[ sp ], vt1 = l63v[ sp ], x2t1 = l63x2[ sp ], hup = l63h[ sp ];
            }
            if( tt == st )
                xt1 = x1, vt1 = v1, x2t1 = x21, hup = hBelow;
            // first row / column initialisation
            if( initRow && tt == r )
            {
                Y[ s ] = cQE;
                Y2[ s ] = cQE2;
                U[ s ] = uInit;
            }
            // score profile
            if( tt >= st0 && tt < pEnd )
            {
                const i32 a = T[ s ], b = (i32)qr[ qoff + tt ];
                i32 val = a == b ? sc_mch : sc_mis;
                if( a == 4 || b == 4 )
                    val = vNE2;
                Sp[ s ] = val;
            }
            // DP cell (kswcpp_core.h:653-766)
            {
                i32 z = Sp[ s ];
                const i32 ut = U[ s ];
                i32 a = (int8_t)( xt1 + vt1 );
                i32 b = (int8_t)( Y[ s ] + ut );
                i32 a2 = (int8_t)( x2t1 + vt1 );
                i32 b2 = (int8_t)( Y2[ s ] + ut );
                i32 d;
                if( left )
                {
                    d = a > z ? 1 : 0;
                    z = max( z, a );
                    d = b > z ? 2 : d;
                    z = max( z, b );
                    d = a2 > z ? 3 : d;
                    z = max( z, a2 );
                    d = b2 > z ? 4 : d;
                    z = max( z, b2 );
                }
                else
                {
                    d = z > a ? 0 : 1;
                    z = max( z, a );
                    d = z > b ? d : 2;
                    z = max( z, b );
                    d = z > a2 ? d : 3;
                    z = max( z, a2 );
                    z = max( z, b2 ); // state 4 never recorded (kswcpp_core.h:693-699)
                }
                z = min( z, sc_mch );
                const i32 nu = (int8_t)( z - vt1 ), nv = (int8_t)( z - ut );
                i32 tmp = (int8_t)( z - vQ );
                a = (int8_t)( a - tmp );
                b = (int8_t)( b - tmp );
                tmp = (int8_t)( z - vQ2 );
                a2 = (int8_t)( a2 - tmp );
                b2 = (int8_t)( b2 - tmp );
                const i32 nx = (int8_t)( max( a, 0 ) - vQE ), ny = (int8_t)( max( b, 0 ) - vQE );
                const i32 nx2 = (int8_t)( max( a2, 0 ) - vQE2 ), ny2 = (int8_t)( max( b2, 0 ) - vQE2 );
                if( left )
                {
                    d |= a > 0 ? 0x08 : 0;
                    d |= b > 0 ? 0x10 : 0;
                    d |= a2 > 0 ? 0x20 : 0;
                    d |= b2 > 0 ? 0x40 : 0;
                }
                else
                {
                    d |= 0 > a ? 0 : 0x08;
                    d |= 0 > b ? 0 : 0x10;
                    d |= 0 > a2 ? 0 : 0x20;
                    d |= 0 > b2 ? 0 : 0x40;
                }
                if( tt >= st && tt <= en )
                {
                    U[ s ] = nu;
                    V[ s ] = nv;
                    X[ s ] = nx;
                    Y[ s ] = ny;
                    X2[ s ] = nx2;
                    Y2[ s ] = ny2;
                    pr[ tt ] = (uint8_t)d;
                }
            }
            // ---- calcMaxScore pieces (kswcpp_core.h:156-299)
            // H[en0] = en0 > 0 ? Hold[en0-1] + u[en0] : Hold[en0] + v[en0] with this diagonal's u / v
            if( tt == en0 )
                hEn0c = (TH)( en0 > 0 ? hup + U[ s ] : H[ s ] + V[ s ] );
            if( r > 0 && tt >= st0 && tt < en0 )
            {
                const i32 h = (TH)( H[ s ] + V[ s ] );
                H[ s ] = h;
                laneMax = max( laneMax, h );
            }
            if( tt == st0 )
                hSt0c = H[ s ];
        }
        i32 max_H, max_t, hEnd, hS;
        if( r > 0 )
        {
            const i32 hEn0 = lane_bcast( hEn0c, en0 & 63 );
#pragma unroll
            for( int s = 0; s < R; s++ )
                if( TT[ s ] == en0 )
                    H[ s ] = hEn0;
            hEnd = hEn0;
            hS = st0 == en0 ? hEn0 : lane_bcast( hSt0c, st0 & 63 );
            // The exact (max_H, max_t) of calcMaxScore is only consumed when the diagonal raises ez.max or could
            // z-drop (ksw_apply_zdrop needs nothing else); two ballot tests decide that, the reduction below runs
            // only then.  max_H = -inf otherwise makes the update code a no-op.
            max_H = (i32)0x80000000;
            max_t = 0;
            bool need = hEn0 > (i32)ez.max || __any( laneMax > (i32)ez.max ) != 0;
            if( !need && J.zdrop >= 0 )
                need = hEn0 < (i32)ez.max - J.zdrop && __any( laneMax >= (i32)ez.max - J.zdrop ) == 0;
            if( need )
            {
                i32 bh = (i32)0x80000000, bk = 0x7fffffff; // SIMD part: (h desc, chunk asc)
                i32 tailH = (i32)0x80000000; // scalar remainder [en1, en0): one lane per cell
#pragma unroll
                for( int s = 0; s < R; s++ )
                {
                    const i32 tt = TT[ s ];
                    if( tt >= st0 && tt < en0 )
                    {
                        if( tt < en1 )
                            best_pair( bh, bk, H[ s ], ( tt - st0 ) / HL );
                        else
                            tailH = H[ s ];
                    }
                }
                // all-reduce over the lanes of one SIMD class (equal lane mod HL)
                if( HL == 4 )
                    best_pair( bh, bk, dpp_ctrl<0x124>( bh ), dpp_ctrl<0x124>( bk ) ); // row_ror:4
                best_pair( bh, bk, dpp_ctrl<0x128>( bh ), dpp_ctrl<0x128>( bk ) ); // row_ror:8
                best_pair( bh, bk, __shfl_xor( bh, 16, 64 ), __shfl_xor( bk, 16, 64 ) );
                best_pair( bh, bk, __shfl_xor( bh, 32, 64 ), __shfl_xor( bk, 32, 64 ) );
                i32 vH = hEn0, vT = en0; // the initial (H[en0], en0) wins ties
                if( bh > vH )
                    vH = bh, vT = st0 + bk * HL;
                // independent horizontal maxima over the HL classes (values repeat with period HL lanes)
                i32 mh = vH, mt = vT;
                mh = max( mh, dpp_ctrl<0xB1>( mh ) ); // quad_perm [1,0,3,2]
                mt = max( mt, dpp_ctrl<0xB1>( mt ) );
                mh = max( mh, dpp_ctrl<0x4E>( mh ) ); // quad_perm [2,3,0,1]
                mt = max( mt, dpp_ctrl<0x4E>( mt ) );
                if( HL == 8 )
                {
                    mh = max( mh, dpp_ctrl<0x124>( mh ) ); // row_ror:4 (period-8 values: quad q-1 == quad q+1)
                    mt = max( mt, dpp_ctrl<0x124>( mt ) );
                }
                max_H = __builtin_amdgcn_readfirstlane( mh );
                max_t = __builtin_amdgcn_readfirstlane( mt );
                // scalar remainder [en1, en0): ascending t, strict >
                for( i32 t = en1; t < en0; ++t )
                {
                    const i32 h = lane_bcast( tailH, t & 63 );
                    if( h > max_H )
                        max_H = h, max_t = t;
                }
            }
        }
        else
        {
            // r == 0: H[0] = v[0] - (q+e) (kswcpp_core.h:244-249); lane 0 of slot 0 owns t = 0
            const i32 h0 = (TH)( lane_bcast( V[ 0 ], 0 ) - qe0 );
            if( lane == 0 )
                H[ 0 ] = h0;
            max_H = h0;
            max_t = 0;
            hEnd = h0;
            hS = h0;
        }
        if( en0 == tlen - 1 && hEnd > ez.mte )
            ez.mte = hEnd, ez.mte_q = r - en;
        if( r - st0 == qlen - 1 && hS > ez.mqe )
            ez.mqe = hS, ez.mqe_t = st0;
        // ksw_apply_zdrop (kswcpp_core.h:22-44), is_rot = 1
        if( max_H > (i32)ez.max )
        {
            ez.max = (u32)max_H & 0x7fffffffu;
            ez.max_t = max_t;
            ez.max_q = r - max_t;
        }
        else if( max_H != (i32)0x80000000 && max_t >= ez.max_t && r - max_t >= ez.max_q )
        {
            const i32 tl = max_t - ez.max_t, ql = ( r - max_t ) - ez.max_q;
            const i32 l = tl > ql ? tl - ql : ql - tl;
            if( J.zdrop >= 0 && (i32)( ez.max - (u32)max_H ) > J.zdrop + l * e2 )
            {
                ez.zdropped = 1;
                stop = true;
            }
        }
        if( !stop && r == qlen + tlen - 2 && en0 == tlen - 1 )
            ez.score = hEnd;
        if( EARLY && early && r >= qlen - 1 )
        {
            i32 bnd = (i32)0x80000000;
#pragma unroll
            for( int s = 0; s < R; s++ )
            {
                const i32 tt = TT[ s ];
                if( tt >= st0 && tt <= en0 )
                    bnd = max( bnd, H[ s ] + sc_mch * min( qoff + tt, tlen - 1 - tt ) );
            }
            bnd = wave_max_i32( bnd );
            if( r >= qlen && max( max( bnd, boundPrev ), topH + sc_mch * qlen ) <= (i32)ez.max )
                stop = true;
            boundPrev = bnd;
        }
        topH += uInit; // H(r,-1)
        last_st = st;
        last_en = en;
    }
    __syncthreads( ); // direction bytes of all lanes visible to the back-tracing lane
    i32 i0 = -1, j0b = -1;
    if( !ez.zdropped && !( J.flag & KSW_EZ_EXTZ_ONLY ) )
        i0 = tlen - 1, j0b = qlen - 1;
    else if( !ez.zdropped && ( J.flag & KSW_EZ_EXTZ_ONLY ) && ez.mqe > (i32)ez.max )
    {
        ez.reach_end = 1;
        i0 = ez.mqe_t, j0b = qlen - 1;
    }
    else if( ez.max_t >= 0 && ez.max_q >= 0 )
        i0 = ez.max_t, j0b = ez.max_q;
    else
        return;
    ksw_backtrack_lane0( P, cig, (i64)n_col, qlen, tlen, w, J.flag, i0, j0b, nCigar, pathSteps, qr, ldsBytes );
}
} // namespace ma
#endif
