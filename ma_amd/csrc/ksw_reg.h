// ksw_reg.h -- register-resident variant of the kswcpp wavefront kernel (see ksw_wave.h for the contract
// and the bit-exactness rules; this file changes WHERE the state lives, not what is computed).
//
// One wavefront per DP job.  Diagonal lane t is owned by wave lane (t mod 64); each wave lane keeps S
// "slots" (t = base + 64*c + lane, c < S) of the int8 difference vectors u,v,x,y,x2,y2, the score
// profile s, the target base and the exact score H in VGPRs.  `base` follows the 64-aligned start of
// the active window, so slot contents shift down by one register every 64 diagonals.  x[t-1], v[t-1],
// x2[t-1] and H[t-1] come from the neighbouring lane with wave shuffles (lane 0 takes lane 63 of the
// slot below); chunks that do not intersect the window are skipped wave-uniformly.  Only the reversed
// query (a few hundred bytes) sits in LDS; there is no barrier and no LDS round trip on the diagonal
// loop.  Direction bytes stream to an HBM row per diagonal exactly as in ksw_wave.h and the back-trace
// is shared.
#pragma once
#include "ksw_wave.h"

#if defined( __HIPCC__ )
namespace ma
{
// slots a job needs: aligned window width <= min(qlen, tlen, w+1) + 30, plus up to 63 lanes of base slack
MA_HD i32 ksw_need_slots( i32 qlen, i32 tlen, i32 w )
{
    if( w < 0 )
        w = tlen > qlen ? tlen : qlen;
    i64 m = qlen < tlen ? qlen : tlen;
    if( (i64)w + 1 < m )
        m = (i64)w + 1;
    return (i32)( ( m + 30 + 15 + 63 + 63 ) / 64 ); // +15: the score profile may overshoot en
}

template <int S, typename TH, int HL, typename QF, typename TF>
__device__ void ksw_reg_core( const KswScoring& SC, const KswJobView& J, QF qbase, TF tbase, uint8_t* qr /*LDS*/,
                              uint8_t* P /*HBM direction bytes*/, u32* cig, KswEz& ez, u32& nCigar, u64& cells,
                              u64& pathSteps )
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
    const int8_t sc_mch = (int8_t)( SC.match < 0 ? -SC.match : SC.match );
    const int8_t sc_mis = (int8_t)( SC.mismatch > 0 ? -SC.mismatch : SC.mismatch );
    const i32 qe0 = q + e; // q+e before the swap (used for H[0] on the first diagonal)
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
        const i64 min_sc = sc_mis < 0 ? sc_mis : 0;
        if( -min_sc > 2 * ( q + e ) )
            return;
    }
    const i64 n_col = ksw_ncol( qlen, tlen, J.w ) * 16;
    i64 long_thres = e != e2 ? ( q2 - q ) / ( e - e2 ) - 1 : 0;
    if( q2 + e2 + long_thres * e2 > q + e + long_thres * e )
        ++long_thres;
    const i64 long_diff = long_thres * ( e - e2 ) - ( q2 - q ) - e2;
    const i32 L = ( ( tlen + 15 ) / 16 ) * 16;
    const i32 qrBytes = ( ( qlen + 15 ) / 16 ) * 16 + 32;
    const i32 NEG = sizeof( TH ) == 2 ? -32768 : (i32)0x80000000;
    const i32 cQE = (int8_t)( -q - e ), cQE2 = (int8_t)( -q2 - e2 );
    const i32 vQE = q + e, vQE2 = q2 + e2;

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

    i32 U[ S ], V[ S ], X[ S ], Y[ S ], X2[ S ], Y2[ S ], Sp[ S ], T[ S ], H[ S ];
    i32 base = 0;
    i32 hBelow = NEG; // H[base-1]: the only lane below the base that is ever read again (as H[en0-1])
#pragma unroll
    for( int c = 0; c < S; c++ )
    {
        U[ c ] = V[ c ] = X[ c ] = Y[ c ] = cQE;
        X2[ c ] = Y2[ c ] = cQE2;
        Sp[ c ] = 0;
        T[ c ] = tgtAt( 64 * c + lane );
        H[ c ] = NEG;
    }
    const bool left = !( J.flag & KSW_EZ_RIGHT );
    i64 last_st = -1, last_en = -1;
    const i64 nDiag = (i64)qlen + tlen - 1;
    for( i64 r = 0; r < nDiag; ++r )
    {
        const KswBounds B = ksw_bounds( r, qlen, tlen, w );
        if( B.out )
        {
            ez.zdropped = 1;
            break;
        }
        const i32 st = B.st, en = B.en, st0 = B.st0, en0 = B.en0;
        // ---- carry-in (kswcpp_core.h:562-579): lane st-1 always lives in slot 0 of the OLD base
        i32 x1, x21, v1;
        if( st > 0 )
        {
            if( st - 1 >= last_st && st - 1 <= last_en )
            {
                const int src = ( st - 1 - base ) & 63;
                x1 = __shfl( X[ 0 ], src, 64 );
                x21 = __shfl( X2[ 0 ], src, 64 );
                v1 = __shfl( V[ 0 ], src, 64 );
            }
            else
                x1 = cQE, x21 = cQE2, v1 = cQE;
        }
        else
        {
            x1 = cQE;
            x21 = cQE2;
            v1 = (int8_t)( r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2 );
        }
        // ---- follow the window: shift slots down when the 64-aligned base advances
        const i32 newBase = ( st >> 6 ) << 6;
        if( newBase != base )
        {
            hBelow = __shfl( H[ 0 ], 63, 64 );
#pragma unroll
            for( int c = 0; c + 1 < S; c++ )
            {
                U[ c ] = U[ c + 1 ], V[ c ] = V[ c + 1 ], X[ c ] = X[ c + 1 ], Y[ c ] = Y[ c + 1 ];
                X2[ c ] = X2[ c + 1 ], Y2[ c ] = Y2[ c + 1 ], Sp[ c ] = Sp[ c + 1 ], T[ c ] = T[ c + 1 ], H[ c ] = H[ c + 1 ];
            }
            base = newBase;
            U[ S - 1 ] = V[ S - 1 ] = X[ S - 1 ] = Y[ S - 1 ] = cQE;
            X2[ S - 1 ] = Y2[ S - 1 ] = cQE2;
            Sp[ S - 1 ] = 0;
            T[ S - 1 ] = tgtAt( base + 64 * ( S - 1 ) + lane );
            H[ S - 1 ] = NEG;
        }
        // ---- first row/column initialisation (kswcpp_core.h:580-585)
        if( en >= r )
        {
            const i32 idx = (i32)r - base;
            const i32 uInit = (int8_t)( r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2 );
#pragma unroll
            for( int c = 0; c < S; c++ )
                if( ( idx >> 6 ) == c && ( idx & 63 ) == lane )
                {
                    Y[ c ] = cQE;
                    Y2[ c ] = cQE2;
                    U[ c ] = uInit;
                }
        }
        // ---- score profile over [st0, st0 + cover) (kswcpp_core.h:598-615)
        const i32 cover = ( ( en0 - st0 ) / 16 + 1 ) * 16;
        const i32 qoff = qlen - 1 - (i32)r;
#pragma unroll
        for( int c = 0; c < S; c++ )
        {
            const i32 lo = base + 64 * c;
            if( lo + 63 < st0 || lo >= st0 + cover )
                continue; // wave-uniform
            const i32 tt = lo + lane;
            if( tt >= st0 && tt < st0 + cover )
            {
                const i32 a = T[ c ], b = (i32)qr[ qoff + tt ];
                i32 val = a == b ? sc_mch : sc_mis;
                if( a == 4 || b == 4 )
                    val = (int8_t)( -e2 );
                Sp[ c ] = val;
            }
        }
        // ---- DP cells of the aligned lanes [st,en], highest slot first (old neighbours)
        uint8_t* pr = P + (size_t)( r * n_col ) - st;
        cells += (u64)( en - st + 1 );
        // old H[t-1] of every slot is needed for H[en0]; take it before the row update
        i32 hEn0cand = 0;
        const i32 en0Idx = en0 - base;
#pragma unroll
        for( int c = S - 1; c >= 0; c-- )
        {
            const i32 lo = base + 64 * c;
            if( lo + 63 < st || lo > en )
                continue; // wave-uniform
            const i32 tt = lo + lane;
            // neighbours t-1 (old values)
            i32 xt1 = __shfl_up( X[ c ], 1, 64 ), vt1 = __shfl_up( V[ c ], 1, 64 ), x2t1 = __shfl_up( X2[ c ], 1, 64 );
            i32 hup = __shfl_up( H[ c ], 1, 64 );
            if( c > 0 )
            {
                const i32 bx = __shfl( X[ c > 0 ? c - 1 : 0 ], 63, 64 ), bv = __shfl( V[ c > 0 ? c - 1 : 0 ], 63, 64 ),
                          bx2 = __shfl( X2[ c > 0 ? c - 1 : 0 ], 63, 64 ), bh = __shfl( H[ c > 0 ? c - 1 : 0 ], 63, 64 );
                if( lane == 0 )
                    xt1 = bx, vt1 = bv, x2t1 = bx2, hup = bh;
            }
            else if( lane == 0 )
                hup = hBelow;
            if( tt == st )
                xt1 = x1, vt1 = v1, x2t1 = x21;
            const bool act = tt >= st && tt <= en;
            if( act )
            {
                i32 z = Sp[ c ];
                const i32 ut = U[ c ];
                i32 a = (int8_t)( xt1 + vt1 );
                i32 b = (int8_t)( Y[ c ] + ut );
                i32 a2 = (int8_t)( x2t1 + vt1 );
                i32 b2 = (int8_t)( Y2[ c ] + ut );
                u32 d;
                if( left )
                {
                    d = a > z ? 1 : 0;
                    z = z > a ? z : a;
                    d = b > z ? 2 : d;
                    z = z > b ? z : b;
                    d = a2 > z ? 3 : d;
                    z = z > a2 ? z : a2;
                    d = b2 > z ? 4 : d;
                    z = z > b2 ? z : b2;
                }
                else
                {
                    d = z > a ? 0 : 1;
                    z = z > a ? z : a;
                    d = z > b ? d : 2;
                    z = z > b ? z : b;
                    d = z > a2 ? d : 3;
                    z = z > a2 ? z : a2;
                    z = z > b2 ? z : b2; // state 4 never recorded (kswcpp_core.h:693-699)
                }
                z = z < sc_mch ? z : sc_mch;
                U[ c ] = (int8_t)( z - vt1 );
                V[ c ] = (int8_t)( z - ut );
                i32 tmp = (int8_t)( z - q );
                a = (int8_t)( a - tmp );
                b = (int8_t)( b - tmp );
                tmp = (int8_t)( z - q2 );
                a2 = (int8_t)( a2 - tmp );
                b2 = (int8_t)( b2 - tmp );
                if( left )
                {
                    X[ c ] = (int8_t)( ( a > 0 ? a : 0 ) - vQE );
                    d |= a > 0 ? 0x08 : 0;
                    Y[ c ] = (int8_t)( ( b > 0 ? b : 0 ) - vQE );
                    d |= b > 0 ? 0x10 : 0;
                    X2[ c ] = (int8_t)( ( a2 > 0 ? a2 : 0 ) - vQE2 );
                    d |= a2 > 0 ? 0x20 : 0;
                    Y2[ c ] = (int8_t)( ( b2 > 0 ? b2 : 0 ) - vQE2 );
                    d |= b2 > 0 ? 0x40 : 0;
                }
                else
                {
                    X[ c ] = (int8_t)( ( 0 > a ? 0 : a ) - vQE );
                    d |= 0 > a ? 0 : 0x08;
                    Y[ c ] = (int8_t)( ( 0 > b ? 0 : b ) - vQE );
                    d |= 0 > b ? 0 : 0x10;
                    X2[ c ] = (int8_t)( ( 0 > a2 ? 0 : a2 ) - vQE2 );
                    d |= 0 > a2 ? 0 : 0x20;
                    Y2[ c ] = (int8_t)( ( 0 > b2 ? 0 : b2 ) - vQE2 );
                    d |= 0 > b2 ? 0 : 0x40;
                }
                pr[ tt ] = (uint8_t)d;
            }
            // H[en0] = en0 > 0 ? Hold[en0-1] + u[en0] : Hold[en0] + v[en0] (new u / v of this diagonal)
            if( ( en0Idx >> 6 ) == c && ( en0Idx & 63 ) == lane )
                hEn0cand = (TH)( en0 > 0 ? hup + U[ c ] : H[ c ] + V[ c ] );
        }
        // ---- calcMaxScore (kswcpp_core.h:156-299)
        i32 max_H, max_t;
        const i32 hEn0 = __shfl( hEn0cand, en0Idx & 63, 64 );
        i32 hSt0 = 0; // new H[st0] for the mqe test
        if( r > 0 )
        {
            const i32 en1 = st0 + ( ( en0 - st0 ) / HL ) * HL;
            i32 bh = (i32)0x80000000, bc = 0x7fffffff; // SIMD part: (h desc, chunk asc)
            i32 th = (i32)0x80000000, tt_ = 0x7fffffff; // scalar tail: (h desc, t asc)
            i32 st0cand = 0;
#pragma unroll
            for( int c = 0; c < S; c++ )
            {
                const i32 lo = base + 64 * c;
                if( lo + 63 < st0 || lo >= en0 + 1 )
                    continue; // wave-uniform
                const i32 t = lo + lane;
                if( t >= st0 && t < en0 )
                {
                    const i32 h = (TH)( H[ c ] + V[ c ] );
                    H[ c ] = h;
                    if( t < en1 )
                    {
                        const i32 k = ( t - st0 ) / HL;
                        if( h > bh || ( h == bh && k < bc ) )
                            bh = h, bc = k;
                    }
                    else if( h > th || ( h == th && t < tt_ ) )
                        th = h, tt_ = t;
                }
                if( t == en0 )
                    H[ c ] = hEn0;
                if( t == st0 )
                    st0cand = H[ c ];
            }
            hSt0 = __shfl( st0cand, ( st0 - base ) & 63, 64 );
            // lanes of one SIMD class share (lane + base - st0) mod HL, i.e. lane mod HL
            for( int m = HL; m < 64; m <<= 1 )
                ksw_red_pair( bh, bc, m );
            i32 vH = hEn0, vT = en0;
            if( bh > vH )
                vH = bh, vT = st0 + bc * HL;
            i32 mh = vH, mt = vT;
            for( int m = 1; m < HL; m <<= 1 )
            {
                const i32 oh = __shfl_xor( mh, m, 64 ), ot = __shfl_xor( mt, m, 64 );
                mh = oh > mh ? oh : mh;
                mt = ot > mt ? ot : mt;
            }
            max_H = __shfl( mh, 0, 64 );
            max_t = __shfl( mt, 0, 64 );
            // scalar remainder [en1,en0): first position of the maximum, taken only if it beats max_H
            for( int m = 1; m < 64; m <<= 1 )
                ksw_red_pair( th, tt_, m );
            if( th > max_H )
                max_H = th, max_t = tt_;
        }
        else
        {
            // r == 0: H[0] = v[0] - (q+e) (kswcpp_core.h:244-249); lane 0 slot 0 owns t = 0
            i32 h0 = (TH)( V[ 0 ] - qe0 );
            h0 = __shfl( h0, 0, 64 );
            if( lane == 0 )
                H[ 0 ] = h0;
            max_H = h0;
            max_t = 0;
            hSt0 = h0;
        }
        const i32 hEnd = r > 0 ? hEn0 : hSt0; // H[en0] after the update
        if( en0 == tlen - 1 && hEnd > ez.mte )
            ez.mte = hEnd, ez.mte_q = (i32)( r - en );
        {
            const i32 hS = st0 == en0 ? hEnd : hSt0;
            if( r - st0 == qlen - 1 && hS > ez.mqe )
                ez.mqe = hS, ez.mqe_t = st0;
        }
        bool stop = false;
        {
            const i32 Hm = max_H, t = max_t;
            if( Hm > (i32)ez.max )
            {
                ez.max = (u32)Hm & 0x7fffffffu;
                ez.max_t = t;
                ez.max_q = (i32)r - t;
            }
            else if( t >= ez.max_t && (i32)r - t >= ez.max_q )
            {
                const i32 tl = t - ez.max_t, ql = ( (i32)r - t ) - ez.max_q;
                const i32 l = tl > ql ? tl - ql : ql - tl;
                if( J.zdrop >= 0 && (i32)( ez.max - (u32)Hm ) > J.zdrop + l * e2 )
                {
                    ez.zdropped = 1;
                    stop = true;
                }
            }
        }
        if( stop )
            break;
        if( r == (i64)qlen + tlen - 2 && en0 == tlen - 1 )
            ez.score = hEnd;
        last_st = st;
        last_en = en;
    }
    __syncthreads( ); // direction bytes of all lanes visible to the back-tracing lane
    i32 i0 = -1, j0 = -1;
    if( !ez.zdropped && !( J.flag & KSW_EZ_EXTZ_ONLY ) )
        i0 = tlen - 1, j0 = qlen - 1;
    else if( !ez.zdropped && ( J.flag & KSW_EZ_EXTZ_ONLY ) && ez.mqe > (i32)ez.max )
    {
        ez.reach_end = 1;
        i0 = ez.mqe_t, j0 = qlen - 1;
    }
    else if( ez.max_t >= 0 && ez.max_q >= 0 )
        i0 = ez.max_t, j0 = ez.max_q;
    else
        return;
    ksw_backtrack_lane0( P, cig, n_col, qlen, tlen, w, J.flag, i0, j0, nCigar, pathSteps );
}
} // namespace ma
#endif
