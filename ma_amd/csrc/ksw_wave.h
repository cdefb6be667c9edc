// ksw_wave.h -- kswcpp banded two-piece-affine DP (libs/kswcpp/inc/kswcpp_core.h:308-841) for gfx950:
// one wavefront per DP job, the anti-diagonal's lanes t mapped onto the 64 lanes of the wave, the
// int8 difference vectors u,v,x,y,x2,y2, the score profile s, the target/query copies and the exact
// score row H staged in LDS (or in HBM scratch when a job's rows exceed the LDS budget), direction
// bytes streamed to an HBM scratch row per diagonal (coalesced, one byte per lane), back-trace by a
// single lane.  No MFMA: nothing here is a contraction.
//
// Bit-exactness with the reference's 16-lane SSE4.1 path is by construction: every 16-aligned lane
// block [st,en] the reference computes is computed here with the same wrapping int8 arithmetic and
// the same loop bounds, including the unaligned score-profile stride (kswcpp_core.h:598-615), the
// aligned carry-in rule (562-579), the 8-/4-lane calcMaxScore reduction with its independent
// horizontal maxima (156-299), the dangling-else on state 4 (693-699) and mte_q from the aligned en.
#pragma once
#include "ma_common.h"

#if defined( __HIPCC__ )
namespace ma
{
#if defined( MA_KSW_PROF )
static __device__ unsigned long long g_ksw_prof[ 16 ]; // phase cycle counters (diagnostics build only)
#endif
#define KSW_EZ_RIGHT 0x02
#define KSW_EZ_EXTZ_ONLY 0x40
#define KSW_EZ_REV_CIGAR 0x80

struct KswScoring
{
    i32 match, mismatch, q, e, q2, e2; // KswCppParam<5> (kswcpp.h:44-129)
    // not a score: short extensions may share a wavefront (ksw_grp.h; ksw_job_class_pipe).  0 off, 1 queries up to 64 bases, 2 also
    // 65..128 with four rows per lane (A/B), 1000 + n: extensions of n..254 query bases on the proven narrow band (ksw_band.h)
    i32 grp = 1;
    i32 band_mis = 5; // mismatches on the main diagonal up to which a job is tried on the narrow band (ksw_band_likely; MA_KSW_BAND_MAXMIS: tuning hook)
    i32 band_long = 1; // long extension jobs (queries beyond 254 bases) one per wavefront on the proven band of 120 (ksw_band.h; MA_KSW_BANDL=0: A/B hook)
};

// Working storage of one job (flat pointers: LDS or HBM)
struct KswMem
{
    int8_t* u; // u|v|x|y|x2|y2|s|sf|qr contiguous, L = tlen_*16 each, qr = qlen_*16 (+32 zero slack)
    i32 L;
    void* H; // L entries of int16 or int32
    uint8_t* p; // direction bytes: (qlen+tlen-1) * ncol
    u32* cig; // qlen + tlen + 2 entries
    uint8_t* stage; // LDS the back-trace may stage direction rows in once the diagonal loop is done (or null)
    u32 stageBytes;
};

struct KswJobView // what the kernel needs to fetch the sequences of one job
{
    i32 qlen, tlen, w, zdrop, flag;
};

MA_HD u64 ksw_state_bytes( i32 qlen, i32 tlen )
{
    const u64 L = (u64)( ( tlen + 15 ) / 16 ) * 16;
    return L * 8 + (u64)( ( qlen + 15 ) / 16 ) * 16 + 32;
}
MA_HD bool ksw_h16( const KswScoring& S, i32 qlen, i32 tlen ) // riskOfOverflow<int16_t> (kswcpp.h:101-115)
{
    i32 mn = S.mismatch > 0 ? -S.mismatch : S.mismatch;
    mn = mmin( mn, -S.q );
    mn = mmin( mn, -S.e );
    mn = mmin( mn, -S.q2 );
    mn = mmin( mn, -S.e2 );
    const i64 sz = mmax( qlen, tlen );
    const i64 mx = S.match < 0 ? -S.match : S.match;
    return !( sz * (int8_t)mn < -32768 || sz * (int8_t)mx > 32767 );
}
MA_HD i64 ksw_ncol( i32 qlen, i32 tlen, i32 w ) // n_col_ (kswcpp_core.h:400-402) in 16-lane blocks
{
    if( w < 0 )
        w = tlen > qlen ? tlen : qlen;
    i64 n = qlen < tlen ? qlen : tlen;
    return ( ( n < w + 1 ? n : w + 1 ) + 15 ) / 16 + 1;
}

// Diagonals kswcpp can run before the band leaves the rectangle (kswcpp_core.h:541-553: the loop ends at the first r with
// st0 > en0): (r - w + 1) >> 1 > tlen - 1 from r = 2 tlen + w - 1 on, r - qlen + 1 > (r + w) >> 1 from r ~ 2 qlen + w on.
// An end extension of a long read -- query = the rest of the read, target = 1000 padded reference bases, band 512 -- runs
// 2 500 diagonals however long the query is; direction-matrix scratch and the query window in LDS follow THIS bound.
MA_HD i64 ksw_max_diags( i32 qlen, i32 tlen, i32 w )
{
    if( w < 0 )
        w = tlen > qlen ? tlen : qlen;
    const i64 nd = (i64)qlen + tlen - 1;
    const i64 lim = 2 * (i64)( qlen < tlen ? qlen : tlen ) + w + 2;
    return nd < lim ? nd : lim;
}
// bytes of LDS the reversed query of a job takes in the register kernel (ksw_pk.h): a 64-byte head plus the window of the
// bases a diagonal can reach
MA_HD i64 ksw_q_lds( i32 qlen, i32 tlen, i32 w )
{
    const i64 all = ( ( (i64)qlen + 15 ) / 16 ) * 16 + 32;
    const i64 win = ksw_max_diags( qlen, tlen, w ) + 64 + 64 + 48;
    return all < win ? all : win;
}

struct KswBounds
{
    i32 st, en, st0, en0;
    bool out;
};
__device__ __forceinline__ KswBounds ksw_bounds( i64 r, i32 qlen, i32 tlen, i64 w ) // kswcpp_core.h:541-559
{
    KswBounds b;
    i64 st = 0, en = tlen - 1;
    if( st < r - qlen + 1 )
        st = r - qlen + 1;
    if( en > r )
        en = r;
    if( st < ( ( r - w + 1 ) >> 1 ) )
        st = ( r - w + 1 ) >> 1;
    if( en > ( ( r + w ) >> 1 ) )
        en = ( r + w ) >> 1;
    b.out = st > en;
    b.st0 = (i32)st;
    b.en0 = (i32)en;
    b.st = (i32)( ( st / 16 ) * 16 );
    b.en = (i32)( ( en + 16 ) / 16 * 16 - 1 );
    return b;
}

struct KswEz
{
    u32 max;
    i32 zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end;
};

// (h desc, chunk asc) reduction used by the calcMaxScore emulation
__device__ __forceinline__ void ksw_red_pair( i32& h, i32& c, int laneMask )
{
    const i32 oh = __shfl_xor( h, laneMask, 64 );
    const i32 oc = __shfl_xor( c, laneMask, 64 );
    if( oh > h || ( oh == h && oc < c ) )
    {
        h = oh;
        c = oc;
    }
}

// ksw_backtrack__ (kswcpp_core.h:76-150); off[r] / off_end[r] are recomputed from r.  The walk is
// wave-uniform (every value is made scalar with v_readfirstlane, so it runs on the scalar unit); the
// direction rows it is about to cross are staged from HBM into `stage` (LDS, stageBytes, may be 0) by all
// 64 lanes, a block of rows at a time, so a step costs an LDS read instead of an HBM round trip.  The
// current cigar run is kept in registers and written once per run.  Leaves the CIGAR in cig[0..nCigar).
// RINGROWS: rows are n_col (a power of two) bytes and cell (r, i) sits at column i mod n_col (ksw_ext.h).
template <bool RINGROWS = false>
__device__ __forceinline__ void ksw_backtrack_lane0( const uint8_t* P, u32* cig, i64 n_col, i32 qlen, i32 tlen, i32 w,
                                                     i32 flag, i32 i0, i32 j0, u32& nCigar, u64& pathSteps,
                                                     uint8_t* stage = nullptr, u32 stageBytes = 0 )
{
    const int lane = threadIdx.x & 63;
    const i32 rowsCap = ( stage && n_col * 2 <= (i64)stageBytes ) ? (i32)( (i64)stageBytes / n_col ) : 0;
    i32 rlo = 1, rhi = 0; // staged rows [rlo, rhi]
    u32 n = 0, curOp = 0xffffffffu, curLen = 0, steps = 0;
    auto push = [ & ]( u32 op, u32 len ) {
        if( op == curOp )
            curLen += len;
        else
        {
            if( curOp != 0xffffffffu )
            {
                if( lane == 0 )
                    cig[ n ] = curLen << 4 | curOp;
                n++;
            }
            curOp = op;
            curLen = len;
        }
    };
    i32 i = __builtin_amdgcn_readfirstlane( i0 ), j = __builtin_amdgcn_readfirstlane( j0 ), state = 0;
    while( i >= 0 && j >= 0 )
    {
        int force_state = -1;
        const i32 r = i + j;
        const KswBounds B = ksw_bounds( r, qlen, tlen, w );
        if( i < B.st )
            force_state = 2;
        if( i > B.en )
            force_state = 1;
        u32 tmp = 0;
        if( force_state < 0 )
        {
            if( rowsCap )
            {
                if( r < rlo || r > rhi )
                {
                    __syncthreads( );
                    rhi = r;
                    rlo = r - rowsCap + 1 > 0 ? r - rowsCap + 1 : 0;
                    const i64 bytes = (i64)( rhi - rlo + 1 ) * n_col;
                    const uint8_t* src = P + (i64)rlo * n_col;
                    for( i64 k = (i64)lane * 16; k < bytes; k += 1024 )
                        *(uint4*)( stage + k ) = *(const uint4*)( src + k );
                    __syncthreads( );
                }
                const i64 col = RINGROWS ? (i64)( i & (i32)( n_col - 1 ) ) : (i64)( i - B.st );
                tmp = (u32)__builtin_amdgcn_readfirstlane( (i32)stage[ (i64)( r - rlo ) * n_col + col ] );
            }
            else
            {
                const i64 col = RINGROWS ? (i64)( i & (i32)( n_col - 1 ) ) : (i64)( i - B.st );
                tmp = (u32)__builtin_amdgcn_readfirstlane( (i32)P[ (i64)r * n_col + col ] );
            }
        }
        if( state == 0 )
            state = tmp & 7;
        else if( !( tmp >> ( state + 2 ) & 1 ) )
            state = 0;
        if( state == 0 )
            state = tmp & 7;
        if( force_state >= 0 )
            state = force_state;
        steps++;
        if( state == 0 )
            push( 0, 1 ), --i, --j;
        else if( state == 1 || state == 3 )
            push( 2, 1 ), --i;
        else
            push( 1, 1 ), --j;
    }
    if( i >= 0 )
        push( 2, (u32)( i + 1 ) );
    if( j >= 0 )
        push( 1, (u32)( j + 1 ) );
    push( 0xfffffffeu, 0 ); // flush the last run
    pathSteps += steps;
    __syncthreads( );
    if( !( flag & KSW_EZ_REV_CIGAR ) )
    {
        for( u32 a = (u32)lane; a < ( n >> 1 ); a += 64 )
        {
            const u32 t = cig[ a ];
            cig[ a ] = cig[ n - 1 - a ];
            cig[ n - 1 - a ] = t;
        }
        __syncthreads( );
    }
    nCigar = n;
}

// One job on one wave (blockDim.x == 64). query/target are fetched through functors so the caller
// decides where bases come from (plain byte arrays for ma_ksw_batch, read + 2-bit pack for the pipeline).
template <typename TH, int HL, typename QF, typename TF>
__device__ void ksw_wave_core( const KswScoring& SC, const KswJobView& J, QF qbase, TF tbase, const KswMem& M, KswEz& ez,
                               u32& nCigar, u64& cells, u64& pathSteps )
{
    const int lane = threadIdx.x & 63;
    const i32 qlen = J.qlen, tlen = J.tlen;
    // ksw_reset_extz
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
    const i64 qe = q + e;
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
    const i32 L = M.L;
    int8_t* u8 = M.u;
    int8_t* v8 = u8 + L;
    int8_t* x8 = v8 + L;
    int8_t* y8 = x8 + L;
    int8_t* x28 = y8 + L;
    int8_t* y28 = x28 + L;
    int8_t* s8 = y28 + L;
    uint8_t* sf = (uint8_t*)( s8 + L );
    uint8_t* qr = sf + L;
    const i32 qrBytes = ( ( qlen + 15 ) / 16 ) * 16 + 32;
    TH* H = (TH*)M.H;
    // init (kswcpp_core.h:466-491)
    for( i32 t = lane; t < L; t += 64 )
    {
        u8[ t ] = v8[ t ] = x8[ t ] = y8[ t ] = (int8_t)( -q - e );
        x28[ t ] = y28[ t ] = (int8_t)( -q2 - e2 );
        s8[ t ] = 0;
        sf[ t ] = t < tlen ? (uint8_t)tbase( t ) : (uint8_t)0;
        H[ t ] = (TH)( sizeof( TH ) == 2 ? -32768 : (i32)0x80000000 );
    }
    for( i32 t = lane; t < qrBytes; t += 64 )
        qr[ t ] = t < qlen ? (uint8_t)qbase( qlen - 1 - t ) : (uint8_t)0;
    __syncthreads( );
    const bool left = !( J.flag & KSW_EZ_RIGHT );
    i64 last_st = -1, last_en = -1;
    bool stop = false;
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
        int8_t x1, x21, v1;
        if( st > 0 )
        {
            if( st - 1 >= last_st && st - 1 <= last_en )
                x1 = x8[ st - 1 ], x21 = x28[ st - 1 ], v1 = v8[ st - 1 ];
            else
                x1 = (int8_t)( -q - e ), x21 = (int8_t)( -q2 - e2 ), v1 = (int8_t)( -q - e );
        }
        else
        {
            x1 = (int8_t)( -q - e );
            x21 = (int8_t)( -q2 - e2 );
            v1 = (int8_t)( r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2 );
        }
        if( en >= r && lane == 0 )
        {
            y8[ r ] = (int8_t)( -q - e );
            y28[ r ] = (int8_t)( -q2 - e2 );
            u8[ r ] = (int8_t)( r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2 );
        }
        // score profile from the unaligned st0 in 16-lane strides; loads before stores
        {
            const uint8_t* qrr = qr + ( qlen - 1 - r );
            const i32 cover = ( ( en0 - st0 ) / 16 + 1 ) * 16;
            // a store may alias sf[0..14] only in the last stride, whose loads precede it; all loads of
            // this wave are issued before any store below
            for( i32 base = 0; base < cover; base += 64 )
            {
                const i32 k = base + lane;
                int8_t val = 0;
                const bool act = k < cover;
                if( act )
                {
                    const uint8_t a = sf[ st0 + k ], b = qrr[ st0 + k ];
                    val = a == b ? sc_mch : sc_mis;
                    if( a == 4 || b == 4 )
                        val = (int8_t)( -e2 );
                }
                __syncthreads( );
                if( act )
                    s8[ st0 + k ] = val;
            }
        }
        __syncthreads( );
        // DP over the aligned lanes [st,en], highest 64-lane chunk first so that x[t-1], v[t-1] are
        // still the previous diagonal's values when a chunk reads them
        uint8_t* pr = M.p + (size_t)( r * n_col ) - st;
        const i32 nl = en - st + 1;
        cells += (u64)nl;
        for( i32 cb = ( ( nl - 1 ) / 64 ) * 64; cb >= 0; cb -= 64 )
        {
            const i32 tt = st + cb + lane;
            const bool act = cb + lane < nl;
            int8_t z = 0, xt1 = 0, vt1 = 0, x2t1 = 0, ut = 0, yy = 0, yy2 = 0;
            if( act )
            {
                z = s8[ tt ];
                if( tt == st )
                    xt1 = x1, vt1 = v1, x2t1 = x21;
                else
                    xt1 = x8[ tt - 1 ], vt1 = v8[ tt - 1 ], x2t1 = x28[ tt - 1 ];
                ut = u8[ tt ];
                yy = y8[ tt ];
                yy2 = y28[ tt ];
            }
            __syncthreads( );
            if( act )
            {
                int8_t a = (int8_t)( xt1 + vt1 );
                int8_t b = (int8_t)( yy + ut );
                int8_t a2 = (int8_t)( x2t1 + vt1 );
                int8_t b2 = (int8_t)( yy2 + ut );
                uint8_t d;
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
                    z = z > b2 ? z : b2; // state 4 never recorded (dangling else, kswcpp_core.h:693-699)
                }
                z = z < sc_mch ? z : sc_mch;
                u8[ tt ] = (int8_t)( z - vt1 );
                v8[ tt ] = (int8_t)( z - ut );
                int8_t tmp = (int8_t)( z - q );
                a = (int8_t)( a - tmp );
                b = (int8_t)( b - tmp );
                tmp = (int8_t)( z - q2 );
                a2 = (int8_t)( a2 - tmp );
                b2 = (int8_t)( b2 - tmp );
                if( left )
                {
                    x8[ tt ] = (int8_t)( ( a > 0 ? a : 0 ) - ( q + e ) );
                    d |= a > 0 ? 0x08 : 0;
                    y8[ tt ] = (int8_t)( ( b > 0 ? b : 0 ) - ( q + e ) );
                    d |= b > 0 ? 0x10 : 0;
                    x28[ tt ] = (int8_t)( ( a2 > 0 ? a2 : 0 ) - ( q2 + e2 ) );
                    d |= a2 > 0 ? 0x20 : 0;
                    y28[ tt ] = (int8_t)( ( b2 > 0 ? b2 : 0 ) - ( q2 + e2 ) );
                    d |= b2 > 0 ? 0x40 : 0;
                }
                else
                {
                    x8[ tt ] = (int8_t)( ( 0 > a ? 0 : a ) - ( q + e ) );
                    d |= 0 > a ? 0 : 0x08;
                    y8[ tt ] = (int8_t)( ( 0 > b ? 0 : b ) - ( q + e ) );
                    d |= 0 > b ? 0 : 0x10;
                    x28[ tt ] = (int8_t)( ( 0 > a2 ? 0 : a2 ) - ( q2 + e2 ) );
                    d |= 0 > a2 ? 0 : 0x20;
                    y28[ tt ] = (int8_t)( ( 0 > b2 ? 0 : b2 ) - ( q2 + e2 ) );
                    d |= 0 > b2 ? 0 : 0x40;
                }
                pr[ tt ] = d;
            }
            __syncthreads( );
        }
        // calcMaxScore (kswcpp_core.h:156-299)
        i32 max_H, max_t;
        if( r > 0 )
        {
            const i32 en1 = st0 + ( ( en0 - st0 ) / HL ) * HL;
            // H[en0] from the OLD H[en0-1] (read before the row update below)
            TH hEn0 = (TH)( en0 > 0 ? H[ en0 - 1 ] + u8[ en0 ] : H[ en0 ] + v8[ en0 ] );
            __syncthreads( );
            // row update + per-lane running maximum; lane j always sees SIMD lane (j % HL)
            i32 bh = (i32)0x80000000, bc = 0x7fffffff; // (h, chunk index) ; chunk = (t-st0)/HL
            for( i32 k = lane; st0 + k < en0; k += 64 )
            {
                const i32 t = st0 + k;
                const TH h = (TH)( H[ t ] + (TH)v8[ t ] );
                H[ t ] = h;
                if( t < en1 )
                {
                    const i32 c = k / HL;
                    if( (i32)h > bh || ( (i32)h == bh && c < bc ) )
                        bh = (i32)h, bc = c;
                }
            }
            if( lane == 0 )
                H[ en0 ] = hEn0;
            // combine lanes with equal (lane % HL): xor over the bits above log2(HL)
            for( int m = HL; m < 64; m <<= 1 )
                ksw_red_pair( bh, bc, m );
            // lane j (< HL) now holds SIMD lane j: compare with the initial (H[en0], en0) which wins ties
            i32 vH = (i32)hEn0, vT = en0;
            if( bh > vH )
                vH = bh, vT = st0 + bc * HL;
            // independent horizontal maxima over the HL SIMD lanes
            i32 mh = vH, mt = vT;
            for( int m = 1; m < HL; m <<= 1 )
            {
                const i32 oh = __shfl_xor( mh, m, 64 ), ot = __shfl_xor( mt, m, 64 );
                mh = oh > mh ? oh : mh;
                mt = ot > mt ? ot : mt;
            }
            max_H = __shfl( mh, 0, 64 );
            max_t = __shfl( mt, 0, 64 );
            __syncthreads( );
            // scalar remainder [en1, en0) with true arg-max (kswcpp_core.h:238-243); H already updated
            for( i32 t = en1; t < en0; ++t )
            {
                const i32 h = (i32)H[ t ];
                if( h > max_H )
                    max_H = h, max_t = t;
            }
        }
        else
        {
            if( lane == 0 )
                H[ 0 ] = (TH)( v8[ 0 ] - qe );
            __syncthreads( );
            max_H = (i32)H[ 0 ];
            max_t = 0;
        }
        if( en0 == tlen - 1 && (i32)H[ en0 ] > ez.mte )
            ez.mte = (i32)H[ en0 ], ez.mte_q = (i32)( r - en );
        if( r - st0 == qlen - 1 && (i32)H[ st0 ] > ez.mqe )
            ez.mqe = (i32)H[ st0 ], ez.mqe_t = st0;
        // ksw_apply_zdrop (kswcpp_core.h:22-44), is_rot = 1
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
            ez.score = (i32)H[ tlen - 1 ];
        last_st = st;
        last_en = en;
        __syncthreads( );
    }
    __syncthreads( );
    // back-trace target (kswcpp_core.h:796-835)
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
    ksw_backtrack_lane0( M.p, M.cig, n_col, qlen, tlen, w, J.flag, i0, j0, nCigar, pathSteps, M.stage, M.stageBytes );
}
} // namespace ma
#endif
