// ksw_pk.h -- the exact kswcpp wavefront kernel (every ez field, every 16-aligned lane block of the SSE code with
// its overshoot lanes; contract and bit-exactness rules in ksw_wave.h) with TWO diagonal cells per lane:
// cell t lives in half (t & 1) of lane ((t >> 1) & 63) of ring slot ((t >> 7) mod R), a slot holds 128 cells.
// The int8 difference vectors u,v,x,y,x2,y2 and the score profile are packed as value << 8 (+ a tie-break tag in
// the low byte) in the two 16-bit halves of a VGPR, so one v_pk_* instruction advances two cells and ONE max chain
// yields both z and the direction state (ksw_ext.h explains the encoding).  The exact score H stays 32 bit (two
// registers per lane), so both the int16 and the int32 flavour of kswcpp (riskOfOverflow, kswcpp.h:101-115) run
// here.  Everything else -- ring rotation with the 16-aligned window start, carry-in, first-row initialisation,
// unaligned score-profile stride, calcMaxScore classes, z-drop, direction rows, back-trace -- follows ksw_wave.h.
// This is the kernel for wide bands (long reads: band 512, thousands of diagonals): per diagonal it issues roughly
// half the VALU instructions of its one-cell-per-lane predecessor (profiles/).
#pragma once
#include "ksw_ext.h"

#if defined( __HIPCC__ )
namespace ma
{
#if defined( MA_KSW_PROF )
// diagnostics build only: wave cycles of the phases of one diagonal of ksw_pk_core, summed over all jobs (tools/pk_prof.py)
static __device__ unsigned long long g_pk_prof[ 16 ];
#define PK_PROF_T( v ) const unsigned long long v = clock64( )
#define PK_PROF_ADD( i, a, b ) pkp[ i ] += ( b ) - ( a )
#else
#define PK_PROF_T( v )
#define PK_PROF_ADD( i, a, b )
#endif
// ring slots (128 cells each) a job needs here: the touched range of a diagonal is [st, st + m + 29]
MA_HD i32 ksw_pk_slots( i32 qlen, i32 tlen, i32 w )
{
    if( w < 0 )
        w = tlen > qlen ? tlen : qlen;
    i64 m = qlen < tlen ? qlen : tlen;
    if( (i64)w + 1 < m )
        m = (i64)w + 1;
    return (i32)( ( m + 30 + 127 ) / 128 );
}

__device__ __forceinline__ i32 pk_lo8( u32 x ) // int8 value of the low half
{
    return (i32)( x << 16 ) >> 24;
}
__device__ __forceinline__ i32 pk_hi8( u32 x ) // int8 value of the high half
{
    return (i32)x >> 24;
}

// The int16 / int32 flavour of H (riskOfOverflow) and the left / right aligned variant are run-time (wave-uniform)
// switches, not template parameters: four instantiations inlined into one launch kernel pushed the register
// allocator from ~110 to 248 VGPRs plus scratch.
template <int R, bool EARLY, typename QF, typename TF>
__device__ void ksw_pk_core( const KswScoring& SC, const KswJobView& J, QF qbase, TF tbase, uint8_t* qr /*LDS*/,
                             uint8_t* P /*HBM direction bytes*/, u32* cig, KswEz& ez, u32& nCigar, u64& cells,
                             u64& pathSteps, u32 ldsBytes, int2* snap /*LDS, 64 * R entries*/ )
{
    constexpr i32 RING = 128 * R;
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
    const bool h16 = ksw_h16( SC, qlen, tlen );
    const i32 HL = h16 ? 8 : 4, HLs = h16 ? 3 : 2; // lanes of one SSE register of H, log2
    const bool LEFT = !( J.flag & KSW_EZ_RIGHT );
    // int16 wrap-around of the reference's 16-bit H as a shift pair with a wave-uniform amount (a select between x and
    // its sign extension costs a v_bfe, a v_cndmask and the vcc shuffling around it)
    const i32 hSh = __builtin_amdgcn_readfirstlane( h16 ? 16 : 0 );
    auto TH = [ & ]( i32 x ) -> i32 { return (i32)( (u32)x << hSh ) >> hSh; };
    const i32 NEG = h16 ? -32768 : (i32)0x80000000;
    auto initOf = [ & ]( i32 r ) -> i32 {
        return (int8_t)( r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2 );
    };
    // tags (ksw_ext.h): LEFT keeps the first maximum of (s, a, b, a2, b2): d = 4 - tag; RIGHT the last of (s, a, b, a2)
    const u32 tS = LEFT ? 4 : 0, tX = LEFT ? 3 : 1, tY = 2, tX2 = LEFT ? 1 : 3, tY2 = 0;
#if defined( MA_PK_KV )
#define PK_K( x ) pk_opaque( x ) // experiment: the slot body's constants in VGPRs (for builds with fewer waves per SIMD)
#else
#define PK_K( x ) ( x )
#endif
    const u32 K_X0 = pk_val( -q - e, tX ), K_Y0 = pk_val( -q - e, tY ), K_X20 = pk_val( -q2 - e2, tX2 ),
              K_Y20 = pk_val( -q2 - e2, tY2 ), K_V0 = pk_val( -q - e, 0 ), K_S0 = pk_val( 0, tS );
    const u32 K_TX = PK_K( pk_val( 0, tX ) ), K_TY = pk_val( 0, tY ), K_TX2 = PK_K( pk_val( 0, tX2 ) ), K_TY2 = pk_val( 0, tY2 );
    const u32 K_Q = PK_K( pk_val( q, 0 ) ), K_Q2 = PK_K( pk_val( q2, 0 ) ), K_QE = PK_K( pk_val( q + e, 0 ) ), K_QE2 = PK_K( pk_val( q2 + e2, 0 ) );
    const u32 K_MCH = PK_K( pk_val( sc_mch, tS ) ), K_NDIFF = PK_K( pk_val( sc_mis - sc_mch, 0 ) ), K_NADJ = pk_val( -e2 - sc_mis, 0 );
    const u32 K_CLIP = PK_K( pk_val( sc_mch, 0xff ) );
    // lane 0 continues lane 63 of the previous ring slot: a bit mask for ONE v_bitop3 per neighbour view (written as
    // lane == 0 ? px[sp] : px[s] the compiler indexes the register array dynamically: 4 v_cndmask per view for R = 5)
    const u32 M_LANE0 = pk_opaque( lane == 0 ? 0xffffffffu : 0u );
    const u32 M_LEFT = pk_opaque( LEFT ? 0xffffffffu : 0u ), K_DX = pk_opaque( LEFT ? 0x00070007u : 0u ), K_DS = pk_opaque( LEFT ? 0x00030003u : 0u );
    const u32 K_FX = PK_K( pk_sub( K_TX, LEFT ? 0u : 0x00010001u ) ), K_FY = PK_K( pk_sub( K_TY, LEFT ? 0u : 0x00010001u ) ),
              K_FX2 = PK_K( pk_sub( K_TX2, LEFT ? 0u : 0x00010001u ) ), K_FY2 = PK_K( pk_sub( K_TY2, LEFT ? 0u : 0x00010001u ) );
    const u32 K_ONES = pk_opaque( 0x00010001u );

    // The reversed query in LDS.  A diagonal r reads the bases r - t of its cells, and the loop below ends after
    // ksw_max_diags diagonals at the latest, so only the first ksw_max_diags bases of the query -- the END of the reversed
    // array -- can ever be read: a 20 kb query against 1000 padded target bases (the end extension of a long read) keeps
    // 2.7 KB in LDS instead of 20.  Layout: head = qr[0, 64) (what the target view below can reach), tail = qr[qLo, qrBytes)
    // at LDS offset 64; qrT[i] addresses the tail with the original index.
    const i32 qLo = [ & ]( ) {
        const i32 lo = qlen - 1 - (i32)ksw_max_diags( qlen, tlen, J.w ) - 48;
        return lo > 64 ? ( lo & ~15 ) : 0;
    }( );
    const uint8_t* qrT = qLo ? qr + 64 - qLo : qr;
    // Bases >= 4 (N) anywhere in what this job can read switch the score profile to its general form; without them (the
    // pack holds no N, reads rarely do) match / mismatch is all there is: three instructions per slot instead of seven.
    u32 nAcc = 0;
    if( qLo == 0 )
        for( i32 t = lane; t < qrBytes; t += 64 )
        {
            const uint8_t b = t < qlen ? (uint8_t)qbase( qlen - 1 - t ) : (uint8_t)0;
            qr[ t ] = b;
            nAcc |= b;
        }
    else
    {
        {
            const uint8_t b = lane < qlen ? (uint8_t)qbase( qlen - 1 - lane ) : (uint8_t)0;
            qr[ lane ] = b;
            nAcc |= b;
        }
        for( i32 t = qLo + lane; t < qrBytes; t += 64 )
        {
            const uint8_t b = t < qlen ? (uint8_t)qbase( qlen - 1 - t ) : (uint8_t)0;
            qr[ 64 + t - qLo ] = b;
            nAcc |= b;
        }
    }
    __syncthreads( );

    // target byte as the reference's contiguous scratch sees it: sf[t] for t < L, then the qr region (a score profile
    // never reaches further than 15 bytes into it: pEnd <= en0 + 16)
    auto tgtAt = [ & ]( i32 tt ) -> u32 {
        if( tt < tlen )
            return (u32)tbase( tt ) & 0xffu;
        if( tt < L )
            return 0u;
        const i32 k = tt - L;
        if( k >= qrBytes )
            return 0u;
        return k < 64 || qLo == 0 ? (u32)qr[ k ] : ( k >= qLo ? (u32)qrT[ k ] : 0u );
    };

    u32 U[ R ], V[ R ], X[ R ], Y[ R ], X2[ R ], Y2[ R ], Sp[ R ], T[ R ];
    i32 Hlo[ R ], Hhi[ R ], TT[ R ];
    u32 PT[ R ]; // the two cell indices of the lane mod 2^16, one per half (range tests with packed arithmetic)
#pragma unroll
    for( int s = 0; s < R; s++ )
    {
        TT[ s ] = 128 * s + 2 * lane;
        PT[ s ] = (u32)( 128 * s + 2 * lane ) | (u32)( 128 * s + 2 * lane + 1 ) << 16;
        U[ s ] = V[ s ] = K_V0;
        X[ s ] = K_X0;
        Y[ s ] = K_Y0;
        X2[ s ] = K_X20;
        Y2[ s ] = K_Y20;
        Sp[ s ] = K_S0;
        T[ s ] = tgtAt( TT[ s ] ) | tgtAt( TT[ s ] + 1 ) << 16;
        nAcc |= T[ s ];
        Hlo[ s ] = Hhi[ s ] = NEG;
    }
    int hasN = __any( ( nAcc & 0x00fc00fcu ) != 0 ) ? 1 : 0; // wave-uniform; a recycled ring slot can still set it
    i32 last_st = -1, last_en = -1, cur_st = 0;
    // calcMaxScore (kswcpp_core.h:156-299) of one diagonal: the H of the cells of [st0, en0) as the lanes hold them (hl / hh: the
    // low / high cell of ring slot s, tts: the low cell's index), H[en0] = hE -> the SSE code's (max_H, max_t)
    auto exactMax = [ & ]( const i32( &hl )[ R ], const i32( &hh )[ R ], const i32( &tts )[ R ], i32 st0, i32 en0, i32 hE, i32& max_H, i32& max_t ) {
        const i32 en1 = st0 + ( ( ( en0 - st0 ) >> HLs ) << HLs );
        // classes (t - st0) mod HL: a lane's low cells all share one class, its high cells the next one
        i32 bhL = (i32)0x80000000, bkL = 0x7fffffff, bhH = (i32)0x80000000, bkH = 0x7fffffff;
        i32 tailL = (i32)0x80000000, tailH = (i32)0x80000000; // scalar remainder [en1, en0)
#pragma unroll
        for( int s = 0; s < R; s++ )
        {
            const i32 tt = tts[ s ];
            if( tt >= st0 && tt < en0 )
            {
                if( tt < en1 )
                    best_pair( bhL, bkL, hl[ s ], ( tt - st0 ) >> HLs );
                else
                    tailL = hl[ s ];
            }
            if( tt + 1 >= st0 && tt + 1 < en0 )
            {
                if( tt + 1 < en1 )
                    best_pair( bhH, bkH, hh[ s ], ( tt + 1 - st0 ) >> HLs );
                else
                    tailH = hh[ s ];
            }
        }
        // all-reduce over the lanes of one class: lanes with equal (lane mod HL/2)
        if( HL == 4 )
        {
            best_pair( bhL, bkL, dpp_ctrl<0x122>( bhL ), dpp_ctrl<0x122>( bkL ) ); // row_ror:2
            best_pair( bhH, bkH, dpp_ctrl<0x122>( bhH ), dpp_ctrl<0x122>( bkH ) );
        }
        best_pair( bhL, bkL, dpp_ctrl<0x124>( bhL ), dpp_ctrl<0x124>( bkL ) ); // row_ror:4
        best_pair( bhH, bkH, dpp_ctrl<0x124>( bhH ), dpp_ctrl<0x124>( bkH ) );
        best_pair( bhL, bkL, dpp_ctrl<0x128>( bhL ), dpp_ctrl<0x128>( bkL ) ); // row_ror:8
        best_pair( bhH, bkH, dpp_ctrl<0x128>( bhH ), dpp_ctrl<0x128>( bkH ) );
        best_pair( bhL, bkL, __shfl_xor( bhL, 16, 64 ), __shfl_xor( bkL, 16, 64 ) );
        best_pair( bhH, bkH, __shfl_xor( bhH, 16, 64 ), __shfl_xor( bkH, 16, 64 ) );
        best_pair( bhL, bkL, __shfl_xor( bhL, 32, 64 ), __shfl_xor( bkL, 32, 64 ) );
        best_pair( bhH, bkH, __shfl_xor( bhH, 32, 64 ), __shfl_xor( bkH, 32, 64 ) );
        // per class: the initial (H[en0], en0) wins ties; then independent horizontal maxima over the classes
        i32 mh = hE, mt = en0;
        if( bhL > hE )
            mh = bhL, mt = st0 + ( bkL << HLs );
        {
            const i32 vh = bhH > hE ? bhH : hE, vt = bhH > hE ? st0 + ( bkH << HLs ) : en0;
            mh = max( mh, vh );
            mt = max( mt, vt );
        }
        mh = max( mh, dpp_ctrl<0xB1>( mh ) ); // quad_perm [1,0,3,2]: the neighbouring lane's two classes
        mt = max( mt, dpp_ctrl<0xB1>( mt ) );
        if( HL == 8 )
        {
            mh = max( mh, dpp_ctrl<0x4E>( mh ) ); // quad_perm [2,3,0,1]
            mt = max( mt, dpp_ctrl<0x4E>( mt ) );
        }
        max_H = __builtin_amdgcn_readfirstlane( mh );
        max_t = __builtin_amdgcn_readfirstlane( mt );
        // scalar remainder [en1, en0): ascending t, strict >
        for( i32 t = en1; t < en0; ++t )
        {
            const i32 h = lane_bcast( ( t & 1 ) ? tailH : tailL, ( t >> 1 ) & 63 );
            if( h > max_H )
                max_H = h, max_t = t;
        }
    };
    // pending position of the last diagonal that raised ez.max (the lanes' H of that diagonal are in `snap`)
    bool pend = false;
    i32 pSt0 = 0, pEn0 = 0, pR = 0, pCur = 0, pHEn0 = 0;
    auto resolvePending = [ & ]( ) {
        if( !pend )
            return;
        i32 hl[ R ], hh[ R ], tts[ R ];
#pragma unroll
        for( int s = 0; s < R; s++ )
        {
            const int2 v = snap[ s * 64 + lane ];
            hl[ s ] = v.x, hh[ s ] = v.y;
            // the lane's cell of slot s inside the ring window [pCur, pCur + RING) of that diagonal
            i32 d = ( 128 * s + 2 * lane - pCur ) % RING;
            tts[ s ] = pCur + ( d < 0 ? d + RING : d );
        }
        i32 mh, mt;
        exactMax( hl, hh, tts, pSt0, pEn0, pHEn0, mh, mt );
        ez.max_t = mt;
        ez.max_q = pR - mt;
        pend = false;
    };
    i32 hBelow = NEG; // H[st-1]: the only recycled cell that is read again (as H[en0-1] when en0 == st)
    const i32 nDiag = qlen + tlen - 1;
    bool stop = false;
    // early stop of pipeline extensions (ksw_reg.h).  Bands that cut the rectangle (qlen > w + 1: the long end extensions
    // of long reads) are covered by the second part of the proof there; for them the bound is evaluated sparsely -- two
    // consecutive diagonals out of 16, and only once the band has reached the last target column (before that a cell on
    // the alignment's diagonal always has room left) -- because it costs about as much as 0.15 diagonals.
    const bool early = EARLY && ( J.flag & KSW_EZ_EXTZ_ONLY );
    // Global jobs of the pipeline (NeedlemanWunsch::ksw, needlemanWunsch.cpp:82-169: the gap fills between seeds): the caller
    // reads the cigar traced back from the corner and nothing else, so neither the running H of the cells nor any ez field is
    // needed -- as long as the band cannot run out before the corner (|tlen - qlen| + 2 <= w, checked against the bounds
    // below for all r; the caller's w is |tlen - qlen| + 10 at least) and no z-drop is asked for.  Per diagonal that is the
    // H arithmetic of every slot, the pick of H[en0] and the maximum tests: a quarter of the instructions of a one-slot job.
    const bool noH = EARLY && !( J.flag & KSW_EZ_EXTZ_ONLY ) && J.zdrop < 0 && ( tlen > qlen ? tlen - qlen : qlen - tlen ) + 2 <= w;
    const int noHU = __builtin_amdgcn_readfirstlane( noH ? 1 : 0 );
    const bool earlySparse = qlen > w + 1;
    const i32 earlyFrom = earlySparse ? w + 3 : qlen; // first diagonal at which B_r and B_{r-1} cover every in-band chain
    i32 topH = 0, boundPrev = 0x7fffffff; // H(r-1,-1) of the first-row boundary; B_{r-1} (ksw_reg.h)
    uint8_t* prow = P; // direction row of the current diagonal
#if defined( MA_KSW_PROF )
    unsigned long long pkp[ 12 ] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    for( i32 r = 0; r < nDiag && !stop; ++r )
    {
        PK_PROF_T( tp0 );
        // ---- bounds (kswcpp_core.h:541-559)
        i32 st0 = 0, en0 = tlen - 1;
        st0 = max( st0, r - qlen + 1 );
        en0 = min( en0, r );
        st0 = max( st0, ( r - w + 1 ) >> 1 );
        en0 = min( en0, ( r + w ) >> 1 );
        if( __builtin_expect( st0 > en0, 0 ) )
        {
            ez.zdropped = 1;
            break;
        }
        const i32 st = st0 & ~15, en = en0 | 15;
        // ---- carry-in (kswcpp_core.h:562-579) and ring rotation; st advances by 16 at most
        u32 x1 = K_X0 & 0xffffu, x21 = K_X20 & 0xffffu, v1 = K_V0 & 0xffffu; // one 16-bit half each
        if( st == 0 )
            v1 = ( (u32)initOf( r ) & 0xffu ) << 8;
        if( __builtin_expect( st != cur_st, 0 ) ) // every 32nd diagonal
        {
            const int jOld = ( cur_st >> 7 ) % R; // slot that holds [cur_st, cur_st + 16)
            const int src = ( ( st - 1 ) >> 1 ) & 63; // cell st-1 is the HIGH half of this lane
            const bool useOld = st - 1 >= last_st && st - 1 <= last_en;
#pragma unroll
            for( int s = 0; s < R; s++ )
                if( s == jOld )
                {
                    if( useOld )
                    {
                        x1 = (u32)lane_bcast( (i32)X[ s ], src ) >> 16;
                        x21 = (u32)lane_bcast( (i32)X2[ s ], src ) >> 16;
                        v1 = (u32)lane_bcast( (i32)V[ s ], src ) >> 16;
                    }
                    hBelow = lane_bcast( Hhi[ s ], src );
                    // recycle the 16 cells (8 lanes) that left the window for the 16 cells entering at the top
                    if( TT[ s ] < st )
                    {
                        TT[ s ] += RING;
                        PT[ s ] = pk_add( PT[ s ], pk_bcast( RING ) );
                        U[ s ] = V[ s ] = K_V0;
                        X[ s ] = K_X0;
                        Y[ s ] = K_Y0;
                        X2[ s ] = K_X20;
                        Y2[ s ] = K_Y20;
                        Sp[ s ] = K_S0;
                        Hlo[ s ] = Hhi[ s ] = NEG;
                        T[ s ] = tgtAt( TT[ s ] ) | tgtAt( TT[ s ] + 1 ) << 16;
                    }
                    if( !hasN && __any( ( T[ s ] & 0x00fc00fcu ) != 0 ) )
                        hasN = 1;
                }
            cur_st = st;
        }
        const i32 uInit = initOf( r );
        const bool initRow = en >= r; // kswcpp_core.h:580-585
        const int initRowU = __builtin_amdgcn_readfirstlane( initRow ? 1 : 0 ); // only the first ~w diagonals: a scalar branch
        const i32 pEnd = st0 + ( ( en0 - st0 ) / 16 + 1 ) * 16; // score profile refreshes [st0, pEnd)
        const u32 profSt = pk_bcast_s( st0 ), profLen = pk_bcast_s( pEnd - st0 );
        const i32 qoff = qlen - 1 - r;
        uint8_t* pr = prow - st; // P + r * n_col - st
        prow += n_col;
        cells += (u64)( en - st + 1 );
        const i32 hi = max( en, pEnd - 1 );
        const i32 en1 = st0 + ( ( ( en0 - st0 ) >> HLs ) << HLs );
        const u32 hLen = (u32)( en0 - st0 );
        const int b0 = st >> 7, j0 = __builtin_amdgcn_readfirstlane( b0 % R );
        const int nAct = __builtin_amdgcn_readfirstlane( ( hi >> 7 ) - b0 + 1 ); // ring slots from j0 on that hold touched cells
        // wave-uniform access to ONE cell of the ring (t in [st, st + RING)): slot, then a lane read
        auto slotOf = [ & ]( i32 t ) -> int {
            int sl = j0 + ( ( t >> 7 ) - b0 );
            return __builtin_amdgcn_readfirstlane( sl >= R ? sl - R : sl );
        };
        auto pickCell = [ & ]( const i32( &lo )[ R ], const i32( &hiH )[ R ], i32 t ) -> i32 {
            const int sl = slotOf( t ), ln = ( t >> 1 ) & 63;
            i32 v = 0;
#pragma unroll
            for( int s = 0; s < R; s++ )
                if( s == sl )
                    v = lane_bcast( ( t & 1 ) ? hiH[ s ] : lo[ s ], ln );
            return v;
        };
        auto pickByte = [ & ]( const u32( &A )[ R ], i32 t ) -> i32 { // int8 value of cell t of a packed difference vector
            const int sl = slotOf( t ), ln = ( t >> 1 ) & 63;
            u32 v = 0;
#pragma unroll
            for( int s = 0; s < R; s++ )
                if( s == sl )
                    v = (u32)lane_bcast( (i32)A[ s ], ln );
            return ( t & 1 ) ? pk_hi8( v ) : pk_lo8( v );
        };
        const int stLane = ( st >> 1 ) & 63; // cell st = low half of this lane of slot j0
        PK_PROF_T( tp1 );
        // previous-lane views (lane i <- lane i-1; lane 0 continues lane 63 of the previous slot of the ring)
        u32 px[ R ], pv[ R ], px2[ R ];
#pragma unroll
        for( int s = 0; s < R; s++ )
        {
            px[ s ] = lanes_ror1( X[ s ] );
            pv[ s ] = lanes_ror1( V[ s ] );
            px2[ s ] = lanes_ror1( X2[ s ] );
        }
        i32 laneMax = (i32)0x80000000; // largest new H of this lane's cells in [st0, en0)
        // query bases of all slots up front: the LDS latency (~100 cycles) is paid once per diagonal, not once per slot
        // (30 % of the wave cycles of the long-read launches were spent parked on these loads)
        u32 QB[ R ];
#pragma unroll
        for( int s = 0; s < R; s++ )
            QB[ s ] = (u32)qrT[ qoff + TT[ s ] ] | (u32)qrT[ qoff + TT[ s ] + 1 ] << 16;
        const int hasNU = __builtin_amdgcn_readfirstlane( hasN );
        __builtin_amdgcn_sched_barrier( 0 );
        PK_PROF_T( tp2 );
#pragma unroll
        for( int s = 0; s < R; s++ )
        {
            // wave-uniform skip of slots whose cells are all above the touched range
            if( ( s >= j0 ? s - j0 : s - j0 + R ) >= nAct )
                continue;
            const int sp = s == 0 ? R - 1 : s - 1;
            const i32 tt = TT[ s ]; // low cell; the high half is cell tt + 1
            // neighbours t-1 (values of the previous diagonal)
            u32 xt1 = cells_shift1( X[ s ], R == 1 ? px[ s ] : pk_bfi( M_LANE0, px[ sp ], px[ s ] ) );
            u32 vt1 = cells_shift1( V[ s ], R == 1 ? pv[ s ] : pk_bfi( M_LANE0, pv[ sp ], pv[ s ] ) );
            u32 x2t1 = cells_shift1( X2[ s ], R == 1 ? px2[ s ] : pk_bfi( M_LANE0, px2[ sp ], px2[ s ] ) );
            if( s == j0 )
            {
                // cell st takes the carry-in.  A real scalar branch: if-converted (the compiler's choice for three selects) EVERY
                // slot pays three v_bitop3 and three v_cndmask for the one slot that has the cell
                asm volatile( "" );
                const u32 m = lane == stLane ? 0x0000ffffu : 0u;
                xt1 = pk_bfi( m, x1, xt1 );
                vt1 = pk_bfi( m, v1, vt1 );
                x2t1 = pk_bfi( m, x21, x2t1 );
            }
            // first row / column initialisation of cell r
            if( __builtin_expect( initRowU != 0, 0 ) )
            {
                // cell r = half (r - tt) of the lane with tt <= r <= tt + 1
                const u32 m = pk_opaque( tt == r ? 0x0000ffffu : ( tt + 1 == r ? 0xffff0000u : 0u ) );
                Y[ s ] = pk_bfi( m, K_Y0, Y[ s ] );
                Y2[ s ] = pk_bfi( m, K_Y20, Y2[ s ] );
                U[ s ] = pk_bfi( m, pk_val( uInit, 0 ), U[ s ] );
            }
            // score profile of the cells in [st0, pEnd): match / mismatch, -e2 when either base is N
            {
                const u32 b = QB[ s ];
                u32 val;
                if( __builtin_expect( hasNU != 0, 0 ) )
                {
                    const u32 isN = pk_lshr( T[ s ] | b, 2 );
                    const u32 differ = pk_min1( ( T[ s ] ^ b ) | isN, K_ONES );
                    val = pk_mad( differ, K_NDIFF, K_MCH );
                    val = pk_mad( isN, K_NADJ, val );
                }
                else
                    val = pk_mad( pk_min1( T[ s ] ^ b, K_ONES ), K_NDIFF, K_MCH );
                // cells in [st0, pEnd): (t - st0) mod 2^16 < pEnd - st0, per half (the window is < 2^15 cells wide)
                const u32 inProf = pk_nonzero15( pk_subsatu( profLen, pk_sub( PT[ s ], profSt ) ) );
                Sp[ s ] = pk_bfi( inProf, val, Sp[ s ] );
            }
            // DP cell (kswcpp_core.h:653-766)
            u32 nu, nv;
            {
                u32 z = Sp[ s ];
                const u32 ut = U[ s ];
                u32 a = pk_add( xt1, vt1 );
                u32 b = pk_add( Y[ s ], ut );
                u32 a2 = pk_add( x2t1, vt1 );
                u32 b2 = pk_add( Y2[ s ], ut );
                // LEFT: d = 4 - tag of max(s, a, b, a2, b2); RIGHT: d = tag of max(s, a, b, a2), state 4 is never recorded
                // (kswcpp_core.h:693-699).  One instruction stream for both (the two variants as run-time branches cost ~20
                // scalar instructions and four branches per slot): 4 - t = (t ^ 7) - 3 for a tag t in 0..7.
                const u32 z4 = pk_max( pk_max( z, a ), pk_max( b, a2 ) );
                z = pk_max( z4, b2 );
                const u32 d = pk_sub( ( pk_bfi( M_LEFT, z, z4 ) & 0x00070007u ) ^ K_DX, K_DS );
                const u32 zc = pk_min( z, K_CLIP ) & 0xff00ff00u;
                nu = pk_sub( zc, vt1 );
                nv = pk_sub( zc, ut );
                u32 tmp = pk_sub( zc, K_Q );
                a = pk_sub( a, tmp );
                b = pk_sub( b, tmp );
                tmp = pk_sub( zc, K_Q2 );
                a2 = pk_sub( a2, tmp );
                b2 = pk_sub( b2, tmp );
                const u32 nx = pk_sub( pk_max( a, K_TX ), K_QE ), ny = pk_sub( pk_max( b, K_TY ), K_QE );
                const u32 nx2 = pk_sub( pk_max( a2, K_TX2 ), K_QE2 ), ny2 = pk_sub( pk_max( b2, K_TY2 ), K_QE2 );
                // bit 15 of a half = continuation flag: LEFT a > 0, RIGHT !(a < 0) = a > -1 (the thresholds K_F*)
                const u32 fa = pk_subsat( K_FX, a ), fb = pk_subsat( K_FY, b ), fa2 = pk_subsat( K_FX2, a2 ), fb2 = pk_subsat( K_FY2, b2 );
                const u32 dd = and_or( fa >> 12, 0x00080008u, and_or( fb >> 11, 0x00100010u, and_or( fa2 >> 10, 0x00200020u, and_or( fb2 >> 9, 0x00400040u, d ) ) ) );
                if( tt >= st && tt <= en ) // st is even, en odd: a lane is inside with both cells or not at all
                {
                    U[ s ] = nu;
                    V[ s ] = nv;
                    X[ s ] = nx;
                    Y[ s ] = ny;
                    X2[ s ] = nx2;
                    Y2[ s ] = ny2;
                    *(uint16_t*)( pr + tt ) = (uint16_t)__builtin_amdgcn_perm( 0u, dd, 0x0c0c0200u );
                }
            }
            // ---- calcMaxScore pieces (kswcpp_core.h:156-299) with this diagonal's u / v
            // (cells of [st0, en0] lie inside [st, en], so nu / nv are the committed values there).  Only the running H
            // of the cells in [st0, en0) is advanced here; H[en0] and H[st0] belong to ONE lane of ONE slot each and
            // are picked out of the registers after the loop with wave-uniform lane reads.
            if( r > 0 && !noHU )
            {
                // cells of [st0, en0): one unsigned compare per half, (t - st0) < en0 - st0
                const i32 vlo = pk_lo8( nv ), vhi = pk_hi8( nv );
                const u32 d0 = (u32)( tt - st0 );
                const bool inLo = d0 < hLen, inHi = d0 + 1u < hLen;
                const i32 nlo = TH( Hlo[ s ] + vlo ), nhi = TH( Hhi[ s ] + vhi );
                Hlo[ s ] = inLo ? nlo : Hlo[ s ];
                Hhi[ s ] = inHi ? nhi : Hhi[ s ];
                laneMax = max( laneMax, max( inLo ? nlo : (i32)0x80000000, inHi ? nhi : (i32)0x80000000 ) );
            }
            // keep the slots' instruction streams apart: interleaving them multiplies the live temporaries by R
            __builtin_amdgcn_sched_barrier( 0 );
        }
        PK_PROF_T( tp3 );
#if defined( MA_KSW_PROF )
        pkp[ 7 ] += (unsigned long long)nAct;
        unsigned long long tp4 = tp3;
#endif
        i32 max_H = (i32)0x80000000, max_t = 0, hEnd = 0, hS = 0;
        bool raised = false;
        if( noHU )
            ;
        else if( r > 0 )
        {
            // H[en0] = en0 > 0 ? Hold[en0-1] + u[en0] : Hold[en0] + v[en0].  Cell en0 - 1 was advanced above when it
            // lies in [st0, en0): its old value is the new one minus this diagonal's v (16-bit wrap-around is a ring
            // homomorphism, so TH commutes); below st it has left the ring and is the carry-in hBelow.
            i32 hEn0;
            const int sE = slotOf( en0 ), lE = ( en0 >> 1 ) & 63;
            if( __builtin_expect( ( en0 & 127 ) != 0 && en0 - 1 >= st, 1 ) )
            {
                // the usual case: cells en0 - 1 and en0 sit in the same ring slot -- one pass over the slots reads the three
                // registers involved and puts H[en0] in place (each separate pick is an R-way chain of scalar branches)
                const i32 c = en0 - 1;
                const int lC = ( c >> 1 ) & 63;
                hEn0 = 0;
#pragma unroll
                for( int s = 0; s < R; s++ )
                    if( s == sE )
                    {
                        i32 hOld = lane_bcast( ( c & 1 ) ? Hhi[ s ] : Hlo[ s ], lC );
                        if( c >= st0 )
                        {
                            const u32 vC = (u32)lane_bcast( (i32)V[ s ], lC );
                            hOld -= ( c & 1 ) ? pk_hi8( vC ) : pk_lo8( vC );
                        }
                        const u32 uE = (u32)lane_bcast( (i32)U[ s ], lE );
                        hEn0 = TH( hOld + ( ( en0 & 1 ) ? pk_hi8( uE ) : pk_lo8( uE ) ) );
                        if( en0 & 1 )
                            Hhi[ s ] = lane == lE ? hEn0 : Hhi[ s ];
                        else
                            Hlo[ s ] = lane == lE ? hEn0 : Hlo[ s ];
                    }
            }
            else
            {
                if( en0 > 0 )
                {
                    const i32 c = en0 - 1;
                    i32 hOld = hBelow;
                    if( c >= st )
                    {
                        hOld = pickCell( Hlo, Hhi, c );
                        if( c >= st0 )
                            hOld -= pickByte( V, c );
                    }
                    hEn0 = TH( hOld + pickByte( U, en0 ) );
                }
                else
                    hEn0 = TH( pickCell( Hlo, Hhi, 0 ) + pickByte( V, 0 ) );
#pragma unroll
                for( int s = 0; s < R; s++ )
                    if( s == sE )
                    {
                        if( en0 & 1 )
                            Hhi[ s ] = lane == lE ? hEn0 : Hhi[ s ];
                        else
                            Hlo[ s ] = lane == lE ? hEn0 : Hlo[ s ];
                    }
            }
            hEnd = hEn0;
#if defined( MA_KSW_PROF )
            tp4 = clock64( );
#endif
            hS = r - st0 != qlen - 1 ? 0 : st0 == en0 ? hEn0 : pickCell( Hlo, Hhi, st0 ); // only read on the last query row (mqe below)
            // The exact (max_H, max_t) -- the class-wise first maxima of the SSE code, ~240 instructions -- is only consumed when
            // the diagonal raises ez.max or could z-drop (ksw_reg.h).  A diagonal that RAISES ez.max (every other one while
            // an alignment runs: 17 % of the 10 kb DP stage) needs only the value right away, max( H[en0], the lanes' maxima );
            // its position max_t is read by the z-drop test and by the caller at the end, and only that of the LAST raise:
            // the lanes' H go to LDS (one ds_write_b64 per ring slot) and the position is worked out of that snapshot when
            // somebody asks (resolvePending).
            max_H = (i32)0x80000000;
            max_t = 0;
            raised = hEn0 > (i32)ez.max || __any( laneMax > (i32)ez.max ) != 0;
            bool need = false;
            if( raised )
            {
                // the new ez.max = the largest H of the diagonal.  While an alignment runs, ONE lane is above the old maximum
                // (the cell on the alignment's diagonal): its value is a lane read; the reduction over the lanes (12 DPP steps)
                // only runs when several are
                const unsigned long long above = __ballot( laneMax > (i32)ez.max );
                i32 m = hEn0;
                if( above != 0 )
                {
                    i32 mm;
                    if( ( above & ( above - 1 ) ) == 0 )
                        mm = __builtin_amdgcn_readlane( laneMax, (int)__builtin_ctzll( above ) );
                    else
                    {
                        asm volatile( "" ); // keeps the reduction on its own side of the branch
                        mm = wave_max_i32( laneMax );
                    }
                    m = max( m, mm );
                }
#pragma unroll
                for( int s = 0; s < R; s++ )
                    snap[ s * 64 + lane ] = make_int2( Hlo[ s ], Hhi[ s ] );
                pend = true, pSt0 = st0, pEn0 = en0, pR = r, pCur = cur_st, pHEn0 = hEn0;
                ez.max = (u32)m & 0x7fffffffu;
            }
            else if( J.zdrop >= 0 )
                need = hEn0 < (i32)ez.max - J.zdrop && __any( laneMax >= (i32)ez.max - J.zdrop ) == 0;
            if( __builtin_expect( need, 0 ) )
            {
                resolvePending( );
                exactMax( Hlo, Hhi, TT, st0, en0, hEn0, max_H, max_t );
            }
#if defined( MA_KSW_PROF )
            pkp[ 8 ] += raised ? 1 : 0;
            pkp[ 9 ] += need ? 1 : 0;
#endif
        }
        else
        {
            // r == 0: H[0] = v[0] - (q+e) (kswcpp_core.h:244-249); cell 0 = low half of lane 0 of slot 0
            const i32 h0 = TH( pk_lo8( (u32)lane_bcast( (i32)V[ 0 ], 0 ) ) - qe0 );
            if( lane == 0 )
                Hlo[ 0 ] = h0;
            max_H = h0;
            max_t = 0;
            hEnd = h0;
            hS = h0;
        }
        PK_PROF_T( tp5 );
        if( noHU )
            ;
        else if( en0 == tlen - 1 && hEnd > ez.mte )
            ez.mte = hEnd, ez.mte_q = r - en;
        if( !noHU && r - st0 == qlen - 1 && hS > ez.mqe )
            ez.mqe = hS, ez.mqe_t = st0;
        // ksw_apply_zdrop (kswcpp_core.h:22-44), is_rot = 1
        if( raised || noHU )
            ; // ez.max is up to date, (max_t, max_q) pending
        else if( max_H > (i32)ez.max )
        {
            ez.max = (u32)max_H & 0x7fffffffu; // r == 0
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
        if( !noHU && !stop && r == qlen + tlen - 2 && en0 == tlen - 1 )
            ez.score = hEnd;
        if( EARLY && __builtin_expect( early && r >= earlyFrom - 1 && ( earlySparse ? en0 == tlen - 1 && ( r & 15 ) <= 1 : ( r & 7 ) <= 1 ), 0 ) )
        {
            // early stop of pipeline extensions (proof in ksw_reg.h)
            i32 bnd = (i32)0x80000000;
#pragma unroll
            for( int s = 0; s < R; s++ )
            {
                const i32 tt = TT[ s ];
                if( tt >= st0 && tt <= en0 )
                    bnd = max( bnd, Hlo[ s ] + sc_mch * min( qoff + tt, tlen - 1 - tt ) );
                if( tt + 1 >= st0 && tt + 1 <= en0 )
                    bnd = max( bnd, Hhi[ s ] + sc_mch * min( qoff + tt + 1, tlen - 2 - tt ) );
            }
            bnd = wave_max_i32( bnd );
            if( !earlySparse )
            {
                // two consecutive diagonals out of 8 (the bound costs ~0.15 diagonals; these jobs run ~2 * qlen >= 500 of them)
                if( ( r & 7 ) == 1 && r >= qlen && max( max( bnd, boundPrev ), topH + sc_mch * qlen ) <= (i32)ez.max )
                    stop = true;
            }
            else if( ( r & 15 ) == 1 && r >= earlyFrom && max( bnd, boundPrev ) <= (i32)ez.max )
                stop = true; // boundPrev = B_{r-1} of the diagonal before (r & 15 == 0); no first-row chain starts after diagonal w + 1
            boundPrev = bnd;
        }
#if defined( MA_EXP_SALU ) // experiment (tools/dp_bound_experiment.sh): which issue port bounds the loop?
        {
            u32 t0 = (u32)__builtin_amdgcn_readfirstlane( r );
            asm volatile( "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                          "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                          : "+s"( t0 ) );
            if( t0 == 0xdeadbeefu )
                cells++;
        }
#endif
#if defined( MA_EXP_VALU )
        {
            u32 t0 = U[ 0 ];
            asm volatile( "v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n"
                          "v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n"
                          : "+v"( t0 ) );
            if( t0 == 0xdeadbeefu )
                cells++;
        }
#endif
        topH += uInit; // H(r,-1)
        last_st = st;
        last_en = en;
        PK_PROF_T( tp6 );
        PK_PROF_ADD( 0, tp0, tp1 );
        PK_PROF_ADD( 1, tp1, tp2 );
        PK_PROF_ADD( 2, tp2, tp3 );
        PK_PROF_ADD( 3, tp3, tp4 );
        PK_PROF_ADD( 4, tp4, tp5 );
        PK_PROF_ADD( 5, tp5, tp6 );
        PK_PROF_ADD( 6, 0ull, 1ull );
    }
#if defined( MA_KSW_PROF )
    if( lane == 0 && R == 5 )
    {
        for( int i = 0; i < 10; i++ )
            atomicAdd( &g_pk_prof[ i ], pkp[ i ] );
        atomicAdd( &g_pk_prof[ 10 ], 1ull );
    }
    const unsigned long long tpE = clock64( );
#endif
    resolvePending( );
    __syncthreads( ); // direction bytes of all lanes visible to the back-trace
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
#if defined( MA_KSW_PROF )
    if( lane == 0 && R == 5 )
        atomicAdd( &g_pk_prof[ 11 ], clock64( ) - tpE );
#endif
}
} // namespace ma
#endif
