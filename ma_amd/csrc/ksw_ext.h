// ksw_ext.h -- the pipeline's extension DP (NeedlemanWunsch::dynPrg with bLocalBeginning / bLocalEnd and
// ksw_dual_ext, needlemanWunsch.cpp:239-622): kswcpp_dispatch with KSW_EZ_EXTZ_ONLY, of which the callers read
// only ez.max_q, ez.max_t and the cigar traced back from that cell.
//
// Regime.  With qlen <= w+1 and r <= w the band never cuts the DP rectangle: on diagonal r the reference's
// [st0, en0] is [max(0, r-qlen+1), min(r, tlen-1)], every cell of it is a true cell of the rectangle, and the
// cells the 16-lane blocks compute outside of it (the aligned overshoot) are never read by a true cell: the
// lower neighbour of cell st0 was cell st0-1 of the previous diagonal, a first-row cell is initialised before it
// is used (kswcpp_core.h:580-585), and the back-trace cannot leave the rectangle.  So only true cells are
// computed here.  A job that gets to r > w without having stopped is handed back (return false) and re-run by
// the exact kernel (ksw_pk.h); with the early stop below that does not happen for short reads.
//
// Layout.  One wavefront per job, TWO cells per lane: cell t lives in half (t & 1) of lane ((t mod RING) >> 1)
// of register slot ((t mod RING) >> 7), RING = 128 * R cells.  The int8 difference vectors are kept as
// value << 8 in a 16-bit half, so packed 16-bit adds / subs wrap exactly like the reference's epi8 arithmetic
// and v_pk_max/min_i16 order them correctly; the low byte of a half carries a small tag that makes ONE max
// chain deliver both z and the direction state d (first maximum wins for the left-aligned variant, last for the
// right-aligned one, kswcpp_core.h:653-699).  The query flows through the lanes (one cell per diagonal), the
// target base, the exact score H (int16: riskOfOverflow<int16_t> must hold) and the cell index stay put; cells
// that fell out of the band are recycled 16 at a time for cells RING further up.  H is tracked as
// H(t-1, r-1) + u(t, r), which equals the reference's H(t, r-1) + v(t, r) wherever both exist.
// Per diagonal the wave needs max H; the reference's max_t (8-lane classes with independent horizontal maxima,
// kswcpp_core.h:156-299) is only evaluated on diagonals that raise ez.max or could z-drop.
// Early stop: see ksw_reg.h (same bound, same proof; here the regime makes every cell exact DP).
#pragma once
#include "ksw_wave.h"
#include "ksw_reg.h"

#if defined( __HIPCC__ )
namespace ma
{
typedef short ksw_s2 __attribute__( ( ext_vector_type( 2 ) ) );
typedef unsigned short ksw_u2 __attribute__( ( ext_vector_type( 2 ) ) );
#define KS2( x ) __builtin_bit_cast( ksw_s2, (u32)( x ) )
#define KU2( x ) __builtin_bit_cast( ksw_u2, (u32)( x ) )
#define KR( x ) __builtin_bit_cast( u32, ( x ) )
__device__ __forceinline__ u32 pk_add( u32 a, u32 b ) { return KR( KU2( a ) + KU2( b ) ); }
__device__ __forceinline__ u32 pk_sub( u32 a, u32 b ) { return KR( KU2( a ) - KU2( b ) ); }
__device__ __forceinline__ u32 pk_max( u32 a, u32 b ) { return KR( __builtin_elementwise_max( KS2( a ), KS2( b ) ) ); }
__device__ __forceinline__ u32 pk_min( u32 a, u32 b ) { return KR( __builtin_elementwise_min( KS2( a ), KS2( b ) ) ); }
__device__ __forceinline__ u32 pk_minu( u32 a, u32 b ) { return KR( __builtin_elementwise_min( KU2( a ), KU2( b ) ) ); }
// min(x, 1) per half as ONE instruction: written with the builtin the compiler turns it into x != 0 ? 1 : 0 and a
// multiply by it into per-half compares + v_cndmask + v_perm (8 instructions in the score profile of ksw_pk.h)
__device__ __forceinline__ u32 pk_min1( u32 x, u32 ones /* 0x00010001 in a register */ )
{
    u32 r;
    asm( "v_pk_min_u16 %0, %1, %2" : "=v"( r ) : "v"( x ), "v"( ones ) );
    return r;
}
__device__ __forceinline__ u32 pk_subsat( u32 a, u32 b ) { return KR( __builtin_elementwise_sub_sat( KS2( a ), KS2( b ) ) ); }
__device__ __forceinline__ u32 pk_subsatu( u32 a, u32 b ) { return KR( __builtin_elementwise_sub_sat( KU2( a ), KU2( b ) ) ); }
__device__ __forceinline__ u32 pk_ashr8( u32 a ) { return KR( KS2( a ) >> (short)8 ); }
__device__ __forceinline__ u32 pk_lshr( u32 a, int n ) { return KR( KU2( a ) >> (unsigned short)n ); }
__device__ __forceinline__ u32 pk_lshr15( u32 a ) { return KR( KU2( a ) >> (unsigned short)15 ); }
__device__ __forceinline__ u32 pk_mad( u32 a, u32 b, u32 c ) { return KR( KU2( a ) * KU2( b ) + KU2( c ) ); }
__device__ __forceinline__ u32 pk_bfi( u32 mask, u32 a, u32 b ) // mask ? a : b, bitwise
{
    // ONE v_bitop3_b32 (truth table 0xCA = multiplexer).  Written as b ^ ((a ^ b) & mask) the compiler emits v_xor + v_bitop3;
    // v_bitop3_b32 issues at the full VALU rate (2.2 cycles per wave64 instruction, v_bfi_b32 / v_and_or_b32 at 4.1:
    // profiles/r02_valu_mix.txt)
    return __builtin_amdgcn_bitop3_b32( mask, a, b, 0xCA );
}
__device__ __forceinline__ u32 and_or( u32 a, u32 m, u32 c ) // (a & m) | c as one full-rate v_bitop3_b32 (not v_and_or_b32)
{
    return __builtin_amdgcn_bitop3_b32( a, m, c, 0xEA );
}
// Opaque to the optimiser: keeps a per-half 0 / 0xffff mask a plain 32-bit value, so that pk_bfi stays ONE v_bfi_b32
// instead of being rewritten into per-half compares + v_cndmask + v_perm (4-5 instructions per select).
__device__ __forceinline__ u32 pk_opaque( u32 x )
{
    asm( "" : "+v"( x ) );
    return x;
}
__device__ __forceinline__ u32 pk_nonzero15( u32 x ) // per half: 0xffff if 1 <= x <= 0x8000 else 0 (x = 0)
{
    return KR( KS2( pk_opaque( KR( KU2( x ) + KU2( 0x7fff7fffu ) ) ) ) >> (short)15 );
}
__device__ __forceinline__ u32 pk_bcast( i32 v ) // both halves = v (16 bit)
{
    return ( (u32)v & 0xffffu ) * 0x00010001u;
}
__device__ __forceinline__ u32 pk_bcast_s( i32 v ) // the same for a wave-uniform value: one scalar instruction
{
    u32 r;
    asm( "s_pack_ll_b32_b16 %0, %1, %1" : "=s"( r ) : "s"( __builtin_amdgcn_readfirstlane( v ) ) );
    return r;
}
__device__ __forceinline__ u32 pk_val( i32 v8, u32 tag ) // int8 value in the high byte, tag in the low byte
{
    return pk_bcast( (i32)( ( ( (u32)v8 & 0xffu ) << 8 ) | tag ) );
}
// x with lane 0 replaced by the wave-uniform value v (v_writelane_b32; the lane select is the constant 0)
__device__ __forceinline__ u32 lane0_write( u32 x, u32 v )
{
    asm( "v_writelane_b32 %0, %1, 0" : "+v"( x ) : "s"( __builtin_amdgcn_readfirstlane( (i32)v ) ) );
    return x;
}
// lane i <- lane i-1 with lane 0 <- lane 63 (one register = a ring of 64 lanes)
__device__ __forceinline__ u32 lanes_ror1( u32 x ) { return (u32)dpp_wave_ror1( (i32)x ); }
// per-cell shift by one: half lo <- previous lane's hi, half hi <- own lo
__device__ __forceinline__ u32 cells_shift1( u32 cur, u32 prevLanes ) { return __builtin_amdgcn_alignbit( cur, prevLanes, 16 ); }

// registers slots a job needs in this kernel, 0 = not eligible (see the regime above)
MA_HD int ksw_ext_slots( const KswScoring& SC, i32 qlen, i32 tlen, i32 w, i32 zdrop, i32 flag )
{
    if( qlen < 1 || tlen < 1 || w < 0 || qlen > w + 1 || !ksw_h16( SC, qlen, tlen ) )
        return 0;
    // global jobs (NeedlemanWunsch::ksw, needlemanWunsch.cpp:82-169: only the cigar is read): all diagonals must be
    // inside the regime and nothing may depend on the running maximum
    if( !( flag & 0x40 ) && ( zdrop >= 0 || (i64)qlen + tlen - 2 > (i64)w ) )
        return 0;
    // the difference vectors must stay inside int8 (they do in kswcpp for such scores; here H tracking relies on it)
    const i32 a = SC.q + SC.e, b = SC.q2 + SC.e2, mch = SC.match < 0 ? -SC.match : SC.match;
    const i32 mis = SC.mismatch < 0 ? -SC.mismatch : SC.mismatch;
    if( SC.q < 0 || SC.e < 1 || SC.q2 < 0 || SC.e2 < 1 || 2 * ( a > b ? a : b ) + mch + mis > 120 )
        return 0;
    if( qlen > 256 )
        return 0; // the query is held four bases per lane
    // one slot of 128 cells: the live window (<= qlen cells) plus the cells that left the band and are not yet handed
    // on; those are handed on 16 at a time for qlen <= 113 and lane by lane (2 cells) for qlen up to 126
    if( qlen + 2 <= 128 || tlen <= 128 )
        return 1;
    if( qlen + 2 <= 256 || tlen <= 256 )
        return 2;
    return 0;
}

// GLOBAL: no KSW_EZ_EXTZ_ONLY, zdrop < 0: every diagonal is computed, no score is tracked, the back-trace starts
// at (tlen-1, qlen-1) (kswcpp_core.h:796-835) and ez keeps its initial values apart from the cigar.
// ksw_backtrack__ (kswcpp_core.h:76-150) over ring rows (RING bytes per diagonal, cell (r, i) at column i mod RING).
// Inside the regime the path never leaves the DP rectangle, so the force_state cases cannot occur.  Everything is
// wave-uniform and kept on the scalar unit; rows are staged into LDS `rows` at a time by all 64 lanes.
template <int RING>
__device__ __forceinline__ void ksw_backtrack_ring( const uint8_t* P, u32* cig, i32 flag, i32 i0, i32 j0, u32& nCigar,
                                                    u64& pathSteps, uint8_t* stage, u32 stageBytes, i32 qlen, i32 tlen )
{
    const int lane = threadIdx.x & 63;
    const i32 rows = (i32)( stageBytes / RING ); // >= 1
    i32 rlo = 1 << 30;
    u32 n = 0, curOp = 3, curLen = 0, steps = 0;
    i32 i = __builtin_amdgcn_readfirstlane( i0 ), j = __builtin_amdgcn_readfirstlane( j0 ), state = 0;
    while( i >= 0 && j >= 0 )
    {
        const i32 r = i + j;
        if( r < rlo )
        {
            __syncthreads( );
            rlo = r - rows + 1 > 0 ? r - rows + 1 : 0;
            const i32 bytes = ( r - rlo + 1 ) * RING;
            const uint8_t* src = P + (size_t)rlo * RING;
            // only the 16-byte chunks that hold live cells [st0, en0] of their row were written; read no others
            for( i32 k = lane * 16; k < bytes; k += 1024 )
            {
                const i32 rr = rlo + k / RING, col = k & ( RING - 1 );
                const i32 st = ( rr - qlen + 1 > 0 ? rr - qlen + 1 : 0 ) & ~15, en = rr < tlen - 1 ? rr : tlen - 1;
                if( ( ( col - st ) & ( RING - 1 ) ) <= en - st )
                    *(uint4*)( stage + k ) = *(const uint4*)( src + k );
            }
            __syncthreads( );
        }
        const i32 tmp = __builtin_amdgcn_readfirstlane( (i32)stage[ ( r - rlo ) * RING + ( i & ( RING - 1 ) ) ] );
        if( state != 0 && !( ( tmp >> ( state + 2 ) ) & 1 ) )
            state = 0;
        if( state == 0 )
            state = tmp & 7;
        const u32 op = state == 0 ? 0u : ( ( state == 1 || state == 3 ) ? 2u : 1u );
        steps++;
        if( op == curOp )
            curLen++;
        else
        {
            if( curLen )
                cig[ n++ ] = curLen << 4 | curOp; // uniform store
            curOp = op;
            curLen = 1;
        }
        i -= op != 1u ? 1 : 0;
        j -= op != 2u ? 1 : 0;
    }
    // leftovers: deletions for target, insertions for query (kswcpp_core.h:139-144)
    auto push = [ & ]( u32 op, u32 len ) {
        if( op == curOp )
            curLen += len;
        else
        {
            if( curLen )
                cig[ n++ ] = curLen << 4 | curOp;
            curOp = op;
            curLen = len;
        }
    };
    if( i >= 0 )
        push( 2, (u32)( i + 1 ) );
    if( j >= 0 )
        push( 1, (u32)( j + 1 ) );
    if( curLen )
        cig[ n++ ] = curLen << 4 | curOp;
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

template <int R, bool LEFT, bool GLOBAL, typename QF, typename TF>
__device__ bool ksw_ext_core( const KswScoring& SC, const KswJobView& J, QF qbase, TF tbase, uint8_t* lds, u32 ldsBytes,
                              uint8_t* P /*HBM direction rows, RING bytes each*/, u32* cig, KswEz& ez, u32& nCigar,
                              u64& cells, u64& pathSteps, uint2* snap /*LDS, 64 * R entries*/
#if defined( MA_KSW_PROF )
                              ,
                              unsigned long long* prof
#endif
)
{
    constexpr i32 RING = 128 * R;
    const int lane = threadIdx.x & 63;
#if defined( MA_KSW_PROF )
    const unsigned long long tpA = clock64( );
#endif
    const i32 qlen = J.qlen, tlen = J.tlen, w = J.w;
    ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
    ez.max = 0;
    ez.score = ez.mqe = ez.mte = (i32)0x80000000;
    ez.zdropped = 0;
    ez.reach_end = 0;
    nCigar = 0;
    int8_t q = (int8_t)SC.q, e = (int8_t)SC.e, q2 = (int8_t)SC.q2, e2 = (int8_t)SC.e2;
    const i32 sc_mch = (int8_t)( SC.match < 0 ? -SC.match : SC.match );
    const i32 sc_mis = (int8_t)( SC.mismatch > 0 ? -SC.mismatch : SC.mismatch );
    const i32 qe0 = q + e; // q+e before the swap: kswcpp seeds H[0] with it (kswcpp_core.h:244-249)
    if( q2 + e2 < q + e )
    {
        int8_t t = q;
        q = q2;
        q2 = t;
        t = e;
        e = e2;
        e2 = t;
    }
    {
        const i32 min_sc = sc_mis < 0 ? sc_mis : 0;
        if( -min_sc > 2 * ( q + e ) )
            return true; // kswcpp returns an untouched ez (kswcpp_core.h:340-341)
    }
    i32 long_thres = e != e2 ? ( q2 - q ) / ( e - e2 ) - 1 : 0;
    if( q2 + e2 + long_thres * e2 > q + e + long_thres * e )
        ++long_thres;
    const i32 long_diff = long_thres * ( e - e2 ) - ( q2 - q ) - e2;
    // first-row / first-column boundary differences (kswcpp_core.h:562-585): value for diagonal / cell r
    auto initOf = [ & ]( i32 r ) -> i32 {
        return (int8_t)( r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2 );
    };
    // tags: LEFT keeps the first maximum of (s, a, b, a2, b2): d = 4 - tag; RIGHT the last of (s, a, b, a2): d = tag
    constexpr u32 tS = LEFT ? 4 : 0, tX = LEFT ? 3 : 1, tY = 2, tX2 = LEFT ? 1 : 3, tY2 = 0;
    const u32 K_X0 = pk_val( -q - e, tX ), K_Y0 = pk_val( -q - e, tY ), K_X20 = pk_val( -q2 - e2, tX2 ),
              K_Y20 = pk_val( -q2 - e2, tY2 ), K_V0 = pk_val( -q - e, 0 );
    const u32 K_TX = pk_val( 0, tX ), K_TY = pk_val( 0, tY ), K_TX2 = pk_val( 0, tX2 ), K_TY2 = pk_val( 0, tY2 );
    // flag thresholds: sign( K - a ) is "a > tag" for LEFT and "a >= tag" for RIGHT
    const u32 K_FX = pk_sub( K_TX, LEFT ? 0u : 0x00010001u ), K_FY = pk_sub( K_TY, LEFT ? 0u : 0x00010001u ),
              K_FX2 = pk_sub( K_TX2, LEFT ? 0u : 0x00010001u ), K_FY2 = pk_sub( K_TY2, LEFT ? 0u : 0x00010001u );
    const u32 K_Q = pk_val( q, 0 ), K_Q2 = pk_val( q2, 0 ), K_QE = pk_val( q + e, 0 ), K_QE2 = pk_val( q2 + e2, 0 );
    const u32 K_CLIP = pk_val( sc_mch, 0xff ), K_NEG = 0x80008000u, K_MATCH = pk_bcast( sc_mch );
    const u32 M_LANE0LO = lane == 0 ? 0x0000ffffu : 0u;
    // the constants of the per-diagonal path live in VGPRs: as SGPRs they overflow the scalar file (spills read back
    // with v_readlane inside the loop) and a VOP3P instruction takes only one scalar operand anyway
    const u32 V_CLIP = pk_opaque( K_CLIP );
    // score table (kswcpp_core.h:598-615): bytes 0 match, 1..3 mismatch, 4 = -e2 (either base is N), 5 = tag of s
    const u32 V_SCLO = pk_opaque( ( (u32)sc_mch & 0xffu ) | ( ( (u32)sc_mis & 0xffu ) * 0x01010100u ) );
    const u32 V_SCHI = pk_opaque( ( (u32)( -e2 ) & 0xffu ) | tS << 8 );
    const u32 V_Q = pk_opaque( K_Q ), V_Q2 = pk_opaque( K_Q2 ), V_QE = pk_opaque( K_QE ), V_QE2 = pk_opaque( K_QE2 );

    // First-column boundary of every diagonal r < qlen, tabulated in LDS once per job (16 bytes per diagonal: the words the
    // cell update takes for cell 0's left neighbour): [0] v = initOf(r) << 24, [1] H(-1, r) << 16, [2] query base r << 16.
    // The loop reads its entry with one ds_read_b128 instead of deriving the three values with ~20 scalar instructions and
    // five v_writelane per diagonal -- the scalar unit, shared by all waves of a CU, is this kernel's busiest port.
    // H(-1, r) = the offset kswcpp seeds H[0] with + the sum of initOf(0..r), in closed form (same sequence as the first row).
    auto hBoundary = [ & ]( i32 n ) -> i32 { // H(n-1, -1) = H(-1, n-1)
        const i32 a = max( 0, min( n, long_thres ) - 1 ); // cells 1..n-1 below long_thres
        const i32 has = long_thres >= 1 && long_thres < n ? 1 : 0;
        const i32 rest = ( n - 1 ) - a - has;
        return ( q + e ) - qe0 + ( n < 1 ? 0 : -( q + e ) - e * a + ( has ? long_diff : 0 ) - e2 * rest );
    };
    {
        uint4* tab = (uint4*)lds;
        for( i32 i = lane; i < qlen; i += 64 )
        {
            uint4 w;
            w.x = ( (u32)initOf( i ) & 0xffu ) << 24;
            w.y = (u32)hBoundary( i + 1 ) << 16;
            w.z = ( (u32)qbase( i ) & 0xffu ) << 16;
            w.w = 0;
            tab[ i ] = w;
        }
        __syncthreads( );
    }
#if defined( MA_KSW_PROF )
    asm volatile( "s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory" );
    const unsigned long long tpB = clock64( );
    if( !GLOBAL )
        prof[ 12 ] += tpB - tpA; // job descriptor + query bytes
#endif
    auto tgt2 = [ & ]( i32 t ) -> u32 { // target bases of cells t, t+1
        // an N of the target is coded 12, one of the query 4..5: base ^ base is 0 for a match, 1..3 for a mismatch and
        // >= 4 as soon as either is N (the score look-up below)
        if( t >= tlen )
            return 0u;
        u32 ab = tbase.pair( t ) & ( t + 1 < tlen ? 0x00ff00ffu : 0x000000ffu );
        if( TF::CLEAN )
            return ab;
        const u32 n = pk_lshr( ab, 2 ); // >= 1 where the code is >= 4
        return pk_bfi( pk_sub( 0u, pk_minu( n, 0x00010001u ) ), 0x000c000cu, ab );
    };
    auto uInit2 = [ & ]( i32 t ) -> u32 { // first-row u of cells t, t+1
        return ( ( (u32)initOf( t ) & 0xffu ) << 8 ) | ( ( (u32)initOf( t + 1 ) & 0xffu ) << 24 );
    };
    u32 U[ R ], V[ R ], X[ R ], Y[ R ], X2[ R ], Y2[ R ], T[ R ], Tn[ R ], H[ R ], Qf[ R ], TTpk[ R ], PB[ R ];
    i32 TT[ R ];
#pragma unroll
    for( int s = 0; s < R; s++ )
    {
        TT[ s ] = 128 * s + 2 * lane;
        TTpk[ s ] = (u32)TT[ s ] | (u32)( TT[ s ] + 1 ) << 16;
        PB[ s ] = pk_sub( pk_bcast( tlen - 1 ), TTpk[ s ] );
        U[ s ] = uInit2( TT[ s ] );
        V[ s ] = K_V0;
        X[ s ] = K_X0;
        Y[ s ] = K_Y0;
        X2[ s ] = K_X20;
        Y2[ s ] = K_Y20;
        T[ s ] = tgt2( TT[ s ] );
        Tn[ s ] = tgt2( TT[ s ] + RING ); // the target of the cells this lane takes over next: requested a ring ahead,
                                         // so that handing cells on never waits for memory
        H[ s ] = 0;
        Qf[ s ] = 0x00040004u;
    }
#if defined( MA_KSW_PROF )
    asm volatile( "s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory" );
    const unsigned long long tp0 = clock64( );
    if( !GLOBAL )
        prof[ 13 ] += tp0 - tpB; // target bytes + state initialisation
#endif
    // calcMaxScore of one diagonal (kswcpp_core.h:156-299) from the lanes' packed H (Hs) and cell offsets t - st0 (DDs)
    auto exactMax = [ & ]( const u32( &Hs )[ R ], const u32( &DDs )[ R ], i32 st0, i32 en0, i32& mH, i32& mT ) {
        // ---- the reference's max_t (kswcpp_core.h:156-299): 8 classes (t - st0) mod 8 over the chunks
        // [st0, en1), each class keeps its first maximum and the chunk base it came from, the initial
        // (H[en0], en0) wins ties, max_t is the largest of the classes' values; then [en1, en0) one by one
        const i32 pe = en0 & ( RING - 1 );
        u32 hreg = Hs[ 0 ];
#pragma unroll
        for( int s = 1; s < R; s++ )
            if( ( pe >> 7 ) == s )
                hreg = Hs[ s ];
        const i32 hEn0 = (i32)( (u32)lane_bcast( (i32)hreg, ( pe & 127 ) >> 1 ) << ( pe & 1 ? 0 : 16 ) ) >> 16;
        const i32 nS = ( ( en0 - st0 ) / 8 ) * 8; // cells of the 8-lane part
        i32 kLo = (i32)0x80000000, kHi = (i32)0x80000000;
#pragma unroll
        for( int s = 0; s < R; s++ )
        {
            const u32 inv = pk_sub( 0xffffffffu, DDs[ s ] ); // 0xffff - (t - st0): earlier chunks win ties
            const i32 lo = (i32)__builtin_amdgcn_perm( Hs[ s ], inv, 0x05040100u );
            const i32 hi = (i32)__builtin_amdgcn_perm( Hs[ s ], inv, 0x07060302u );
            if( ( DDs[ s ] & 0xffffu ) < (u32)nS )
                kLo = max( kLo, lo );
            if( ( DDs[ s ] >> 16 ) < (u32)nS )
                kHi = max( kHi, hi );
        }
        // lanes with equal (lane mod 4) hold the same two classes
        kLo = max( kLo, dpp_ctrl<0x124>( kLo ) );
        kHi = max( kHi, dpp_ctrl<0x124>( kHi ) );
        kLo = max( kLo, dpp_ctrl<0x128>( kLo ) );
        kHi = max( kHi, dpp_ctrl<0x128>( kHi ) );
        {
            auto a16 = __builtin_amdgcn_permlane16_swap( (u32)kLo, (u32)kLo, false, false );
            kLo = max( (i32)a16[ 0 ], (i32)a16[ 1 ] );
            auto b16 = __builtin_amdgcn_permlane16_swap( (u32)kHi, (u32)kHi, false, false );
            kHi = max( (i32)b16[ 0 ], (i32)b16[ 1 ] );
            auto a32 = __builtin_amdgcn_permlane32_swap( (u32)kLo, (u32)kLo, false, false );
            kLo = max( (i32)a32[ 0 ], (i32)a32[ 1 ] );
            auto b32 = __builtin_amdgcn_permlane32_swap( (u32)kHi, (u32)kHi, false, false );
            kHi = max( (i32)b32[ 0 ], (i32)b32[ 1 ] );
        }
        mH = hEn0, mT = en0;
        if( nS > 0 )
        {
            const i32 hl = kLo >> 16, hh = kHi >> 16;
            const i32 tl = hl > hEn0 ? st0 + ( ( 0xffff - ( kLo & 0xffff ) ) & ~7 ) : en0;
            const i32 th = hh > hEn0 ? st0 + ( ( 0xffff - ( kHi & 0xffff ) ) & ~7 ) : en0;
            i32 vh = max( max( hl, hh ), hEn0 ), vt = max( tl, th );
            vh = max( vh, dpp_ctrl<0xB1>( vh ) ); // the four lanes of a quad hold the eight classes
            vt = max( vt, dpp_ctrl<0xB1>( vt ) );
            vh = max( vh, dpp_ctrl<0x4E>( vh ) );
            vt = max( vt, dpp_ctrl<0x4E>( vt ) );
            mH = __builtin_amdgcn_readfirstlane( vh );
            mT = __builtin_amdgcn_readfirstlane( vt );
        }
        for( i32 t = st0 + nS; t < en0; ++t )
        {
            const i32 p = t & ( RING - 1 );
            u32 hr = Hs[ 0 ];
#pragma unroll
            for( int s = 1; s < R; s++ )
                if( ( p >> 7 ) == s )
                    hr = Hs[ s ];
            const i32 h = (i32)( (u32)lane_bcast( (i32)hr, ( p & 127 ) >> 1 ) << ( p & 1 ? 0 : 16 ) ) >> 16;
            if( h > mH )
                mH = h, mT = t;
        }
    };
    // A diagonal that raises ez.max needs the new value at once, but its position (max_t, max_q) only when the z-drop test or
    // the caller reads it -- and only that of the LAST raise: the lanes' H and offsets go to LDS, and the ~90 instructions of the
    // class-wise reduction run once per job instead of on every other diagonal of a good alignment (150 bp: 21.3 -> 18.9 ms)
    bool pend = false;
    i32 pSt0 = 0, pEn0 = 0, pR = 0;
    auto resolvePending = [ & ]( ) {
        if( !pend )
            return;
        u32 Hs[ R ], DDs[ R ];
#pragma unroll
        for( int s = 0; s < R; s++ )
        {
            const uint2 v = snap[ s * 64 + lane ];
            Hs[ s ] = v.x, DDs[ s ] = v.y;
        }
        i32 mH, mT;
        exactMax( Hs, DDs, pSt0, pEn0, mH, mT );
        ez.max_t = mT;
        ez.max_q = pR - mT;
        pend = false;
    };
    i32 recycled = 0; // cells below this index have been handed to cells RING further up
    // granularity of that hand-over: 16 cells when the ring has room for 15 dead cells beside the live window, else 8, 4 or
    // one lane (2 cells): each hand-over costs the same, so the coarsest one that fits is the cheapest per diagonal
    const i32 room = tlen <= RING ? 16 : RING - qlen + 1; // dead cells the ring can hold beside the live window
    const i32 gran = room >= 16 ? 16 : ( room >= 8 ? 8 : ( room >= 4 ? 4 : 2 ) );
    // cells handed on are >= RING: beyond long_thres their first-row difference is the constant -e2 (initOf)
    const bool uFar = RING > long_thres + 1;
    const u32 K_UFAR = pk_val( -e2, 0 );
    // (sic) When the two gap models were swapped, kswcpp's H[0] = v[0] - (q+e) uses the UNswapped sum, which offsets every
    // score of the matrix by (q+e)_swapped - (q+e)_given: part of hBoundary.
    auto hTopBefore = [ & ]( i32 n ) -> i32 { return hBoundary( n ); }; // H(n-1, -1): only the early stop reads it
    const u32 M_LANE0 = lane == 0 ? 0xffffffffu : 0u;
    const u32 S_X0 = K_X0 << 16, S_X20 = K_X20 << 16; // cell 0's left neighbour never had a gap open
    i32 boundPrev = 0x7fffffff, nextBound = 0;
    const i32 boundRate = max( 1, ( -sc_mis + sc_mch + 1 ) / 2 );
    const i32 nDiag = qlen + tlen - 1;
    bool stop = false;
    // z-drop schedule.  The maximum of diagonal r + 1 is at least that of diagonal r minus (q + e) of the cheaper gap model: the
    // best cell's right or lower neighbour can always open a gap from it (z is only ever clipped from above, and one of the two
    // neighbours exists until the last cell of the rectangle).  After a diagonal with maximum m the test -- every cell at or
    // below ez.max - zdrop - 1 -- can therefore not pass for ( m - (ez.max - zdrop - 1) ) / (q + e) diagonals.
    const i32 qeDrop = max( 1, (i32)q + (i32)e );
    const i32 zStep = J.zdrop >= 0 ? ( J.zdrop + qeDrop ) / qeDrop : 0x3fffffff;
    i32 zNext = J.zdrop >= 0 ? 0 : 0x7fffffff;
    u32 nCells = 0; // < 2^32: qlen <= 256 cells on at most w + 1 diagonals
    for( i32 r = 0; r < nDiag && !stop; ++r )
    {
        if( r > w )
            return false; // the band starts to cut the rectangle: not this kernel's regime
        const i32 st0 = max( 0, r - qlen + 1 ), en0 = min( r, tlen - 1 );
        // ---- recycle the 16-cell block that left the band
        if( ( st0 & ~( gran - 1 ) ) > recycled )
        {
            const i32 lim = recycled + gran;
#pragma unroll
            for( int s = 0; s < R; s++ )
                if( TT[ s ] < lim )
                {
                    TT[ s ] += RING;
                    TTpk[ s ] = (u32)TT[ s ] | (u32)( TT[ s ] + 1 ) << 16;
                    PB[ s ] = pk_sub( pk_bcast( tlen - 1 ), TTpk[ s ] );
                    U[ s ] = uFar ? K_UFAR : uInit2( TT[ s ] );
                    Y[ s ] = K_Y0;
                    Y2[ s ] = K_Y20;
                    T[ s ] = Tn[ s ];
                    Tn[ s ] = tgt2( TT[ s ] + RING );
                }
            recycled = lim;
        }
        // ---- neighbours t-1 of the previous diagonal; the query moves one cell up
        u32 xt1[ R ], vt1[ R ], x2t1[ R ], hup[ R ];
        {
            u32 px[ R ], pv[ R ], px2[ R ], ph[ R ], pq[ R ];
#pragma unroll
            for( int s = 0; s < R; s++ )
            {
                px[ s ] = lanes_ror1( X[ s ] );
                pv[ s ] = lanes_ror1( V[ s ] );
                px2[ s ] = lanes_ror1( X2[ s ] );
                ph[ s ] = GLOBAL ? 0u : lanes_ror1( H[ s ] );
                pq[ s ] = lanes_ror1( Qf[ s ] );
            }
#pragma unroll
            for( int s = 0; s < R; s++ )
            {
                const int sp = s == 0 ? R - 1 : s - 1; // lane 0 continues lane 63 of the previous slot of the ring
                u32 ax = R == 1 ? px[ s ] : pk_bfi( M_LANE0, px[ sp ], px[ s ] );
                u32 av = R == 1 ? pv[ s ] : pk_bfi( M_LANE0, pv[ sp ], pv[ s ] );
                u32 ax2 = R == 1 ? px2[ s ] : pk_bfi( M_LANE0, px2[ sp ], px2[ s ] );
                u32 ah = R == 1 ? ph[ s ] : pk_bfi( M_LANE0, ph[ sp ], ph[ s ] );
                u32 aq = R == 1 ? pq[ s ] : pk_bfi( M_LANE0, pq[ sp ], pq[ s ] );
                if( s == 0 && st0 == 0 )
                {
                    // cell 0 (slot 0, lane 0, low half until it is recycled at r >= qlen + 15): first-column carry-in
                    // (kswcpp_core.h:562-579) and the query base that enters the band.  The shift takes the low half of
                    // lane 0 from the HIGH half of its predecessor: lane 0 takes this diagonal's table entry there.
                    const uint4 bnd = ( (const uint4*)lds )[ r ];
                    ax = pk_bfi( M_LANE0, S_X0, ax );
                    ax2 = pk_bfi( M_LANE0, S_X20, ax2 );
                    av = pk_bfi( M_LANE0, bnd.x, av );
                    if( !GLOBAL )
                        ah = pk_bfi( M_LANE0, bnd.y, ah );
                    aq = pk_bfi( M_LANE0, bnd.z, aq );
                }
                xt1[ s ] = cells_shift1( X[ s ], ax );
                vt1[ s ] = cells_shift1( V[ s ], av );
                x2t1[ s ] = cells_shift1( X2[ s ], ax2 );
                hup[ s ] = GLOBAL ? 0u : cells_shift1( H[ s ], ah );
                Qf[ s ] = cells_shift1( Qf[ s ], aq );
            }
        }
        const u32 st0pk = pk_bcast_s( st0 ), wpk = pk_bcast_s( en0 - st0 + 1 );
        uint8_t* prow = P + (size_t)r * RING;
        u32 Hm[ R ], DD[ R ], LMs[ R ];
        const int sLo = ( st0 & ( RING - 1 ) ) >> 7, sHi = ( en0 & ( RING - 1 ) ) >> 7;
        const bool allSlots = en0 - st0 >= 128;
#pragma unroll
        for( int s = 0; s < R; s++ )
        {
            if( R > 1 && !allSlots && s != sLo && s != sHi )
            {
                // no live cell in this slot: nothing to update (its registers are only read by the other slot)
                Hm[ s ] = K_NEG;
                DD[ s ] = 0xffffffffu;
                LMs[ s ] = 0;
                continue;
            }
            // ---- live cells of this slot: st0 <= t <= en0
            const u32 dd = pk_sub( TTpk[ s ], st0pk ); // t - st0 (mod 2^16)
            const u32 LM = pk_opaque( pk_nonzero15( pk_subsatu( wpk, dd ) ) ); // 0xffff where live (width <= 514)
            DD[ s ] = dd;
            LMs[ s ] = LM;
            // ---- score: match / mismatch, -e2 when either base is N (kswcpp_core.h:598-615)
            // one byte permute as look-up: index min(q ^ t, 4) -> value byte, index 5 -> the tag of s
            const u32 sel = ( pk_minu( T[ s ] ^ Qf[ s ], 0x00040004u ) << 8 ) | 0x00050005u;
            u32 z = __builtin_amdgcn_perm( V_SCHI, V_SCLO, sel );
            // ---- DP cell (kswcpp_core.h:653-766)
            const u32 ut = U[ s ];
            u32 a = pk_add( xt1[ s ], vt1[ s ] );
            u32 b = pk_add( Y[ s ], ut );
            u32 a2 = pk_add( x2t1[ s ], vt1[ s ] );
            u32 b2 = pk_add( Y2[ s ], ut );
            u32 d;
            if( LEFT )
            {
                z = pk_max( pk_max( z, a ), pk_max( pk_max( b, a2 ), b2 ) );
                d = pk_sub( 0x00040004u, z & 0x00070007u );
            }
            else
            {
                z = pk_max( pk_max( z, a ), pk_max( b, a2 ) );
                d = z & 0x00070007u;
                z = pk_max( z, b2 ); // state 4 is never recorded (kswcpp_core.h:693-699)
            }
            const u32 zc = pk_min( z, V_CLIP ) & 0xff00ff00u;
            const u32 nu = pk_sub( zc, vt1[ s ] ), nv = pk_sub( zc, ut );
            u32 tmp = pk_sub( zc, V_Q );
            a = pk_sub( a, tmp );
            b = pk_sub( b, tmp );
            tmp = pk_sub( zc, V_Q2 );
            a2 = pk_sub( a2, tmp );
            b2 = pk_sub( b2, tmp );
            const u32 nx = pk_sub( pk_max( a, K_TX ), V_QE ), ny = pk_sub( pk_max( b, K_TY ), V_QE );
            const u32 nx2 = pk_sub( pk_max( a2, K_TX2 ), V_QE2 ), ny2 = pk_sub( pk_max( b2, K_TY2 ), V_QE2 );
            // continuation flags = sign bit of a packed difference (no overflow: the scoring guard of ksw_ext_slots keeps
            // every difference vector far inside int8): LEFT a > 0, RIGHT !(a < 0) (kswcpp_core.h:653-699)
            const u32 fa = pk_sub( K_FX, a ), fb = pk_sub( K_FY, b ), fa2 = pk_sub( K_FX2, a2 ), fb2 = pk_sub( K_FY2, b2 );
            d = and_or( fa >> 12, 0x00080008u, and_or( fb >> 11, 0x00100010u, and_or( fa2 >> 10, 0x00200020u, and_or( fb2 >> 9, 0x00400040u, d ) ) ) );
            // ---- commit: u, y, y2 of cells that are not born yet keep their first-row initialisation
            U[ s ] = pk_bfi( LM, nu, ut );
            Y[ s ] = pk_bfi( LM, ny, Y[ s ] );
            Y2[ s ] = pk_bfi( LM, ny2, Y2[ s ] );
            V[ s ] = nv;
            X[ s ] = nx;
            X2[ s ] = nx2;
            if( LM ) // only lanes with a live cell write their two direction bytes (keeps HBM traffic at the live band)
                *(uint16_t*)( prow + 128 * s + 2 * lane ) = (uint16_t)__builtin_amdgcn_perm( 0u, d, 0x0c0c0200u );
            // ---- H(t, r) = H(t-1, r-1) + u(t, r)
            if( !GLOBAL )
            {
                const u32 hn = pk_add( hup[ s ], pk_ashr8( nu ) );
                H[ s ] = hn;
                Hm[ s ] = pk_bfi( LM, hn, K_NEG );
            }
        }
        nCells += (u32)( en0 - st0 + 1 );
#if defined( MA_EXP_SALU ) // experiment (tools/dp_bound_experiment.sh): which issue port bounds the loop?
        {
            u32 t0 = (u32)__builtin_amdgcn_readfirstlane( r );
            asm volatile( "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                          "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                          : "+s"( t0 ) );
            if( t0 == 0xdeadbeefu )
                nCells++;
        }
#endif
#if defined( MA_EXP_VALU )
        {
            u32 t0 = U[ 0 ];
            asm volatile( "v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n"
                          "v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %0, %0, %0\n"
                          : "+v"( t0 ) );
            if( t0 == 0xdeadbeefu )
                nCells++;
        }
#endif
        if( GLOBAL )
            continue;
        // ---- the diagonal's maximum
        u32 hm = Hm[ 0 ];
#pragma unroll
        for( int s = 1; s < R; s++ )
            hm = pk_max( hm, Hm[ s ] );
        // a cheap wave-uniform test instead of a full reduction: does any cell exceed ez.max?
        const u32 ezpk = pk_bcast_s( (i32)ez.max );
        const bool newMax = __any( pk_max( hm, ezpk ) != ezpk ) != 0;
        if( newMax )
        {
            // (the z-drop branch cannot be taken on a raise)
            ez.max = (u32)wave_max_i32( max( (i32)( hm << 16 ) >> 16, (i32)hm >> 16 ) ) & 0x7fffffffu;
#pragma unroll
            for( int s = 0; s < R; s++ )
                snap[ s * 64 + lane ] = make_uint2( H[ s ], DD[ s ] );
            pend = true, pSt0 = st0, pEn0 = en0, pR = r;
            zNext = r + zStep; // this diagonal's maximum IS ez.max: see zStep
        }
        else if( __builtin_expect( r >= zNext, 0 ) )
        {
            // ---- z-drop (kswcpp_core.h:22-44) needs a diagonal whose maximum lies more than zdrop below ez.max; on the job's
            // schedule (zStep below), not on every diagonal (round 4: a packed compare, a ballot and ~10 scalar instructions each)
            const i32 thr = (i32)ez.max - J.zdrop - 1; // candidates: every cell at or below it
            const i32 gm = wave_max_i32( max( (i32)( hm << 16 ) >> 16, (i32)hm >> 16 ) );
            if( thr < -32768 )
                zNext = r + 1; // (packed int16 cells cannot say: as before, no test while the threshold is out of range)
            else if( gm > thr )
                zNext = r + ( gm - thr + qeDrop - 1 ) / qeDrop; // >= r + 1
            else
            {
                zNext = r + 1; // every cell is below the threshold: the test is due on every diagonal from here on
                i32 mH, mT;
                exactMax( H, DD, st0, en0, mH, mT );
                resolvePending( );
                // ksw_apply_zdrop (kswcpp_core.h:22-44), is_rot = 1; mH == max_H
                if( mH > (i32)ez.max )
                {
                    ez.max = (u32)mH & 0x7fffffffu;
                    ez.max_t = mT;
                    ez.max_q = r - mT;
                }
                else if( mT >= ez.max_t && r - mT >= ez.max_q )
                {
                    const i32 tl = mT - ez.max_t, ql = ( r - mT ) - ez.max_q;
                    const i32 l = tl > ql ? tl - ql : ql - tl;
                    if( J.zdrop >= 0 && (i32)( ez.max - (u32)mH ) > J.zdrop + l * e2 )
                    {
                        ez.zdropped = 1;
                        stop = true;
                    }
                }
            }
        }
        // ---- early stop (ksw_reg.h): no later cell can exceed ez.max
        // The bound needs two consecutive diagonals (B_r and B_{r-1}) and costs ~60 instructions each time.  A pair that does not
        // stop the job tells how far off the stop is: along its diagonal chain a cell loses at most mismatch + match per two
        // diagonals (z >= the mismatch score, one step less to go), so a bound that exceeds ez.max by g cannot fall to it in fewer
        // than g / rate diagonals -- the next pair is evaluated then, not two diagonals later (a junk extension: ~6 pairs instead
        // of ~60; skipping is only ever late, never wrong: the job just runs until a later pair proves the stop)
        if( !newMax && !stop && r >= qlen - 1 && r >= nextBound )
        {
            const u32 qo = pk_bcast_s( qlen - 1 - r );
            u32 bm = K_NEG;
#pragma unroll
            for( int s = 0; s < R; s++ )
            {
                const u32 pot = pk_min( pk_add( TTpk[ s ], qo ), PB[ s ] );
                const u32 bnd = pk_mad( pot, K_MATCH, H[ s ] );
                bm = pk_max( bm, pk_bfi( LMs[ s ], bnd, K_NEG ) );
            }
            const i32 bound = wave_max_i32( max( (i32)( bm << 16 ) >> 16, (i32)bm >> 16 ) );
            const i32 all = max( max( bound, boundPrev ), hTopBefore( r ) + sc_mch * qlen );
            if( r >= qlen && all <= (i32)ez.max )
                stop = true;
            else if( boundPrev != 0x7fffffff && r >= qlen )
            {
                nextBound = r + 1 + max( 0, ( max( bound, hTopBefore( r ) + sc_mch * qlen ) - (i32)ez.max ) / boundRate - 1 );
                boundPrev = 0x7fffffff;
            }
            else
                boundPrev = bound;
        }
        else
            boundPrev = 0x7fffffff;
    }
    cells += nCells;
#if defined( MA_KSW_PROF )
    const unsigned long long tp1 = clock64( );
#endif
    if( !GLOBAL )
        resolvePending( );
    __syncthreads( ); // direction bytes visible to the back-trace
    if( !GLOBAL && ( ez.max_t < 0 || ez.max_q < 0 ) )
        return true;
    ksw_backtrack_ring<RING>( P, cig, J.flag, GLOBAL ? tlen - 1 : ez.max_t, GLOBAL ? qlen - 1 : ez.max_q, nCigar,
                              pathSteps, lds, ldsBytes, qlen, tlen );
#if defined( MA_KSW_PROF )
    if( !GLOBAL )
    {
        prof[ 4 ] += tp1 - tp0; // diagonal loop
        prof[ 5 ] += clock64( ) - tp1; // back-trace
        prof[ 6 ] += (unsigned long long)nCells;
        prof[ 7 ] += 1ull;
    }
    else
        prof[ 8 ] += 1ull;
#endif
    return true;
}
} // namespace ma
#endif
