// fm_device.h -- FMD-index primitives on the reference's 64-byte occ/BWT blocks.
// Restates FMIndex::bwt_occ4 (fMIndex.h:446-510), bwt_2occ4 (671-690), extend_backward
// (fMIndex.cpp:21-101), init_interval (fMIndex.h:768-775), bwt_B0/bwt_occ/bwt_invPsi/bwt_sa
// (fMIndex.h:268-343, 788-814) with popcounts on masked 2-bit words instead of the 256-entry
// byte table (identical integer results).
#pragma once
#include "ma_common.h"

namespace ma
{
struct Block64 // one occ/BWT block held in registers: 4 x u64 counts + 8 x u32 words
{
    u64 c[ 4 ];
    u32 w[ 8 ];
};

MA_HD void load_block( const u32* bwt, u64 blk, Block64& b )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    const uint4* p = reinterpret_cast<const uint4*>( bwt + ( blk << 4 ) );
    uint4 v0 = p[ 0 ], v1 = p[ 1 ], v2 = p[ 2 ], v3 = p[ 3 ]; // 4 x global_load_dwordx4 = one 64-B line
    b.c[ 0 ] = (u64)v0.x | ( (u64)v0.y << 32 );
    b.c[ 1 ] = (u64)v0.z | ( (u64)v0.w << 32 );
    b.c[ 2 ] = (u64)v1.x | ( (u64)v1.y << 32 );
    b.c[ 3 ] = (u64)v1.z | ( (u64)v1.w << 32 );
    b.w[ 0 ] = v2.x, b.w[ 1 ] = v2.y, b.w[ 2 ] = v2.z, b.w[ 3 ] = v2.w;
    b.w[ 4 ] = v3.x, b.w[ 5 ] = v3.y, b.w[ 6 ] = v3.z, b.w[ 7 ] = v3.w;
#else
    const u32* p = bwt + ( blk << 4 );
    for( int i = 0; i < 4; i++ )
        b.c[ i ] = (u64)p[ 2 * i ] | ( (u64)p[ 2 * i + 1 ] << 32 );
    for( int i = 0; i < 8; i++ )
        b.w[ i ] = p[ 8 + i ];
#endif
}

// counts of A,C,G,T in BWT rows [block start .. block start + within] (within in 0..127), added to
// the block's cumulative counters.
MA_HD void occ4_in_block( const Block64& b, u32 within, u64 cnt[ 4 ] )
{
    // per word: keep the top nsym symbols (two shifts by nsym <= 16 give the mask for every nsym in 0..16 without
    // a special case), then three popcounts: H = set high bits, L = set low bits, T = both (the symbol 3);
    // G = H - T, C = L - T, A = symbols - T - G - C once at the end.
    u32 H = 0, Lo = 0, T = 0;
    const u32 nsymTotal = within + 1;
#pragma unroll
    for( int w = 0; w < 8; w++ )
    {
        const i32 rem = (i32)nsymTotal - 16 * w;
        const u32 nsym = (u32)( rem < 0 ? 0 : ( rem > 16 ? 16 : rem ) );
        const u32 drop = ( 0xffffffffu >> nsym ) >> nsym; // the low 32 - 2 nsym bits
        const u32 x = b.w[ w ] & ~drop;
        const u32 hi = ( x >> 1 ) & 0x55555555u, lo = x & 0x55555555u;
        H += (u32)popc32( hi );
        Lo += (u32)popc32( lo );
        T += (u32)popc32( hi & lo );
    }
    const u32 g = H - T, c = Lo - T, a = nsymTotal - T - g - c;
    cnt[ 0 ] = b.c[ 0 ] + a;
    cnt[ 1 ] = b.c[ 1 ] + c;
    cnt[ 2 ] = b.c[ 2 ] + g;
    cnt[ 3 ] = b.c[ 3 ] + T;
}

// x.L2[i] as a register value.  IndexView is a kernel argument; without the (empty) asm the compiler turns a
// select between L2 entries into ONE vector load from a selected kernarg address, i.e. a memory round trip that
// depends on the base just read -- on the critical path of every FM step.
MA_HD u64 l2v( const IndexView& x, int i )
{
    u64 v = x.L2[ i ];
#if defined( __HIP_DEVICE_COMPILE__ )
    asm( "" : "+s"( v ) );
#endif
    return v;
}

MA_HD void init_interval( const IndexView& x, u32 c, i64 ik[ 3 ] ) // fMIndex.h:768-775
{
    const u64 lo = c == 0 ? l2v( x, 0 ) : ( c == 1 ? l2v( x, 1 ) : ( c == 2 ? l2v( x, 2 ) : l2v( x, 3 ) ) );
    const u64 hi = c == 0 ? l2v( x, 1 ) : ( c == 1 ? l2v( x, 2 ) : ( c == 2 ? l2v( x, 3 ) : l2v( x, 4 ) ) );
    const u64 rc = c == 0 ? l2v( x, 3 ) : ( c == 1 ? l2v( x, 2 ) : ( c == 2 ? l2v( x, 1 ) : l2v( x, 0 ) ) );
    ik[ 0 ] = (i64)lo + 1;
    ik[ 1 ] = (i64)rc + 1;
    ik[ 2 ] = (i64)( hi - lo );
}

// One FMD backward step. Reads the block of row k-1 and of row l-1 (one 64-B line each; the second
// load is skipped when both rows fall into the same block). nblocks returns 0/1/2 for the
// algorithmic-bytes counter.
MA_HD void extend_backward( const IndexView& x, const i64 ik[ 3 ], u32 c, i64 ok[ 3 ], u32& nblocks )
{
    nblocks = 0;
    if( c >= 4 )
    {
        ok[ 0 ] = ok[ 1 ] = ok[ 2 ] = 0;
        return;
    }
    const i64 start = ik[ 0 ], size = ik[ 2 ], end = start + size;
    i64 k = start - 1, l = end - 1;
    u64 cntk[ 4 ] = { 0, 0, 0, 0 }, cntl[ 4 ] = { 0, 0, 0, 0 };
    const i64 kk = k - ( k >= x.primary ), ll = l - ( l >= x.primary ); // '$' is not stored
    const bool hasK = k != -1, hasL = l != -1;
    Block64 bk, bl;
    if( hasK )
    {
        load_block( x.bwt, (u64)kk >> 7, bk );
        nblocks++;
    }
    if( hasL )
    {
        if( hasK && ( (u64)kk >> 7 ) == ( (u64)ll >> 7 ) )
            bl = bk;
        else
        {
            load_block( x.bwt, (u64)ll >> 7, bl );
            nblocks++;
        }
    }
    if( hasK )
        occ4_in_block( bk, (u32)( (u64)kk & 127 ), cntk );
    if( hasL )
        occ4_in_block( bl, (u32)( (u64)ll & 127 ), cntl );
    // selects instead of c-indexed arrays: a dynamically indexed local array would live in scratch memory
    const u64 s0 = cntl[ 0 ] - cntk[ 0 ], s1 = cntl[ 1 ] - cntk[ 1 ], s2 = cntl[ 2 ] - cntk[ 2 ], s3 = cntl[ 3 ] - cntk[ 3 ];
    u64 c2 = (u64)ik[ 1 ];
    if( start <= x.primary && end > x.primary )
        c2++;
    // cntk_2[i] = cntk_2[i-1] + cnts[3-(i-1)]; result uses cntk_2[3-c]
    const u64 a0 = c2, a1 = a0 + s3, a2 = a1 + s2, a3 = a2 + s1;
    const u64 l2c = c == 0 ? l2v( x, 0 ) : ( c == 1 ? l2v( x, 1 ) : ( c == 2 ? l2v( x, 2 ) : l2v( x, 3 ) ) );
    const u64 ckc = c == 0 ? cntk[ 0 ] : ( c == 1 ? cntk[ 1 ] : ( c == 2 ? cntk[ 2 ] : cntk[ 3 ] ) );
    ok[ 0 ] = (i64)( l2c + ckc + 1 );
    ok[ 1 ] = (i64)( c == 0 ? a3 : ( c == 1 ? a2 : ( c == 2 ? a1 : a0 ) ) );
    ok[ 2 ] = (i64)( c == 0 ? s0 : ( c == 1 ? s1 : ( c == 2 ? s2 : s3 ) ) );
}

// bwt_invPsi (fMIndex.h:329-343): one 64-B block per LF step (B0 and occ hit the same block)
MA_HD i64 inv_psi( const IndexView& x, i64 k )
{
    if( k == x.primary )
        return 0;
    const i64 xx = k - ( k > x.primary );
    Block64 b;
    load_block( x.bwt, (u64)xx >> 7, b );
    const u32 within = (u32)( (u64)xx & 127 );
    const u32 c = ( b.w[ within >> 4 ] >> ( ( ~within & 15 ) << 1 ) ) & 3;
    // bwt_occ(k, c): k == n handled by the caller contract (rows are < n here); k -= (k >= primary)
    // equals xx for k != primary
    u64 cnt[ 4 ];
    occ4_in_block( b, within, cnt );
    const u64 l2c = c == 0 ? l2v( x, 0 ) : ( c == 1 ? l2v( x, 1 ) : ( c == 2 ? l2v( x, 2 ) : l2v( x, 3 ) ) );
    const u64 cc = c == 0 ? cnt[ 0 ] : ( c == 1 ? cnt[ 1 ] : ( c == 2 ? cnt[ 2 ] : cnt[ 3 ] ) );
    return (i64)( l2c + cc );
}

// SA value of a sampled row (k a multiple of the sampling interval in use)
MA_HD i64 sa_sample( const IndexView& x, i64 k )
{
    return ( x.sa_dense ? x.sa_dense : x.sa )[ k >> x.sa_shift ];
}

// bwt_sa (fMIndex.h:788-814)
MA_HD i64 bwt_sa( const IndexView& x, i64 k, u32& steps )
{
    i64 s = 0;
    const i64 mask = ( (i64)1 << x.sa_shift ) - 1;
    while( k & mask )
    {
        ++s;
        k = inv_psi( x, k );
    }
    steps = (u32)s;
    return s + sa_sample( x, k );
}

// ---- Pack helpers (pack.h:900-1087) ----
MA_HD u32 fwd_base( const IndexView& x, u64 p ) // getNucleotideOnPos (pack.h:173-176)
{
    return ( x.pac[ p >> 2 ] >> ( ( ~p & 3 ) << 1 ) ) & 3;
}
MA_HD u32 text_base( const IndexView& x, u64 p ) // vExtractSubsection (pack.h:1147-1236)
{
    return p < x.F ? fwd_base( x, p ) : 3u - fwd_base( x, x.n - 1 - p );
}
MA_HD i64 seq_id_for_position( const IndexView& x, u64 pos ) // uiSequenceIdForPosition (pack.h:933-990)
{
    const i64 iAbs = pos >= x.F ? (i64)( x.n - ( pos + 1 ) ) : (i64)pos;
    u64 l = 0, m = 0, r = (u64)x.n_contigs;
    while( l < r )
    {
        m = ( l + r ) / 2;
        if( iAbs >= (i64)x.cstart[ m ] )
        {
            if( m == (u64)x.n_contigs - 1 )
                break;
            if( iAbs < (i64)x.cstart[ m + 1 ] )
                break;
            l = m + 1;
        }
        else
            r = m;
    }
    return (i64)m;
}
MA_HD bool on_rev( const IndexView& x, u64 p )
{
    return p >= x.F;
}
MA_HD u64 to_rev( const IndexView& x, u64 p )
{
    return x.n - ( p + 1 );
}
MA_HD i64 seq_id_or_rev( const IndexView& x, u64 p ) // pack.h:1029-1034
{
    if( on_rev( x, p ) )
        return seq_id_for_position( x, to_rev( x, p ) ) * 2 + 1;
    return seq_id_for_position( x, p ) * 2;
}
MA_HD u64 end_of_seq_or_rev( const IndexView& x, i64 id ) // pack.h:1040-1045 (sic: -1 on the reverse strand)
{
    if( id % 2 == 1 )
        return to_rev( x, x.cstart[ id / 2 ] ) - 1;
    return x.cstart[ id / 2 ] + x.clen[ id / 2 ];
}
MA_HD u64 start_of_seq_or_rev( const IndexView& x, i64 id ) // pack.h:1047-1052
{
    if( id % 2 == 1 )
        return to_rev( x, x.cstart[ id / 2 ] + x.clen[ id / 2 ] ) + 1;
    return x.cstart[ id / 2 ];
}
MA_HD bool bridging( const IndexView& x, u64 b, u64 size ) // pack.h:1072-1087
{
    if( size == 0 )
        return false;
    const i64 id = seq_id_or_rev( x, b );
    return ( on_rev( x, b ) != on_rev( x, b + size - 1 ) ) || ( id != seq_id_or_rev( x, b + size - 1 ) );
}
} // namespace ma
