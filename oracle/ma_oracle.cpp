// ============================================================================
// TEST INFRASTRUCTURE ONLY -- see ma_oracle.h.  Plain scalar C++ restatement of the reference's
// algorithm for the seed-and-extend path.  Reference citations are relative to /root/reference.
// Validated against the compiled reference (oracle/_ref) by tests/test_oracle_vs_ref.py and
// against tests/golden/*.  Parity: PINNED.
// ============================================================================
#include "ma_oracle.h"

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

typedef int64_t i64;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------
// parameters (libs/ms/inc/ms/util/parameter.h:521-1060, presets 1079-1128)
// ---------------------------------------------------------------------------------------------
extern "C" void ma_or_params_default( ma_or_params* p )
{
    memset( p, 0, sizeof( *p ) );
    p->seeding_technique = 0;
    p->min_seed_len = 16;
    p->min_ambiguity = 0;
    p->max_ambiguity = 100;
    p->min_seed_size_drop = 15;
    p->max_num_soc = 30;
    p->min_num_soc = 1;
    p->harm_score_min = 18;
    p->max_score_lookahead = 3;
    p->switch_qlen = 800;
    p->min_delta_dist = 16;
    p->max_gap_area = 20;
    p->padding = 1000;
    p->bandwidth_ext = 512;
    p->min_bandwidth_gap = 20;
    p->zdrop = 200;
    p->sv_penalty = 100;
    p->match = 2;
    p->mismatch = 4;
    p->gap = 4;
    p->extend = 2;
    p->gap2 = 24;
    p->extend2 = 1;
    p->disable_heuristics = 0;
    p->soc_width = 0;
    p->srand_seed = 1;
    p->genome_size_disable = 10000000;
    p->rel_min_seed_size_amount = 0.005;
    p->harm_score_min_rel = 0.002;
    p->soc_score_decrease_tol = 0.1;
    p->score_diff_tol = 0.0001;
    p->max_delta_dist = 0.1;
    p->min_alignment_score = 75;
    p->report_n_best = 0;
    p->max_supplementary = 1;
    p->max_overlap_supplementary = 0.1;
    p->search_inversions = 0;
    p->zdrop_inversion = 100;
    p->use_paired_reads = 0;
    p->pad_ = 0;
    p->mean_paired_dist = 400;
    p->std_paired_dist = 150;
    p->paired_bonus = 1.25;
}

extern "C" void ma_or_params_illumina( ma_or_params* p )
{
    ma_or_params_default( p ); // parameter.h:1083-1087
    p->seeding_technique = 1;
    p->max_ambiguity = 500;
    p->min_num_soc = 10;
    p->max_num_soc = 20;
}

extern "C" void ma_or_params_pacbio( ma_or_params* p )
{
    ma_or_params_default( p ); // parameter.h:1096-1098
    p->max_supplementary = 100;
    p->min_num_soc = 5;
}

extern "C" void ma_or_params_nanopore( ma_or_params* p )
{
    ma_or_params_pacbio( p ); // parameter.h:1101-1104
    p->seeding_technique = 1;
}

// ---------------------------------------------------------------------------------------------
// Index: fMIndex.h:195-230 (data), fMIndex.cpp:152-314 (construction), pack.h:586-698 (2-bit pack)
// ---------------------------------------------------------------------------------------------
struct ma_or_index
{
    u64 n = 0; // uiRefSeqLength: forward + reverse strand
    u64 F = 0; // forward strand length
    i64 primary = 0;
    u64 L2[ 6 ] = { 0, 0, 0, 0, 0, 0 };
    std::vector<uint32_t> bwt; // occ-injected BWT words (fMIndex.cpp:204-264)
    std::vector<i64> sa; // every 32nd SA value, sa[0] = -1 (fMIndex.cpp:266-314)
    std::vector<uint8_t> pac; // forward strand, 2 bit/base, MSB first (pack.h:676-688)
    std::vector<u64> cstart, clen; // contig table (pack.h SequenceInPack)

    inline uint8_t fwdBase( u64 p ) const // Pack::getNucleotideOnPos
    {
        return ( pac[ p >> 2 ] >> ( ( ~p & 3 ) << 1 ) ) & 3;
    }
    // Pack::vExtract / vExtractSubsection (pack.h:1147-1236): position on the doubled text
    inline uint8_t textBase( u64 p ) const
    {
        return p < F ? fwdBase( p ) : (uint8_t)( 3 - fwdBase( n - 1 - p ) );
    }
};

namespace
{
// work counters (thread local; summed by the batch driver)
struct Counters
{
    u64 v[ 8 ] = { 0, 0, 0, 0, 0, 0, 0, 0 };
};
thread_local Counters* tlsCnt = nullptr;
inline void cnt( int i, u64 x = 1 )
{
    if( tlsCnt )
        tlsCnt->v[ i ] += x;
}

// ---- suffix sorting of T' $ for the oracle's own index builder (any correct suffix sort yields the
// reference's BWT; is.cpp / bwt_large.cpp are merely the reference's choice of algorithm) ----
struct SufCmp
{
    const std::vector<uint8_t>& t;
    const std::vector<u64>& K; // 32-mer starting at i (zero padded)
    u64 n;
    bool operator( )( u64 a, u64 b ) const
    {
        if( a == b )
            return false;
        u64 i = a, j = b;
        while( i + 32 <= n && j + 32 <= n )
        {
            if( K[ i ] != K[ j ] )
                return K[ i ] < K[ j ];
            i += 32;
            j += 32;
        }
        while( i < n && j < n )
        {
            if( t[ i ] != t[ j ] )
                return t[ i ] < t[ j ];
            i++;
            j++;
        }
        return i == n; // the suffix that hits '$' first is smaller
    }
};

void buildFromText( ma_or_index& x, const std::vector<uint8_t>& t )
{
    const u64 n = t.size( );
    x.n = n;
    // L2 (fMIndex.cpp:171-183)
    for( int i = 0; i < 6; i++ )
        x.L2[ i ] = 0;
    for( u64 i = 0; i < n; i++ )
        x.L2[ 1 + t[ i ] ]++;
    for( int i = 2; i <= 4; i++ )
        x.L2[ i ] += x.L2[ i - 1 ];
    // suffix array of t (without the '$' suffix, which is row 0)
    std::vector<u64> K( n + 1, 0 );
    {
        u64 k = 0;
        for( u64 i = n; i-- > 0; )
        {
            k = ( k >> 2 ) | ( (u64)t[ i ] << 62 );
            K[ i ] = k;
        }
    }
    std::vector<u64> sa( n );
    for( u64 i = 0; i < n; i++ )
        sa[ i ] = i;
    // bucket by the leading 12-mer to keep std::sort ranges small
    {
        const int B = 24;
        std::vector<u64> cntB( ( 1u << B ) + 1, 0 );
        auto key = [ & ]( u64 i ) -> u64 {
            // suffixes shorter than 12 need exact handling -> put them via full comparator later;
            // zero padding sorts them before equal-prefix longer ones, which is correct ('$' smallest)
            return K[ i ] >> ( 64 - B );
        };
        for( u64 i = 0; i < n; i++ )
            cntB[ key( i ) + 1 ]++;
        for( size_t b = 1; b < cntB.size( ); b++ )
            cntB[ b ] += cntB[ b - 1 ];
        std::vector<u64> pos( cntB.begin( ), cntB.end( ) - 1 );
        for( u64 i = 0; i < n; i++ )
            sa[ pos[ key( i ) ]++ ] = i;
        SufCmp cmp{ t, K, n };
        for( size_t b = 0; b + 1 < cntB.size( ); b++ )
            if( cntB[ b + 1 ] - cntB[ b ] > 1 )
                std::sort( sa.begin( ) + cntB[ b ], sa.begin( ) + cntB[ b + 1 ], cmp );
    }
    // BWT rows: row 0 is '$...' whose BWT char is t[n-1]; row r>=1 is suffix sa[r-1].
    // primary = row whose suffix is the whole text (BWT char '$', not stored) (is.cpp is_bwt contract).
    std::vector<uint8_t> bw( n ); // '$'-removed BWT string of length n
    {
        u64 o = 0;
        bw[ o++ ] = t[ n - 1 ];
        for( u64 r = 0; r < n; r++ )
        {
            if( sa[ r ] == 0 )
            {
                x.primary = (i64)( r + 1 );
                continue;
            }
            bw[ o++ ] = t[ sa[ r ] - 1 ];
        }
    }
    // occ injection (fMIndex.cpp:204-264): per 128 nt a block of 4 x u64 counts then 8 x u32 packed words
    const u64 nOcc = ( n + 127 ) / 128 + 1;
    x.bwt.assign( ( n + 15 ) / 16 + nOcc * 8, 0 );
    {
        u64 c[ 4 ] = { 0, 0, 0, 0 };
        u64 k = 0;
        for( u64 i = 0; i < n; i++ )
        {
            if( i % 128 == 0 )
            {
                memcpy( &x.bwt[ k ], c, 32 );
                k += 8;
            }
            if( i % 16 == 0 )
                k++;
            x.bwt[ k - 1 ] |= (uint32_t)bw[ i ] << ( ( 15 - ( i & 15 ) ) << 1 );
            c[ bw[ i ] ]++;
        }
        memcpy( &x.bwt[ k ], c, 32 );
    }
    // sampled SA (fMIndex.cpp:266-314): sa[r/32] = SA[r] for rows r % 32 == 0; sa[0] := -1
    x.sa.assign( ( n + 32 ) / 32, 0 );
    for( u64 r = 32; r <= n; r += 32 )
        x.sa[ r / 32 ] = (i64)sa[ r - 1 ];
    x.sa[ 0 ] = -1;
}

// ---- FM-index primitives ----
// counts of the four symbols among the first nsym symbols (MSB first) of a 16-symbol word
inline void wordCounts( uint32_t w, unsigned nsym, u64 add[ 4 ] )
{
    if( nsym == 0 )
        return;
    const uint32_t mask = nsym >= 16 ? 0xffffffffu : ~( ( 1u << ( ( 16 - nsym ) << 1 ) ) - 1u );
    const uint32_t x = w & mask;
    const uint32_t hi = ( x >> 1 ) & 0x55555555u, lo = x & 0x55555555u;
    const unsigned t = __builtin_popcount( hi & lo ), g = __builtin_popcount( hi & ~lo ),
                   c = __builtin_popcount( ~hi & lo );
    add[ 3 ] += t;
    add[ 2 ] += g;
    add[ 1 ] += c;
    add[ 0 ] += ( nsym >= 16 ? 16 : nsym ) - t - g - c; // masked-off positions read as A and are not counted
}

// bwt_occ4 (fMIndex.h:446-510): the reference sums byte-table look-ups of whole words plus a masked
// partial word; popcounts on the 2-bit fields give the same integers
inline void occ4( const ma_or_index& x, i64 k, u64 cntv[ 4 ] )
{
    if( k == (i64)-1 )
    {
        cntv[ 0 ] = cntv[ 1 ] = cntv[ 2 ] = cntv[ 3 ] = 0;
        return;
    }
    k -= ( k >= x.primary ); // '$' is not stored
    const uint32_t* p = &x.bwt[ ( (u64)k >> 7 ) << 4 ];
    memcpy( cntv, p, 32 );
    p += 8;
    const unsigned within = (unsigned)( (u64)k & 127 ) + 1; // symbols [block start .. k]
    u64 add[ 4 ] = { 0, 0, 0, 0 };
    for( unsigned w = 0; w * 16 < within; w++ )
        wordCounts( p[ w ], within - w * 16, add );
    for( int c = 0; c < 4; c++ )
        cntv[ c ] += add[ c ];
}

// FMIndex::extend_backward (fMIndex.cpp:21-101) with bwt_2occ4's forced two-call branch (fMIndex.h:671-690)
inline void extendBackward( const ma_or_index& x, const i64 ik[ 3 ], uint8_t c, i64 ok[ 3 ] )
{
    cnt( 0 ); // every call counts, also the c >= 4 early-out that touches no block
    if( c >= 4 )
    {
        ok[ 0 ] = ok[ 1 ] = ok[ 2 ] = 0;
        return;
    }
    const i64 start = ik[ 0 ], rc = ik[ 1 ], size = ik[ 2 ], end = start + size;
    u64 cntk[ 4 ], cntl[ 4 ], cnts[ 4 ];
    occ4( x, start - 1, cntk );
    occ4( x, end - 1, cntl );
    {
        i64 k = start - 1, l = end - 1;
        i64 kb = k == -1 ? -1 : ( ( k - ( k >= x.primary ) ) >> 7 );
        i64 lb = l == -1 ? -1 : ( ( l - ( l >= x.primary ) ) >> 7 );
        cnt( 1, ( kb >= 0 ) + ( lb >= 0 && lb != kb ) );
    }
    for( int i = 0; i < 4; i++ )
        cnts[ i ] = cntl[ i ] - cntk[ i ];
    u64 cntk2[ 4 ];
    cntk2[ 0 ] = (u64)rc;
    if( start <= x.primary && end > x.primary )
        cntk2[ 0 ]++;
    for( int i = 1; i < 4; i++ )
        cntk2[ i ] = cntk2[ i - 1 ] + cnts[ 3 - ( i - 1 ) ];
    ok[ 0 ] = (i64)( x.L2[ c ] + cntk[ c ] + 1 );
    ok[ 1 ] = (i64)cntk2[ 3 - c ];
    ok[ 2 ] = (i64)cnts[ c ];
}

// init_interval (fMIndex.h:768-775); complement of c>=4 is 5 -> never called with c>=4 by seeding
inline void initInterval( const ma_or_index& x, uint8_t c, i64 ik[ 3 ] )
{
    ik[ 0 ] = (i64)x.L2[ c ] + 1;
    ik[ 1 ] = (i64)x.L2[ 3 - c ] + 1;
    ik[ 2 ] = (i64)( x.L2[ c + 1 ] - x.L2[ c ] );
}

// bwt_B0 (fMIndex.h:268-269)
inline uint8_t bwtB0( const ma_or_index& x, i64 k )
{
    return ( x.bwt[ ( ( (u64)k >> 7 ) << 4 ) + 8 + ( ( (u64)k & 127 ) >> 4 ) ] >> ( ( ~(u64)k & 15 ) << 1 ) ) & 3;
}

// bwt_occ (fMIndex.h:283-323)
inline i64 bwtOcc( const ma_or_index& x, i64 k, uint8_t c )
{
    if( k == (i64)x.n )
        return (i64)( x.L2[ c + 1 ] - x.L2[ c ] );
    if( k == (i64)-1 )
        return 0;
    u64 cntv[ 4 ];
    occ4( x, k, cntv ); // same block, same masking; only the c-th counter is used
    return (i64)cntv[ c ];
}

// bwt_invPsi (fMIndex.h:329-343)
inline i64 invPsi( const ma_or_index& x, i64 k )
{
    i64 xx = k - ( k > x.primary );
    uint8_t c = bwtB0( x, xx );
    xx = (i64)x.L2[ c ] + bwtOcc( x, k, c );
    return k == x.primary ? 0 : xx;
}

// bwt_sa (fMIndex.h:788-814)
inline i64 bwtSa( const ma_or_index& x, i64 k )
{
    i64 s = 0;
    while( k & 31 )
    {
        ++s;
        k = invPsi( x, k );
        cnt( 2 );
    }
    cnt( 3 );
    return s + x.sa[ k / 32 ];
}

// ---- Pack helpers ----
// uiSequenceIdForPosition (pack.h:933-990) incl. its exact binary search
inline i64 seqIdForPosition( const ma_or_index& x, u64 pos )
{
    const i64 iAbs = pos >= x.F ? (i64)( x.n - ( pos + 1 ) ) : (i64)pos;
    u64 l = 0, m = 0, r = x.cstart.size( );
    while( l < r )
    {
        m = ( l + r ) / 2;
        if( iAbs >= (i64)x.cstart[ m ] )
        {
            if( m == x.cstart.size( ) - 1 )
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
inline bool onRev( const ma_or_index& x, u64 p )
{
    return p >= x.F;
}
inline u64 toRev( const ma_or_index& x, u64 p ) // uiPositionToReverseStrand (pack.h:924-927)
{
    return x.n - ( p + 1 );
}
inline i64 seqIdOrRev( const ma_or_index& x, u64 p ) // pack.h:1029-1034
{
    if( onRev( x, p ) )
        return seqIdForPosition( x, toRev( x, p ) ) * 2 + 1;
    return seqIdForPosition( x, p ) * 2;
}
inline u64 endOfSeqOrRev( const ma_or_index& x, i64 id ) // pack.h:1040-1045
{
    if( id % 2 == 1 )
        return toRev( x, x.cstart[ id / 2 ] ) - 1;
    return x.cstart[ id / 2 ] + x.clen[ id / 2 ];
}
inline u64 startOfSeqOrRev( const ma_or_index& x, i64 id ) // pack.h:1047-1052
{
    if( id % 2 == 1 )
        return toRev( x, x.cstart[ id / 2 ] + x.clen[ id / 2 ] ) + 1;
    return x.cstart[ id / 2 ];
}
inline bool bridging( const ma_or_index& x, u64 b, u64 size ) // pack.h:1072-1087
{
    if( size == 0 )
        return false;
    i64 id = seqIdOrRev( x, b );
    return ( onRev( x, b ) != onRev( x, b + size - 1 ) ) || ( id != seqIdOrRev( x, b + size - 1 ) );
}

// ---------------------------------------------------------------------------------------------
// containers
// ---------------------------------------------------------------------------------------------
struct Seg // Segment (segment.h:31-113): query interval (size = length-1) + SAInterval
{
    u64 start, size;
    i64 sa[ 3 ]; // start, startRevComp, size
    u64 end( ) const
    {
        return start + size;
    }
};
struct Seed // seed.h:34-46
{
    u64 q = 0, len = 0, r = 0;
    uint32_t amb = 0;
    u64 socNt = 0;
    bool fwd = true;
    u64 delta = 0;
    u64 end( ) const
    {
        return q + len;
    }
    u64 endRef( ) const
    {
        return r + len;
    }
};
struct SeedSet
{
    std::vector<Seed> v;
    uint32_t socIndex = 0; // xStats.index_of_strip
};
enum MT
{
    MT_SEED = 0,
    MT_MATCH = 1,
    MT_MISS = 2,
    MT_INS = 3,
    MT_DEL = 4
};
struct Aln // alignment.h:55-84
{
    std::vector<std::pair<int, u64>> data;
    u64 length = 0, bRef = 0, eRef = 0, bQ = 0, eQ = 0;
    i64 score = 0;
    uint32_t socIndex = 0;
    bool secondary = false, supplementary = false;
    double mapq = NAN;
    bool first = false; // AlignmentStatistics::bFirst (seed.h:227)
    int other = -1; // xStats.pOther: index of the mate's alignment in the PairedReads output
};

// ---------------------------------------------------------------------------------------------
// BinarySeeding (binarySeeding.h:55-452, binarySeeding.cpp:32-178)
// ---------------------------------------------------------------------------------------------
struct Ctx
{
    const ma_or_index& x;
    const ma_or_params& P;
};

inline void rcSwap( i64 ik[ 3 ] ) // SAInterval::revComp (fMIndex.h:85-88)
{
    std::swap( ik[ 0 ], ik[ 1 ] );
}
inline bool stopExt( const Ctx& c, const i64 ok[ 3 ], const i64 ik[ 3 ] )
{
    if( ok[ 2 ] <= 0 )
        return true;
    if( ok[ 2 ] <= (i64)(unsigned)c.P.min_ambiguity && ik[ 2 ] <= (i64)(unsigned)c.P.max_ambiguity )
        return true;
    return false;
}

// maximallySpanningExtension (binarySeeding.h:55-252). Returns covered interval (start, size) where
// end() = start+size follows the reference's (inclusive-last-index) convention.
static void maxSpanExt( const Ctx& c, u64 center, const uint8_t* q, u64 qlen, std::vector<Seg>& out, u64& covS,
                        u64& covSize )
{
    if( q[ center ] >= 4 )
    {
        covS = center;
        covSize = 1;
        return;
    }
    i64 ik[ 3 ], ok[ 3 ];
    initInterval( c.x, 3 - q[ center ], ik );
    if( ik[ 2 ] == 0 )
    {
        covS = center;
        covSize = 1;
        return;
    }
    u64 end = center;
    for( u64 i = center + 1; i < qlen; i++ )
    {
        extendBackward( c.x, ik, q[ i ] < 4 ? 3 - q[ i ] : 5, ok );
        if( stopExt( c, ok, ik ) )
            break;
        end = i;
        memcpy( ik, ok, sizeof( ik ) );
    }
    rcSwap( ik );
    u64 start = center;
    if( center > 0 )
        for( u64 i = center - 1;; i-- )
        {
            extendBackward( c.x, ik, q[ i ], ok );
            if( stopExt( c, ok, ik ) )
                break;
            start = i;
            memcpy( ik, ok, sizeof( ik ) );
            if( i == 0 )
                break;
        }
    out.push_back( Seg{ start, end - start, { ik[ 0 ], ik[ 1 ], ik[ 2 ] } } );
    // other way
    initInterval( c.x, q[ center ], ik );
    start = center;
    if( center > 0 )
        for( u64 i = center - 1;; i-- )
        {
            extendBackward( c.x, ik, q[ i ], ok );
            if( stopExt( c, ok, ik ) )
                break;
            start = i;
            memcpy( ik, ok, sizeof( ik ) );
            if( i == 0 )
                break;
        }
    rcSwap( ik );
    end = center;
    for( u64 i = center + 1; i < qlen; i++ )
    {
        extendBackward( c.x, ik, q[ i ] < 4 ? 3 - q[ i ] : 5, ok );
        if( stopExt( c, ok, ik ) )
            break;
        end = i;
        memcpy( ik, ok, sizeof( ik ) );
    }
    const Seg& b = out.back( );
    if( b.start == start && b.end( ) == end )
    {
        covS = b.start;
        covSize = b.size;
        return;
    }
    rcSwap( ik );
    out.push_back( Seg{ start, end - start, { ik[ 0 ], ik[ 1 ], ik[ 2 ] } } );
    const Seg& prev = out[ out.size( ) - 2 ];
    const Seg& last = out.back( );
    // ret(center,0); start(x) keeps end; end(x) sets size (geom.h:77-108)
    u64 s = last.start < prev.start ? last.start : prev.start;
    u64 e = last.end( ) > prev.end( ) ? last.end( ) : prev.end( );
    covS = s;
    covSize = e - s;
}

// smemExtension (binarySeeding.h:261-452)
static void smemExt( const Ctx& c, u64 center, const uint8_t* q, u64 qlen, std::vector<Seg>& out, u64& covS,
                     u64& covSize )
{
    u64 retS = center, retE = center; // ret(center, 0)
    if( q[ center ] >= 4 )
    {
        covS = center;
        covSize = 1;
        return;
    }
    i64 ik[ 3 ], ok[ 3 ];
    initInterval( c.x, 3 - q[ center ], ik );
    std::vector<Seg> curr, prev;
    for( u64 i = center + 1; i < qlen; i++ )
    {
        extendBackward( c.x, ik, q[ i ] < 4 ? 3 - q[ i ] : 5, ok );
        if( ok[ 2 ] != ik[ 2 ] )
            curr.push_back( Seg{ center, i - center - 1, { ik[ 1 ], ik[ 0 ], ik[ 2 ] } } );
        if( i == qlen - 1 && ok[ 2 ] != 0 )
            curr.push_back( Seg{ center, i - center, { ok[ 1 ], ok[ 0 ], ok[ 2 ] } } );
        if( ok[ 2 ] == 0 )
            break;
        if( ok[ 2 ] <= (i64)(unsigned)c.P.min_ambiguity && ik[ 2 ] <= (i64)(unsigned)c.P.max_ambiguity )
            break;
        memcpy( ik, ok, sizeof( ik ) );
        retE = i; // ret.end(i)
    }
    std::reverse( curr.begin( ), curr.end( ) );
    std::vector<Seg>*pPrev = &curr, *pCurr = &prev;
    if( center != 0 )
    {
        for( u64 i = center - 1;; i-- )
        {
            bool bHaveOne = false;
            for( Seg& s : *pPrev )
            {
                extendBackward( c.x, s.sa, q[ i ], ok );
                if( ok[ 2 ] <= (i64)(unsigned)c.P.min_ambiguity && !bHaveOne )
                {
                    out.push_back( s );
                    bHaveOne = true;
                }
                else if( ok[ 2 ] > (i64)(unsigned)c.P.min_ambiguity ||
                         ( ok[ 2 ] > 0 && s.size >= (u64)(unsigned)c.P.max_ambiguity ) )
                    pCurr->push_back( Seg{ i, s.size + 1, { ok[ 0 ], ok[ 1 ], ok[ 2 ] } } );
            }
            std::swap( pPrev, pCurr );
            pCurr->clear( );
            if( pPrev->empty( ) )
                break;
            retS = i; // ret.start(i) keeps end
            if( i == 0 )
                break;
        }
    }
    if( !pPrev->empty( ) )
        out.push_back( pPrev->front( ) );
    covS = retS;
    covSize = retE - retS;
}

// procesInterval (binarySeeding.cpp:32-84); recursion on the left part, iteration on the right
static void procesInterval( const Ctx& c, u64 aS, u64 aSize, const uint8_t* q, u64 qlen, std::vector<Seg>& out )
{
    while( true )
    {
        u64 cS, cSize;
        const u64 center = aS + aSize / 2;
        if( c.P.seeding_technique == 0 )
            maxSpanExt( c, center, q, qlen, out, cS, cSize );
        else
            smemExt( c, center, q, qlen, out, cS, cSize );
        const u64 cE = cS + cSize, aE = aS + aSize;
        if( cS != 0 && aS + 1 < cS )
            procesInterval( c, aS, cS - aS, q, qlen, out );
        if( aE > cE + 1 )
        {
            aS = cE; // set(start,size): start(x) keeps end then size(x) overrides
            aSize = aE - cE;
        }
        else
            break;
    }
}

// memExtension (binarySeeding.h:460-537): every maximal exact match, found by extending rightwards from every query
// position and checking the rows that drop out of the interval for left-maximality.  The intervals it hands on carry -1 as
// start of the reverse-complement interval (SAInterval::do_for_difference, fMIndex.h:123-150); extend_backward's start and
// size do not depend on it.
static void memsExt( const Ctx& c, const uint8_t* q, u64 qlen, std::vector<Seg>& out )
{
    const i64 minAmb = (i64)(unsigned)c.P.min_ambiguity, maxAmb = (i64)(unsigned)c.P.max_ambiguity;
    const u64 minSeed = (u64)(unsigned)c.P.min_seed_len;
    for( u64 i = 0; i < qlen; i++ )
    {
        if( q[ i ] >= 4 )
            continue;
        i64 ik[ 3 ];
        initInterval( c.x, (uint8_t)( 3 - q[ i ] ), ik );
        for( u64 j = i + 1; j <= qlen && ik[ 2 ] > minAmb; j++ )
        {
            i64 ok[ 3 ] = { 0, -1, 0 };
            if( j < qlen && q[ j ] < 4 )
                extendBackward( c.x, ik, (uint8_t)( 3 - q[ j ] ), ok );
            if( j - i - 1 > minSeed && ok[ 2 ] < ik[ 2 ] && ik[ 2 ] < maxAmb )
            {
                // ik.revComp( ).do_for_difference( ok.revComp( ), ... )
                const i64 aS = ik[ 1 ], aE = ik[ 1 ] + ik[ 2 ], bS = ok[ 1 ], bE = ok[ 1 ] + ok[ 2 ];
                const i64 uiY = std::min( bS, aE ), uiX = std::max( bE, aS );
                i64 part[ 2 ][ 2 ];
                int np = 0;
                if( aS < uiY )
                    part[ np ][ 0 ] = aS, part[ np ][ 1 ] = uiY - aS, np++;
                if( uiX < aE )
                    part[ np ][ 0 ] = uiX, part[ np ][ 1 ] = aE - uiX, np++;
                for( int p = 0; p < np; p++ )
                {
                    const i64 dS = part[ p ][ 0 ], dN = part[ p ][ 1 ];
                    i64 xd[ 3 ] = { dS, -1, dN }, xe[ 3 ] = { 0, -1, 0 };
                    if( i > 0 )
                        extendBackward( c.x, xd, q[ i - 1 ], xe );
                    if( xe[ 2 ] == 0 )
                        out.push_back( Seg{ i, j - i - 1, dS, -1, dN } );
                    else if( xe[ 2 ] < dN )
                    {
                        i64 kLast = dS;
                        for( i64 k = dS; k <= dS + dN; k++ )
                        {
                            bool cut = k == dS + dN;
                            if( !cut )
                            {
                                i64 xr[ 3 ] = { k, -1, 1 }, xo[ 3 ];
                                extendBackward( c.x, xr, q[ i - 1 ], xo );
                                cut = xo[ 2 ] != 0;
                            }
                            if( cut )
                            {
                                if( k > kLast )
                                    out.push_back( Seg{ i, j - i - 1, kLast, -1, k - kLast } );
                                kLast = k + 1;
                            }
                        }
                    }
                }
            }
            ik[ 0 ] = ok[ 0 ], ik[ 1 ] = ok[ 1 ], ik[ 2 ] = ok[ 2 ];
        }
    }
}

// BinarySeeding::execute (binarySeeding.cpp:86-178), numSeedsLarger (segment.h:278-289)
static void seedRead( const Ctx& c, const uint8_t* q, u64 qlen, std::vector<Seg>& out )
{
    out.clear( );
    if( qlen == 0 )
        return;
    if( c.P.seeding_technique == 2 )
        memsExt( c, q, qlen, out );
    else
        procesInterval( c, 0, qlen, q, qlen, out );
    if( !c.P.disable_heuristics && c.P.min_seed_size_drop != 0 )
    {
        size_t sum = 0;
        for( const Seg& s : out )
            sum += (size_t)s.size / (size_t)c.P.min_seed_size_drop;
        if( (double)sum < c.P.rel_min_seed_size_amount * (double)qlen && c.P.genome_size_disable < c.x.n )
            out.clear( );
    }
}

// ---------------------------------------------------------------------------------------------
// ExtractSeeds (segment.h:89-113,316-369; stripOfConsideration.h:41-53,97-157)
// ---------------------------------------------------------------------------------------------
static void extractSeeds( const Ctx& c, const std::vector<Seg>& segs, u64 qlen, std::vector<Seed>& seeds )
{
    seeds.clear( );
    const u64 minLen = (u64)(unsigned)c.P.min_seed_len;
    const u64 maxAmb = (u64)(unsigned)c.P.max_ambiguity;
    for( const Seg& s : segs )
    {
        if( s.size < minLen )
            continue;
        if( s.sa[ 2 ] > (i64)maxAmb && maxAmb != 0 )
            continue; // bSkip == true (segment.h:365)
        for( i64 p = s.sa[ 0 ]; p < s.sa[ 0 ] + s.sa[ 2 ]; p++ )
        {
            u64 r = (u64)bwtSa( c.x, p );
            bool fwd = r < c.x.n / 2;
            if( !fwd )
                r = c.x.n - r - 1;
            Seed sd;
            sd.q = s.start;
            sd.len = s.size + 1;
            sd.r = r;
            sd.amb = (uint32_t)s.sa[ 2 ];
            sd.fwd = fwd;
            // setDeltaOfSeed, rectangular mode (bSplitStrands == false)
            sd.delta = r + ( qlen - sd.q );
            u64 contig = (u64)seqIdForPosition( c.x, r );
            sd.delta += ( qlen + 1 ) * contig;
            seeds.push_back( sd );
        }
    }
}

// ---------------------------------------------------------------------------------------------
// StripOfConsiderationSeeds::execute (stripOfConsideration.cpp:12-161) + SoCPriorityQueue (soc.h)
// ---------------------------------------------------------------------------------------------
struct SoCOrder // soc.h:26-90
{
    u64 accLen = 0;
    uint32_t amb = 0, cnt = 0;
    void add( const Seed& s )
    {
        amb += s.amb;
        cnt++;
        accLen += s.len;
    }
    void sub( const Seed& s )
    {
        amb -= s.amb;
        accLen -= s.len;
        cnt--;
    }
    bool operator<( const SoCOrder& o ) const
    {
        if( accLen == o.accLen )
            return amb > o.amb;
        return accLen < o.accLen;
    }
};
struct SoCEntry
{
    SoCOrder sc;
    size_t b, e; // iterator pair as indices into the seed array
};
struct SoCQueue
{
    std::vector<Seed>* pSeeds = nullptr;
    std::vector<SoCEntry> vMaxima;
    uint32_t uiSoCIndex = 0;
    static bool heapOrder( const SoCEntry& a, const SoCEntry& b )
    {
        return a.sc < b.sc;
    }
    bool empty( ) const
    {
        return vMaxima.empty( );
    }
    static SoCOrder sumRange( const std::vector<Seed>& s, size_t b, size_t e )
    {
        SoCOrder o;
        for( size_t i = b; i < e; i++ )
            o.add( s[ i ] );
        return o;
    }
    // push_back_no_overlap (soc.h:362-404); adjustScore's two branches are value-identical
    void pushBackNoOverlap( SoCOrder cur, size_t itS, size_t itE, u64 minScore )
    {
        std::vector<Seed>& s = *pSeeds;
        while( !vMaxima.empty( ) && vMaxima.back( ).e > itS )
        {
            if( vMaxima.back( ).sc < cur )
            {
                vMaxima.back( ).sc = sumRange( s, vMaxima.back( ).b, itS );
                vMaxima.back( ).e = itS;
                if( vMaxima.back( ).sc.accLen < minScore || vMaxima.back( ).sc.accLen == 0 )
                    vMaxima.pop_back( );
            }
            else
            {
                cur = sumRange( s, vMaxima.back( ).e, itE );
                itS = vMaxima.back( ).e;
                if( cur.accLen < minScore || cur.accLen == 0 )
                    return;
            }
        }
        vMaxima.push_back( SoCEntry{ cur, itS, itE } );
    }
    // rectangularSoC (soc.h:196-231)
    void rectangular( )
    {
        std::vector<Seed>& s = *pSeeds;
        std::vector<std::pair<u64, u64>> mm;
        for( auto& t : vMaxima )
        {
            mm.emplace_back( s[ t.b ].r, s[ t.b ].r );
            for( size_t i = t.b; i != t.e; i++ )
            {
                mm.back( ).first = std::min( mm.back( ).first, s[ i ].r );
                mm.back( ).second = std::max( mm.back( ).second, s[ i ].r );
            }
        }
        std::sort( s.begin( ), s.end( ), []( const Seed& a, const Seed& b ) { return a.r < b.r; } );
        vMaxima.clear( );
        for( auto& p : mm )
        {
            SoCEntry e;
            e.b = std::lower_bound( s.begin( ), s.end( ), p.first,
                                    []( const Seed& x, u64 pos ) { return x.r < pos; } ) -
                  s.begin( );
            size_t it = e.b;
            while( it != s.size( ) && s[ it ].r <= p.second )
            {
                e.sc.add( s[ it ] );
                it++;
            }
            e.e = it;
            vMaxima.push_back( e );
        }
    }
    // pop (soc.h:240-284)
    SeedSet pop( )
    {
        std::vector<Seed>& s = *pSeeds;
        SeedSet ret;
        ret.socIndex = uiSoCIndex++;
        size_t it = vMaxima.front( ).b, e = vMaxima.front( ).e;
        while( it != s.size( ) && it != e )
        {
            s[ it ].socNt = vMaxima.front( ).sc.accLen;
            ret.v.push_back( s[ it++ ] );
        }
        std::pop_heap( vMaxima.begin( ), vMaxima.end( ), heapOrder );
        vMaxima.pop_back( );
        return ret;
    }
};

static void socSweep( const Ctx& c, std::vector<Seed>& seeds, u64 qlen, SoCQueue& Q )
{
    Q.pSeeds = &seeds;
    Q.vMaxima.clear( );
    Q.uiSoCIndex = 0;
    if( seeds.empty( ) )
        return;
    double fMinLen = std::max( (double)c.P.harm_score_min_rel * qlen, (double)(size_t)c.P.harm_score_min );
    if( c.P.genome_size_disable >= c.x.n )
        fMinLen = 0;
    // getStripSize (stripOfConsideration.h:55-61): int * u64 arithmetic in u64
    u64 strip = c.P.soc_width != 0 ? (u64)c.P.soc_width
                                   : ( (u64)c.P.match * qlen - (u64)c.P.gap ) / (u64)c.P.extend;
    std::sort( seeds.begin( ), seeds.end( ), []( const Seed& a, const Seed& b ) { return a.delta < b.delta; } );
    SoCOrder cur;
    size_t S = 0, E = 0;
    const size_t N = seeds.size( );
    while( E != N && S != N )
    {
        const i64 cidS = seqIdForPosition( c.x, seeds[ S ].r );
        while( E != N && seeds[ S ].delta + strip >= seeds[ E ].delta &&
               cidS == seqIdForPosition( c.x, seeds[ E ].r ) )
        {
            cur.add( seeds[ E ] );
            E++;
        }
        if( (double)cur.accLen >= fMinLen )
            Q.pushBackNoOverlap( cur, S, E, (u64)fMinLen );
        cur.sub( seeds[ S ] );
        S++;
    }
    std::make_heap( Q.vMaxima.begin( ), Q.vMaxima.end( ), SoCQueue::heapOrder );
    Q.rectangular( ); // xRectangularSoc default true (parameter.h:715-718)
}

// ---------------------------------------------------------------------------------------------
// glibc rand(): TYPE_3 additive feedback generator r[i] = r[i-3] + r[i-31] (glibc stdlib/random_r.c).
// glibc is outside /root/reference; this is its published algorithm, pinned against libc in tests.
// ---------------------------------------------------------------------------------------------
struct GlibcRand
{
    uint32_t r[ 34 ];
    int f = 3, b = 0; // indices into r[1..31]-window; implemented as ring over 31 words
    uint32_t ring[ 31 ];
    void seed( uint32_t s )
    {
        if( s == 0 )
            s = 1;
        int32_t word = (int32_t)s;
        ring[ 0 ] = (uint32_t)word;
        for( int i = 1; i < 31; i++ )
        {
            long hi = word / 127773;
            long lo = word % 127773;
            word = (int32_t)( 16807 * lo - 2836 * hi );
            if( word < 0 )
                word += 2147483647;
            ring[ i ] = (uint32_t)word;
        }
        f = 3;
        b = 0;
        for( int i = 0; i < 310; i++ )
            next( );
    }
    int32_t next( )
    {
        ring[ f ] += ring[ b ];
        uint32_t res = ring[ f ] >> 1;
        if( ++f >= 31 )
            f = 0;
        if( ++b >= 31 )
            b = 0;
        return (int32_t)res;
    }
};
} // namespace

extern "C" void ma_or_srand( uint32_t seed, uint32_t state[ 35 ] )
{
    GlibcRand g;
    g.seed( seed );
    memcpy( state, g.ring, sizeof( g.ring ) );
    state[ 31 ] = (uint32_t)g.f;
    state[ 32 ] = (uint32_t)g.b;
}
extern "C" int32_t ma_or_rand( uint32_t state[ 35 ] )
{
    GlibcRand g;
    memcpy( g.ring, state, sizeof( g.ring ) );
    g.f = (int)state[ 31 ];
    g.b = (int)state[ 32 ];
    int32_t r = g.next( );
    memcpy( state, g.ring, sizeof( g.ring ) );
    state[ 31 ] = (uint32_t)g.f;
    state[ 32 ] = (uint32_t)g.b;
    return r;
}

#include "ma_oracle_harm.inc"
#include "ma_oracle_ksw.inc"
#include "ma_oracle_nw.inc"
#include "ma_oracle_f4.inc"
#include "ma_oracle_api.inc"
