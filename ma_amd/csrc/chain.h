// chain.h -- seed chaining on the device: StripOfConsiderationSeeds::execute
// (stripOfConsideration.cpp:12-161, soc.h:26-420) followed by Harmonization::execute
// (harmonization.cpp:14-555, harmonization.h:82-89) incl. the RANSAC line fit
// (ransac.cpp:67-165, sac_model_line.cpp:49-130, lin_regres.h:8-137, test_ransac.h:21-76).
// One read per lane; all per-read working sets live in HBM scratch carved by the read's seed offset.
// Unstable-sort tie orders follow libstdc++ through stdsort.h; RANSAC draws follow glibc's TYPE_3
// rand() starting from srand(seed) state at the beginning of every read (parity mode of SURVEY 8(c)).
#pragma once
#include "fm_device.h"
#include "stdsort.h"
#include <math.h>

namespace ma
{
// phase cycle counters of the diagnostics build (-DMA_CHAIN_PROF, tools/chain_prof.py); run with MA_LANES_PER_WAVE=1 so
// that a lane's clock does not include the turns of the other lanes of its wave
#if defined( MA_CHAIN_PROF ) && defined( __HIPCC__ )
static __device__ unsigned long long g_chain_prof[ 16 ];
#endif
#if defined( MA_CHAIN_PROF ) && defined( __HIP_DEVICE_COMPILE__ )
#define CH_T( v ) const unsigned long long v = clock64( )
#define CH_ADD( i, a, b ) atomicAdd( &g_chain_prof[ i ], ( b ) - ( a ) )
#else
#define CH_T( v )
#define CH_ADD( i, a, b )
#endif

struct ChainParams
{
    u32 max_num_soc, min_num_soc;
    u32 harm_score_min;
    u32 max_score_lookahead;
    u32 switch_qlen;
    u32 min_delta_dist;
    u32 sv_penalty;
    u32 match, gap, extend;
    u32 disable_heuristics;
    u32 soc_width;
    u64 genome_size_disable;
    double harm_score_min_rel;
    double soc_score_decrease_tol;
    double score_diff_tol;
    double max_delta_dist;
    u32 rng_ring[ 31 ]; // glibc random() state after srand(seed) (310 discards done)
    u32 libm_probe; // ma_params::libm_probe: 0 = off (the product), else every libm result is nudged (see LibmProbe)
};

struct SoCEntry // tuple<SoCOrder, it, it> (soc.h:26-90, 192)
{
    u64 accLen;
    u32 amb, cnt;
    u32 b, e;
};
struct Shadow // tuple<Seeds::iterator, nucSeqIndex, nucSeqIndex> (harmonization.cpp:182-189)
{
    u64 a, b;
    u32 seed;
    u32 pad;
};
struct RefMinMax
{
    u64 lo, hi;
};
struct HSet // one harmonized seed set of a read
{
    u64 off; // into the hseed pool
    u32 cnt;
    u32 soc; // xStats.index_of_strip
};

// Per-read scratch, carved from batch-wide arrays with the read's seed offset.
struct ChainScratch
{
    ma_seed* work; // n: working copy of the read's seeds (sorted in place)
    SoCEntry* maxima; // n
    RefMinMax* mm; // n
    ma_seed* setA; // n: popped SoC (forward part after the strand split)
    ma_seed* setB; // n: reverse-strand part
    ma_seed* outA; // n: harmonizeOne output
    Shadow* sh1; // n
    Shadow* sh2; // n
    double* vX; // 3n
    double* vY; // 3n
    double* med; // 3n
    i32* inl; // 3n
    i32* best; // 3n
};

struct GlibcRand // glibc stdlib/random_r.c, TYPE_3: r[i] = r[i-3] + r[i-31], output >> 1
{
    u32 ring[ 31 ];
    i32 f, b;
    MA_HD void init( const u32* st )
    {
        for( int i = 0; i < 31; i++ )
            ring[ i ] = st[ i ];
        f = 3;
        b = 0;
    }
    MA_HD i32 next( )
    {
        ring[ f ] += ring[ b ];
        const u32 res = ring[ f ] >> 1;
        if( ++f >= 31 )
            f = 0;
        if( ++b >= 31 )
            b = 0;
        return (i32)res;
    }
};

// Sensitivity probe for the four libm functions the stage decides with (tan, sin, atan, log): the reference evaluates them
// with glibc, the device with ocml, and the two differ in the last bit for ~13 % of the arguments
// (tests/test_gpu_round2.py).  With mode != 0 every result is moved by one ulp -- 1: up, 2: down, >= 3: up / same / down
// by a hash of the call count and the mode -- so that a test can assert that no decision of a whole corpus depends on the
// last bit of these functions.  mode 0 (always, outside that test) returns the value untouched.
struct LibmProbe
{
    u32 mode, calls;
    MA_HD double operator( )( double v )
    {
        if( mode == 0 || !( v == v ) || v == 0.0 || v - v != 0.0 )
            return v;
        calls++;
        u32 h = calls * 2654435761u + mode * 0x9E3779B9u;
        h ^= h >> 15;
        h *= 0x85EBCA6Bu;
        h ^= h >> 13;
        const i32 d = mode == 1 ? 1 : ( mode == 2 ? -1 : (i32)( h % 3u ) - 1 );
        union
        {
            double f;
            i64 b;
        } u;
        u.f = v;
        u.b += v > 0 ? d : -d;
        return u.f;
    }
};

MA_HD bool soc_less( const SoCEntry& a, const SoCEntry& b ) // SoCOrder::operator< (soc.h:71-76)
{
    if( a.accLen == b.accLen )
        return a.amb > b.amb;
    return a.accLen < b.accLen;
}
struct SoCHeapOrder
{
    MA_HD bool operator( )( const SoCEntry& a, const SoCEntry& b ) const
    {
        return soc_less( a, b );
    }
};
struct SeedByDelta
{
    MA_HD bool operator( )( const ma_seed& a, const ma_seed& b ) const
    {
        return (u64)a.delta < (u64)b.delta;
    }
};
struct SeedByRef
{
    MA_HD bool operator( )( const ma_seed& a, const ma_seed& b ) const
    {
        return (u64)a.r_start < (u64)b.r_start;
    }
};
struct SeedByRefQ
{
    MA_HD bool operator( )( const ma_seed& a, const ma_seed& b ) const
    {
        if( a.r_start == b.r_start )
            return (u64)a.q_start < (u64)b.q_start;
        return (u64)a.r_start < (u64)b.r_start;
    }
};
struct ShadowOrder
{
    MA_HD bool operator( )( const Shadow& xA, const Shadow& xB ) const
    {
        if( xA.a == xB.a )
            return xA.b > xB.b;
        return xA.a < xB.a;
    }
};
struct DoubleLess
{
    MA_HD bool operator( )( double a, double b ) const
    {
        return a < b;
    }
};

MA_HD void soc_sum( const ma_seed* s, u32 b, u32 e, SoCEntry& o )
{
    o.accLen = 0;
    o.amb = 0;
    o.cnt = 0;
    for( u32 i = b; i < e; i++ )
    {
        o.amb += s[ i ].ambiguity;
        o.cnt++;
        o.accLen += (u64)s[ i ].len;
    }
}

// Prefix sums of (len, ambiguity) over the seeds in sweep order: the sums of the reference's loops (soc.h:362-404 re-adds a
// strip's seeds whenever two strips are cut against each other) as differences -- the same integers.
struct SoCPrefix
{
    const u64* len = nullptr; // n + 1 entries
    const u32* amb = nullptr;
};
MA_HD void soc_sum( const ma_seed* s, const SoCPrefix& pre, u32 b, u32 e, SoCEntry& o )
{
    if( pre.len == nullptr || e < b )
    {
        soc_sum( s, b, e, o );
        return;
    }
    o.accLen = pre.len[ e ] - pre.len[ b ];
    o.amb = pre.amb[ e ] - pre.amb[ b ];
    o.cnt = e - b;
}
// push_back_no_overlap (soc.h:362-404)
MA_HD void soc_push_no_overlap( const ma_seed* s, SoCEntry* mx, u32& nmx, SoCEntry cur, u32 itS, u32 itE, u64 minScore,
                                const SoCPrefix& pre = SoCPrefix( ) )
{
    while( nmx > 0 && mx[ nmx - 1 ].e > itS )
    {
        SoCEntry& back = mx[ nmx - 1 ];
        if( soc_less( back, cur ) )
        {
            const u32 bb = back.b;
            soc_sum( s, pre, bb, itS, back );
            back.b = bb;
            back.e = itS;
            if( back.accLen < minScore || back.accLen == 0 )
                nmx--;
        }
        else
        {
            const u32 be = back.e;
            soc_sum( s, pre, be, itE, cur );
            itS = be;
            if( cur.accLen < minScore || cur.accLen == 0 )
                return;
        }
    }
    cur.b = itS;
    cur.e = itE;
    mx[ nmx++ ] = cur;
}

// StripOfConsiderationSeeds::execute; returns number of maxima (heap order, then rectangularSoC)
// Sorting 40-byte seeds by one 64-bit key: the sort runs on (key, index) pairs and the seeds are permuted once.  The
// permutation is the one std::sort produces on the seeds themselves -- its moves depend on comparison results only.
struct KeyIdx
{
    u64 key;
    u32 idx, pad;
};
struct KeyIdxLess
{
    MA_HD bool operator( )( const KeyIdx& a, const KeyIdx& b ) const
    {
        return a.key < b.key;
    }
};
template <typename KEY> MA_HD void sort_seeds_by_key( ma_seed* s, u32 n, KeyIdx* ki, ma_seed* tmp, KEY key )
{
    for( u32 i = 0; i < n; i++ )
    {
        ki[ i ].key = key( s[ i ] );
        ki[ i ].idx = i;
    }
    ss::sort( ki, (i64)n, KeyIdxLess( ) );
    for( u32 i = 0; i < n; i++ )
        tmp[ i ] = s[ ki[ i ].idx ];
    for( u32 i = 0; i < n; i++ )
        s[ i ] = tmp[ i ];
}

// seq_id_for_position with the contig of the previous query remembered: neighbouring seeds of a sweep lie on the same
// contig, and the binary search over the contig table is five dependent loads
struct SeqIdCache
{
    i64 lo = 1, hi = 0, id = 0; // [lo, hi) of absolute forward positions with that id
    MA_HD i64 get( const IndexView& X, u64 pos )
    {
        const i64 iAbs = pos >= X.F ? (i64)( X.n - ( pos + 1 ) ) : (i64)pos;
        if( iAbs >= lo && iAbs < hi )
            return id;
        id = seq_id_for_position( X, pos );
        lo = (i64)X.cstart[ id ];
        hi = id + 1 < (i64)X.n_contigs ? (i64)X.cstart[ id + 1 ] : (i64)0x7fffffffffffffffll;
        if( iAbs < lo )
            hi = lo; // below the first contig start: nothing to remember
        return id;
    }
};

// scratch of the keyed sorts (optional): ki1 / ki2 = n (key, index) pairs each, usable while mm / after mm is filled; tmp = n seeds
// The sweep in two halves, so that a long read's two big sorts can run between them as wave-cooperative kernels
// (wave_sort.h, k_sort_seeds_wave): soc_windows = [sort by delta] + window sweep + make_heap + reference rectangles,
// soc_rebuild = [sort by reference position] + the strips rebuilt over the re-sorted seeds.
MA_HD u32 soc_windows( const IndexView& X, const ChainParams& P, ma_seed* s, u32 n, u32 qlen, SoCEntry* mx, RefMinMax* mm, ma_seed* tmp,
                       bool sortedByDelta, u64* prefix = nullptr );
MA_HD void soc_rebuild( ma_seed* s, u32 n, SoCEntry* mx, const RefMinMax* mm, u32 nmx, KeyIdx* ki2, ma_seed* tmp, bool sortedByRef );
MA_HD u32 soc_sweep( const IndexView& X, const ChainParams& P, ma_seed* s, u32 n, u32 qlen, SoCEntry* mx, RefMinMax* mm,
                     KeyIdx* ki2 = nullptr, ma_seed* tmp = nullptr )
{
    const u32 nmx = soc_windows( X, P, s, n, qlen, mx, mm, tmp, false );
    if( n != 0 )
        soc_rebuild( s, n, mx, mm, nmx, ki2, tmp, false );
    return nmx;
}
// prefix: 12 * (n + 1) bytes of scratch (or null; may be `tmp`, which only the sort uses): the window sums and the re-sums of
// push_back_no_overlap come out of prefix sums (50 kb reads: k_soc_windows 36.6 -> 19.9 ms)
MA_HD u32 soc_windows( const IndexView& X, const ChainParams& P, ma_seed* s, u32 n, u32 qlen, SoCEntry* mx, RefMinMax* mm, ma_seed* tmp,
                       bool sortedByDelta, u64* prefix )
{
    if( n == 0 )
        return 0;
    double fMinLen = mmax( (double)P.harm_score_min_rel * (double)(u64)qlen, (double)(u64)P.harm_score_min );
    if( P.genome_size_disable >= X.n )
        fMinLen = 0;
    const u64 strip = P.soc_width != 0 ? (u64)P.soc_width : ( (u64)P.match * (u64)qlen - (u64)P.gap ) / (u64)P.extend;
    CH_T( c0 );
    if( sortedByDelta )
        ;
    else if( tmp && n >= 64 ) // short reads have a handful of seeds: not worth the two extra passes
        sort_seeds_by_key( s, n, (KeyIdx*)mm, tmp, []( const ma_seed& x ) { return (u64)x.delta; } ); // mm is free until the rectangles
    else
        ss::sort( s, (i64)n, SeedByDelta( ) );
    CH_T( c1 );
    CH_ADD( 0, c0, c1 );
    u32 nmx = 0;
    SoCEntry cur;
    cur.accLen = 0, cur.amb = 0, cur.cnt = 0, cur.b = cur.e = 0;
    u32 S = 0, E = 0;
    SeqIdCache cache; // both ends of the window: they are on the same contig nearly always
    i64 cidE = cache.get( X, (u64)s[ 0 ].r_start );
    SoCPrefix pre;
    if( prefix != nullptr ) // (may be the keyed sort's scratch: that sort is over)
    {
        u64* pl = prefix;
        u32* pa = (u32*)( prefix + n + 1 );
        u64 al = 0;
        u32 aa = 0;
        pl[ 0 ] = 0, pa[ 0 ] = 0;
        for( u32 i = 0; i < n; i++ )
        {
            al += (u64)s[ i ].len, aa += s[ i ].ambiguity;
            pl[ i + 1 ] = al, pa[ i + 1 ] = aa;
        }
        pre.len = pl, pre.amb = pa;
        while( E != n && S != n )
        {
            const i64 cidS = cache.get( X, (u64)s[ S ].r_start );
            const u64 lim = (u64)s[ S ].delta + strip;
            while( E != n && lim >= (u64)s[ E ].delta && cidS == cidE )
            {
                E++;
                if( E != n )
                    cidE = cache.get( X, (u64)s[ E ].r_start );
            }
            // the window holds seed S at least (E == S passes both tests), so E > S: the running sums of the reference
            cur.accLen = pl[ E ] - pl[ S ], cur.amb = pa[ E ] - pa[ S ], cur.cnt = E - S;
            if( (double)cur.accLen >= fMinLen )
                soc_push_no_overlap( s, mx, nmx, cur, S, E, (u64)fMinLen, pre );
            S++;
        }
    }
    else
    while( E != n && S != n )
    {
        const i64 cidS = cache.get( X, (u64)s[ S ].r_start );
        while( E != n && (u64)s[ S ].delta + strip >= (u64)s[ E ].delta && cidS == cidE )
        {
            cur.amb += s[ E ].ambiguity;
            cur.cnt++;
            cur.accLen += (u64)s[ E ].len;
            E++;
            if( E != n )
                cidE = cache.get( X, (u64)s[ E ].r_start );
        }
        if( (double)cur.accLen >= fMinLen )
            soc_push_no_overlap( s, mx, nmx, cur, S, E, (u64)fMinLen );
        cur.amb -= s[ S ].ambiguity;
        cur.accLen -= (u64)s[ S ].len;
        cur.cnt--;
        S++;
    }
    CH_T( c2 );
    CH_ADD( 1, c1, c2 );
    ss::make_heap( mx, (i64)nmx, SoCHeapOrder( ) );
    // rectangularSoC (soc.h:196-231)
    for( u32 k = 0; k < nmx; k++ )
    {
        u64 lo = (u64)s[ mx[ k ].b ].r_start, hi = lo;
        for( u32 i = mx[ k ].b; i != mx[ k ].e; i++ )
        {
            lo = mmin( lo, (u64)s[ i ].r_start );
            hi = mmax( hi, (u64)s[ i ].r_start );
        }
        mm[ k ].lo = lo;
        mm[ k ].hi = hi;
    }
    CH_T( c3 );
    CH_ADD( 2, c2, c3 );
    return nmx;
}
MA_HD void soc_rebuild( ma_seed* s, u32 n, SoCEntry* mx, const RefMinMax* mm, u32 nmx, KeyIdx* ki2, ma_seed* tmp, bool sortedByRef )
{
    CH_T( c3 );
    if( sortedByRef )
        ;
    else if( tmp && ki2 && n >= 64 )
        sort_seeds_by_key( s, n, ki2, tmp, []( const ma_seed& x ) { return (u64)x.r_start; } );
    else
        ss::sort( s, (i64)n, SeedByRef( ) );
    CH_T( c4 );
    CH_ADD( 3, c3, c4 );
    for( u32 k = 0; k < nmx; k++ )
    {
        SoCEntry e;
        e.accLen = 0, e.amb = 0, e.cnt = 0;
        e.b = (u32)ss::lower_bound( s, (i64)n, mm[ k ].lo,
                                    []( const ma_seed& x, u64 pos ) { return (u64)x.r_start < pos; } );
        u32 it = e.b;
        while( it != n && (u64)s[ it ].r_start <= mm[ k ].hi )
        {
            e.amb += s[ it ].ambiguity;
            e.cnt++;
            e.accLen += (u64)s[ it ].len;
            it++;
        }
        e.e = it;
        mx[ k ] = e;
    }
    CH_T( c5 );
    CH_ADD( 4, c4, c5 );
}

// The queue alone (SoCPriorityQueue across the boundary): sweep, then pop() until the heap is empty (soc.h:240-284).
// work[] ends up sorted by reference position; out[k] = k-th popped strip.  Returns the number of strips.
MA_HD u32 soc_dump_read( const IndexView& X, const ChainParams& P, ma_seed* work, u32 n, u32 qlen, SoCEntry* mx, RefMinMax* mm,
                         ma_soc* out, bool heap_layout = false )
{
    u32 nmx = soc_sweep( X, P, work, n, qlen, mx, mm );
    u32 k = 0;
    if( heap_layout )
    {
        // vMaxima as the sweep leaves it (make_heap, then rectangularSoC WITHOUT re-heapifying, stripOfConsideration.cpp:
        // 152-156): a binding that fills the reference's own SoCPriorityQueue with this array gets the reference's pop()
        // order from the reference's pop() (soc.h:240-284)
        for( ; k < nmx; k++ )
        {
            out[ k ].acc_len = mx[ k ].accLen;
            out[ k ].ambiguity = mx[ k ].amb;
            out[ k ].n_seeds = mx[ k ].cnt;
            out[ k ].begin = mx[ k ].b;
            out[ k ].end = mx[ k ].e;
        }
        return k;
    }
    while( nmx > 0 )
    {
        const SoCEntry f = mx[ 0 ];
        out[ k ].acc_len = f.accLen;
        out[ k ].ambiguity = f.amb;
        out[ k ].n_seeds = f.cnt;
        out[ k ].begin = f.b;
        out[ k ].end = f.e;
        k++;
        ss::pop_heap( mx, (i64)nmx, SoCHeapOrder( ) );
        nmx--;
    }
    return k;
}

#define MA_PI_TRUNC 3.14159265 /* harmonization.h:23 (sic) */

MA_HD double delta_distance( const ma_seed& s, const double fAngle, const i64 rStart, LibmProbe& lp ) // harmonization.h:82-89
{
    const double y = (double)(u64)s.r_start + (double)(u64)s.q_start / lp( tan( MA_PI_TRUNC / 2 - fAngle ) );
    const double x = ( y - (double)rStart ) * lp( sin( fAngle ) );
    const double x_1 = (double)(u64)s.q_start / lp( sin( MA_PI_TRUNC / 2 - fAngle ) );
    return fabs( x - x_1 );
}

// k-th smallest of a[0..n) to a[k], everything before it <= a[k] (quickselect, median-of-three pivots)
MA_HD void nth_select( double* a, i64 n, i64 k )
{
    i64 lo = 0, hi = n - 1;
    for( int round = 0; hi > lo; round++ )
    {
        if( hi - lo < 16 || round > 96 )
        {
            if( hi - lo >= 16 )
                ss::sort( a + lo, hi - lo + 1, DoubleLess( ) ); // adversarial input: give up on linear time
            else
                for( i64 i = lo + 1; i <= hi; i++ )
                {
                    const double v = a[ i ];
                    i64 j = i;
                    for( ; j > lo && v < a[ j - 1 ]; j-- )
                        a[ j ] = a[ j - 1 ];
                    a[ j ] = v;
                }
            return;
        }
        const i64 mid = lo + ( hi - lo ) / 2;
        auto order2 = [ & ]( i64 x, i64 y ) {
            if( a[ y ] < a[ x ] )
            {
                const double t = a[ x ];
                a[ x ] = a[ y ];
                a[ y ] = t;
            }
        };
        order2( lo, mid );
        order2( lo, hi );
        order2( mid, hi );
        const double pivot = a[ mid ];
        a[ mid ] = a[ hi - 1 ];
        a[ hi - 1 ] = pivot;
        i64 i = lo, j = hi - 1;
        while( true )
        {
            while( a[ ++i ] < pivot )
                ;
            while( pivot < a[ --j ] )
                ;
            if( i >= j )
                break;
            const double t = a[ i ];
            a[ i ] = a[ j ];
            a[ j ] = t;
        }
        a[ hi - 1 ] = a[ i ];
        a[ i ] = pivot;
        if( k < i )
            hi = i - 1;
        else if( k > i )
            lo = i + 1;
        else
            return;
    }
}

// test_ransac.h:21-40 sorts a private copy and reads the middle element(s): order statistics, whatever the algorithm.
// Selection instead of the full std::sort (the array is scratch here as well); 16 % of k_chain for 50 kb reads.
MA_HD double median_of( double* a, u32 n )
{
    if( n == 0 )
        return 0;
    if( n == 1 )
        return a[ 0 ];
    const u32 k = n / 2;
    nth_select( a, (i64)n, (i64)k );
    if( n % 2 == 0 )
    {
        double below = a[ 0 ]; // a[n/2 - 1] of the sorted array = largest element before position k
        for( u32 i = 1; i < k; i++ )
            below = a[ i ] > below ? a[ i ] : below;
        return ( below + a[ k ] ) / 2;
    }
    return a[ k ];
}

// run_ransac -> (angle, rStart as double); NaNs when no model was found
MA_HD_OUTLINE void run_ransac( const double* X, const double* Y, u32 nPts, double fMAD, GlibcRand& rng, i32* inl, i32* best,
                       double* scratch, double& outAngle, double& outIntercept, LibmProbe& lp )
{
    int iterations = 0;
    int nBest = -2147483647;
    double k = 1.0;
    u32 nBestInl = 0;
    bool haveModel = false;
    while( iterations < k )
    {
        int s0, s1;
        {
            const double trand = (double)nPts / ( 2147483647 + 1.0 );
            int idx = (int)( rng.next( ) * trand );
            s0 = idx;
            int iter = 0;
            do
            {
                idx = (int)( rng.next( ) * trand );
                s1 = idx;
                iter++;
                if( iter > 1000 )
                    break;
                iterations++;
            } while( s1 == s0 );
            iterations--;
        }
        double dH = X[ s0 ] - X[ s1 ];
        double dV = Y[ s0 ] - Y[ s1 ];
        if( dH <= 0 && dV <= 0 )
        {
            dH *= -1;
            dV *= -1;
        }
        double dAngle = -90;
        if( dH > 0 && dV > 0 )
            dAngle = lp( atan( dV / dH ) ) * 180 / 3.141592653589793;
        if( dAngle >= 20 && dAngle <= 70 )
        {
            const double sqrT = fMAD * fMAD;
            u32 nIn = 0;
            const double p3x = X[ s1 ] - X[ s0 ], p3y = Y[ s1 ] - Y[ s0 ], p3z = 0.0;
            for( u32 i = 0; i < nPts; i++ )
            {
                const double p4x = X[ s1 ] - X[ i ], p4y = Y[ s1 ] - Y[ i ], p4z = 0.0;
                const double cx = p4y * p3z - p4z * p3y;
                const double cy = p4z * p3x - p4x * p3z;
                const double cz = p4x * p3y - p4y * p3x;
                const double sqrD = ( cx * cx + cy * cy + cz * cz ) / ( p3x * p3x + p3y * p3y + p3z * p3z );
                if( sqrD < sqrT )
                    inl[ nIn++ ] = (i32)i;
            }
            if( (int)nIn > nBest )
            {
                nBest = (int)nIn;
                for( u32 i = 0; i < nIn; i++ )
                    best[ i ] = inl[ i ];
                nBestInl = nIn;
                haveModel = true;
                const double w = (double)nIn / (double)nPts;
                double pNo = 1 - w * w; // pow(w, 2.0)
                pNo = mmax( 2.220446049250313e-16, pNo );
                pNo = mmin( 1 - 2.220446049250313e-16, pNo );
                k = lp( log( 1 - 0.99 ) ) / lp( log( pNo ) );
            }
        }
        else
            continue;
        iterations += 1;
        if( iterations > 100 )
            break;
    }
    if( !haveModel )
    {
        outAngle = NAN;
        outIntercept = NAN;
        return;
    }
    // lin_regres on the inliers (lin_regres.h:53-137); scratch holds dx | dy
    const u32 n = nBestInl;
    double* dx = scratch;
    double* dy = scratch + n;
    double sum = 0;
    for( u32 i = 0; i < n; i++ )
        sum = sum + X[ best[ i ] ];
    const double mean_x = sum / (double)n;
    sum = 0;
    for( u32 i = 0; i < n; i++ )
        sum = sum + Y[ best[ i ] ];
    const double mean_y = sum / (double)n;
    double sx = 0;
    for( u32 i = 0; i < n; i++ )
    {
        dx[ i ] = X[ best[ i ] ] - mean_x;
        sx = sx + ( dx[ i ] * dx[ i ] );
    }
    for( u32 i = 0; i < n; i++ )
        dy[ i ] = Y[ best[ i ] ] - mean_y;
    double sum_xy = 0;
    for( u32 i = 0; i < n; i++ )
        sum_xy = sum_xy + dx[ i ] * dy[ i ];
    const double slope = sum_xy / sx;
    const double intercept = mean_y - slope * mean_x;
    outAngle = lp( atan( slope ) );
    outIntercept = -intercept / slope;
}

MA_HD i64 double_to_i64( double d ) // (int64_t)d with x86 cvttsd2si semantics for NaN/out-of-range
{
    if( !( d == d ) || d >= 9223372036854775808.0 || d < -9223372036854775808.0 )
        return (i64)0x8000000000000000ull;
    return (i64)d;
}

// linesweep (harmonization.cpp:182-249): in = sh (n), out = ends; returns count
MA_HD_OUTLINE u32 linesweep( Shadow* sh, u32 n, Shadow* ends, const ma_seed* seeds, const i64 rStart, const double fAngle,
                             LibmProbe& lp )
{
    ss::sort( sh, (i64)n, ShadowOrder( ) );
    u32 ne = 0;
    u64 x = 0;
    for( u32 k = 0; k < n; k++ )
    {
        const Shadow t = sh[ k ];
        if( x < t.b )
        {
            ends[ ne++ ] = t;
            x = t.b;
        }
        else
        {
            const double fD = delta_distance( seeds[ t.seed ], fAngle, rStart, lp );
            u32 pos = ne;
            bool closer = true;
            while( pos > 0 && ends[ pos - 1 ].b >= t.b )
            {
                const double fO = delta_distance( seeds[ ends[ pos - 1 ].seed ], fAngle, rStart, lp );
                if( fO <= fD )
                {
                    closer = false;
                    break;
                }
                --pos;
            }
            if( closer )
            {
                while( ne > 0 && ends[ ne - 1 ].b >= t.b )
                    ne--;
                ends[ ne++ ] = t;
            }
        }
    }
    return ne;
}

// harmonizeOne (harmonization.cpp:251-373): S (n seeds, modified) -> out; returns count
MA_HD_OUTLINE u32 harmonize_one( ma_seed* S, u32 n, ma_seed* out, const ChainScratch& C, GlibcRand& rng, LibmProbe& lp )
{
    if( n > 1 )
    {
        for( u32 i = 0; i < n; i++ )
        {
            const double r = (double)(u64)S[ i ].r_start, q = (double)(u64)S[ i ].q_start;
            const u64 len = (u64)S[ i ].len;
            C.vX[ 3 * i ] = r + len / 2.0;
            C.vY[ 3 * i ] = q + len / 2.0;
            C.vX[ 3 * i + 1 ] = r;
            C.vY[ 3 * i + 1 ] = q;
            C.vX[ 3 * i + 2 ] = r + (double)len;
            C.vY[ 3 * i + 2 ] = q + (double)len;
        }
        const u32 np = 3 * n;
        CH_T( h0 );
        // medianAbsoluteDeviation (test_ransac.h:58-76)
        for( u32 i = 0; i < np; i++ )
            C.med[ i ] = C.vY[ i ];
        const double med = median_of( C.med, np );
        for( u32 i = 0; i < np; i++ )
        {
            const double d = C.vY[ i ] - med;
            C.med[ i ] = d < 0 ? -d : d;
        }
        const double fMAD = median_of( C.med, np );
        double fAngle, fIcpt;
        CH_T( h1 );
        CH_ADD( 5, h0, h1 );
        run_ransac( C.vX, C.vY, np, fMAD, rng, C.inl, C.best, C.med, fAngle, fIcpt, lp );
        CH_T( h2 );
        CH_ADD( 6, h1, h2 );
        const i64 rStart = double_to_i64( fIcpt );
        // remove outliers (stable remove_if)
        u32 m = 0;
        for( u32 i = 0; i < n; i++ )
            if( !( delta_distance( S[ i ], fAngle, rStart, lp ) > fMAD ) )
            {
                if( m != i )
                    S[ m ] = S[ i ];
                m++;
            }
        n = m;
        for( u32 i = 0; i < n; i++ )
        {
            C.sh1[ i ].seed = i;
            C.sh1[ i ].a = (u64)S[ i ].q_start;
            C.sh1[ i ].b = (u64)S[ i ].r_start + (u64)S[ i ].len;
        }
        u32 n2 = linesweep( C.sh1, n, C.sh2, S, rStart, fAngle, lp );
        for( u32 i = 0; i < n2; i++ )
        {
            const u32 sd = C.sh2[ i ].seed;
            C.sh1[ i ].seed = sd;
            C.sh1[ i ].a = (u64)S[ sd ].r_start;
            C.sh1[ i ].b = (u64)S[ sd ].q_start + (u64)S[ sd ].len;
        }
        const u32 n3 = linesweep( C.sh1, n2, C.sh2, S, rStart, fAngle, lp );
        CH_T( h3 );
        CH_ADD( 7, h2, h3 );
        for( u32 i = 0; i < n3; i++ )
            out[ i ] = S[ C.sh2[ i ].seed ];
        ss::sort( out, (i64)n3, SeedByRefQ( ) );
        CH_T( h4 );
        CH_ADD( 8, h3, h4 );
        if( n3 <= 1 )
        {
            out[ 0 ] = S[ n / 2 ];
            return 1;
        }
        return n3;
    }
    else if( n == 1 )
    {
        out[ 0 ] = S[ 0 ];
        return 1;
    }
    return 0;
}

// applyFilters (harmonization.cpp:14-173) in place on I (n seeds); returns new begin/count
MA_HD void apply_filters( const ChainParams& P, ma_seed* I, u32 n, u32& outBegin, u32& outCount )
{
    i64 iScore = (i64)( (u64)P.match * (u64)I[ 0 ].len );
    u64 maxScore = (u64)iScore;
    u32 lastStart = 0, optS = 0, optE = 0;
    for( u32 p = 1; p < n; p++ )
    {
        iScore += (i64)( (u64)P.match * (u64)I[ p ].len );
        u64 gap = 0;
        const u64 q1 = (u64)I[ p ].q_start, q0 = (u64)I[ p - 1 ].q_start;
        const u64 r1 = (u64)I[ p ].r_start, r0 = (u64)I[ p - 1 ].r_start;
        if( q1 > q0 )
            gap = q1 - q0;
        if( r1 > r0 )
        {
            if( r1 - r0 < gap )
            {
                gap -= r1 - r0;
                iScore += (i64)( (u64)P.match * ( r1 - r0 ) );
            }
            else
            {
                iScore += (i64)( (u64)P.match * gap );
                gap = ( r1 - r0 ) - gap;
            }
        }
        gap *= (u64)P.extend;
        if( gap > 0 )
            gap += (u64)P.gap;
        if( gap > (u64)P.sv_penalty && P.sv_penalty != 0 )
            gap = (u64)P.sv_penalty;
        if( iScore < (i64)gap )
        {
            iScore = 0;
            lastStart = p;
        }
        else
            iScore -= (i64)gap;
        if( iScore > (i64)maxScore )
        {
            maxScore = (u64)iScore;
            optS = lastStart;
            optE = p;
        }
    }
    u32 end = n;
    if( optE != n )
        if( ++optE != n )
            end = optE;
    outBegin = optS;
    outCount = end - optS;
    // artifact filter on the kept range
    ma_seed* R = I + outBegin;
    const u32 m = outCount;
    if( m > 2 )
    {
        u32 pre = 0, cen = 1;
        while( cen < m - 1 )
        {
            const i64 dPre = (i64)( (u64)R[ pre ].r_start - (u64)R[ pre ].q_start );
            const i64 dCen = (i64)( (u64)R[ cen ].r_start - (u64)R[ cen ].q_start );
            const i64 dPost = (i64)( (u64)R[ cen + 1 ].r_start - (u64)R[ cen + 1 ].q_start );
            i64 toPre = dPre - dCen;
            if( toPre < 0 )
                toPre = -toPre;
            i64 toPost = dPost - dCen;
            if( toPost < 0 )
                toPost = -toPost;
            i64 ad = toPre - toPost;
            if( ad < 0 )
                ad = -ad;
            const double diff = (double)( ad * 2 ) / ( (double)toPre + (double)toPost );
            if( diff < P.max_delta_dist && (u64)toPre > (u64)P.min_delta_dist )
            {
                R[ cen ].len = 0;
                cen++;
            }
            else
            {
                cen++;
                pre = cen - 1;
            }
        }
    }
}

// Output sink of one read: harmonized sets are appended to a shared pool
struct ChainOut
{
    ma_seed* pool; // shared
    u64 pool_cap;
    unsigned long long* pool_used; // atomic bump pointer
    HSet* sets; // this read's table (capacity set_cap)
    u32 set_cap;
    // optional private region of this read: sets go there first (HSet::off = MA_HSET_LOCAL | offset in the region) and
    // the shared pool only takes what does not fit, so the common case needs no device-wide atomic; a later pass
    // compacts everything into a dense pool in read order (k_hset_flatten)
    ma_seed* local = nullptr;
    u32 local_cap = 0;
};
#define MA_HSET_LOCAL ( 1ull << 63 )

#if defined( __HIP_DEVICE_COMPILE__ )
MA_HD u64 bump_alloc( unsigned long long* ctr, u64 n )
{
    return atomicAdd( ctr, (unsigned long long)n );
}
#else
MA_HD u64 bump_alloc( unsigned long long* ctr, u64 n )
{
    return __atomic_fetch_add( ctr, (unsigned long long)n, __ATOMIC_RELAXED );
}
#endif

// Harmonization::execute (harmonization.cpp:374-555) for one read. Returns number of sets; err flags.
// queue / nQueue: the strips were swept elsewhere (ma_batch_set_soc_heap: a SoCPriorityQueue of the reference): C.work holds
// the seeds as rectangularSoC left them, queue[] the array vMaxima; the sweep is skipped.
// preSwept: soc_windows already ran for this read (k_soc_windows, between the two wave-cooperative sorts): C.maxima / C.mm
// hold its result, nPre strips; sortedByRef: C.work is already re-sorted by reference position.
MA_HD u32 chain_read( const IndexView& X, const ChainParams& P, const ChainScratch& C, u32 nSeeds, u32 qlen,
                      const ChainOut& O, u32& err, const ma_soc* queue = nullptr, u32 nQueue = 0, bool preSwept = false, u32 nPre = 0,
                      bool sortedByRef = false )
{
    CH_T( t0 );
    u32 nmx0;
    if( preSwept )
    {
        nmx0 = nPre;
        if( nSeeds != 0 )
            soc_rebuild( C.work, nSeeds, C.maxima, C.mm, nmx0, (KeyIdx*)C.sh1, C.setA, sortedByRef );
    }
    else if( queue != nullptr )
    {
        nmx0 = nQueue < nSeeds ? nQueue : nSeeds; // a strip holds a seed at least: the scratch carved by seed count is enough
        for( u32 k = 0; k < nmx0; k++ )
        {
            SoCEntry e;
            e.accLen = queue[ k ].acc_len, e.amb = queue[ k ].ambiguity, e.cnt = queue[ k ].n_seeds;
            e.b = queue[ k ].begin < nSeeds ? queue[ k ].begin : nSeeds;
            e.e = queue[ k ].end < nSeeds ? queue[ k ].end : nSeeds;
            C.maxima[ k ] = e;
        }
    }
    else
        nmx0 = soc_sweep( X, P, C.work, nSeeds, qlen, C.maxima, C.mm, (KeyIdx*)C.sh1, C.setA );
    u32 nmx = nmx0;
    GlibcRand rng;
    rng.init( P.rng_ring );
    LibmProbe lp{ P.libm_probe, 0 };
    u32 nsets = 0;
    u32 numTries = 0, socIndex = 0, repeat = 0;
    u64 lastHarm = 0, bestSoC = 0;
    const bool heur = !P.disable_heuristics;
    const u64 switchQ = (u64)P.switch_qlen;
    u32 localUsed = 0;
    auto emit = [ & ]( ma_seed* S, u32 n, u32 soc ) {
        u32 b, m;
        apply_filters( P, S, n, b, m );
        if( nsets < O.set_cap )
        {
            u64 off;
            if( O.local && localUsed + m <= O.local_cap )
            {
                off = MA_HSET_LOCAL | (u64)localUsed;
                for( u32 i = 0; i < m; i++ )
                    O.local[ localUsed + i ] = S[ b + i ];
                localUsed += m;
            }
            else
            {
                off = bump_alloc( O.pool_used, m );
                if( off + m <= O.pool_cap )
                    for( u32 i = 0; i < m; i++ )
                        O.pool[ off + i ] = S[ b + i ];
                else
                    err |= MA_ERR_SEED_OVERFLOW;
            }
            O.sets[ nsets ].off = off;
            O.sets[ nsets ].cnt = m;
            O.sets[ nsets ].soc = soc;
        }
        else
            err |= MA_ERR_SCRATCH_OVERFLOW;
        nsets++;
    };
    while( nmx > 0 )
    {
        if( ++numTries > P.max_num_soc )
            break;
        // pop (soc.h:240-284)
        const u32 thisSoc = socIndex++;
        u32 nIn = 0;
        {
            const SoCEntry f = C.maxima[ 0 ];
            for( u32 it = f.b; it != nSeeds && it != f.e; it++ )
                C.setA[ nIn++ ] = C.work[ it ];
            ss::pop_heap( C.maxima, (i64)nmx, SoCHeapOrder( ) );
            nmx--;
        }
        u64 curSoC = 0;
        for( u32 i = 0; i < nIn; i++ )
            curSoC += (u64)C.setA[ i ].len;
        if( heur && numTries > P.min_num_soc )
        {
            if( (u64)qlen > switchQ && switchQ != 0 )
                if( lastHarm > curSoC )
                    continue;
            if( (double)bestSoC * P.soc_score_decrease_tol > (double)curSoC && P.soc_score_decrease_tol > 0 )
                break;
        }
        bestSoC = mmax( bestSoC, curSoC );
        // extractStrand(false) (seed.h:405-420) + un-mirror (harmonization.cpp:437-442)
        u32 nF = 0, nR = 0;
        for( u32 i = 0; i < nIn; i++ )
        {
            if( C.setA[ i ].on_forward == 0 )
            {
                ma_seed s = C.setA[ i ];
                s.r_start = (i64)( X.n - (u64)s.r_start - 1 );
                C.setB[ nR++ ] = s;
            }
            else
            {
                if( nF != i )
                    C.setA[ nF ] = C.setA[ i ];
                nF++;
            }
        }
        // forward strand first, then reverse: both consume RANSAC draws in this order
        const u32 nOutF = harmonize_one( C.setA, nF, C.outA, C, rng, lp );
        // the reverse output reuses setA's tail-free storage: setA is dead after harmonize_one
        ma_seed* outB = C.setA;
        // harmonize_one(setB) may read setB while writing outB (= setA): distinct arrays, fine
        const u32 nOutR = harmonize_one( C.setB, nR, outB, C, rng, lp );
        u64 curHarm = 0;
        for( u32 i = 0; i < nOutF; i++ )
            curHarm += (u64)C.outA[ i ].len;
        for( u32 i = 0; i < nOutR; i++ )
            curHarm += (u64)outB[ i ].len;
        if( heur && numTries > P.min_num_soc )
            if( curHarm < (u64)P.harm_score_min )
                continue;
        if( heur )
            if( (double)curHarm < (double)(u64)qlen * P.harm_score_min_rel )
                continue;
        if( heur && numTries > P.min_num_soc && (u64)qlen > switchQ && switchQ != 0 )
            if( lastHarm > curHarm )
                continue;
        if( nOutF > 0 )
        {
            repeat++;
            emit( C.outA, nOutF, thisSoc );
        }
        if( nOutR > 0 )
        {
            repeat++;
            emit( outB, nOutR, 0 ); // extractStrand returns a fresh Seeds: index_of_strip stays 0
        }
        if( heur && numTries > P.min_num_soc && (u64)qlen < switchQ && switchQ != 0 )
        {
            const double tol = (double)(u64)qlen * P.score_diff_tol;
            if( !( (double)curHarm + tol >= (double)lastHarm && (double)curHarm - tol <= (double)lastHarm ) )
                repeat = 0;
            if( repeat >= P.max_score_lookahead && P.max_score_lookahead != 0 )
                break;
        }
        else
            repeat = 0;
        lastHarm = curHarm;
    }
    if( heur )
        for( u32 ui = 0; ui < repeat && nsets > P.min_num_soc; ui++ )
            nsets--;
    CH_T( t1 );
    CH_ADD( 9, t0, t1 );
#if defined( MA_CHAIN_PROF ) && defined( __HIP_DEVICE_COMPILE__ )
    atomicAdd( &g_chain_prof[ 10 ], 1ull );
    atomicAdd( &g_chain_prof[ 11 ], (unsigned long long)numTries );
    atomicAdd( &g_chain_prof[ 12 ], (unsigned long long)nSeeds );
#endif
    return nsets;
}
} // namespace ma
