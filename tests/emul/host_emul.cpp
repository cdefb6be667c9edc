// Host emulation of the PRODUCT's stage logic (ma_amd/csrc/{seeding,chain,nw,stdsort,fm_device}.h compiled
// for the CPU) so that "-m 'not gpu'" tests can diff it against the oracle without a GPU.  This is a
// test driver: it is not part of libma_amd.so and no product entry point reaches it.  The wave-level
// ksw kernel is device-only; here the DP jobs are answered by the oracle's ma_or_ksw.
#include "../../ma_amd/csrc/chain.h"
#include "../../ma_amd/csrc/nw.h"
#include "../../ma_amd/csrc/seeding.h"
#include "../../oracle/dump_format.h"
#include "../../oracle/ma_oracle.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace ma;

static void glibc_srand_ring( u32 seed, u32 ring[ 31 ] )
{
    u32 st[ 35 ];
    ma_or_srand( seed, st );
    for( int i = 0; i < 31; i++ )
        ring[ i ] = st[ i ];
}

struct HostSink
{
    static const bool STITCH = false;
    std::vector<DpJob>* jobs;
    u64 win_begin;
    void job( u32 qf, u32 qt, u32 rf, u32 rt, i32 w, i32 zdrop, i32 flag, u32 rev )
    {
        DpJob j;
        j.win_begin = win_begin;
        j.read_off = 0;
        j.q_from = qf, j.q_to = qt, j.r_from = rf, j.r_to = rt;
        j.w = w, j.zdrop = zdrop, j.flag = flag, j.rev = rev;
        jobs->push_back( j );
    }
    KswResult next( )
    {
        return KswResult{ -1, -1, nullptr, 0 };
    }
};
struct HostStitch
{
    static const bool STITCH = true;
    std::vector<ma_or_ez>* ez;
    std::vector<std::vector<u32>>* cig;
    size_t k;
    void job( u32, u32, u32, u32, i32, i32, i32, u32 )
    {}
    KswResult next( )
    {
        KswResult R;
        R.max_q = ( *ez )[ k ].max_q;
        R.max_t = ( *ez )[ k ].max_t;
        R.n_cigar = (u32)( *cig )[ k ].size( );
        R.cigar = ( *cig )[ k ].data( );
        k++;
        return R;
    }
};

static void dumpSeed( FILE* f, const char* tag, const ma_seed& s )
{
    fprintf( f, "%s %llu %llu %llu %u %d %llu\n", tag, (unsigned long long)s.q_start, (unsigned long long)s.len,
             (unsigned long long)s.r_start, s.ambiguity, (int)s.on_forward, (unsigned long long)s.delta );
}

int main( int argc, char** argv )
{
    if( argc < 6 )
    {
        fprintf( stderr, "usage: host_emul <case> <preset> <srand_seed> <out> <stage: all|sortcheck|round3check>\n" );
        return 2;
    }
    if( std::string( argv[ 5 ] ) == "sortcheck" )
    {
        // pin ss::sort / heap ops against libstdc++ on inputs full of ties
        srand( atoi( argv[ 3 ] ) );
        struct E
        {
            int k, tag;
        };
        for( int it = 0; it < 20000; it++ )
        {
            int n = rand( ) % 200;
            int range = 1 + rand( ) % 20;
            std::vector<E> a( n );
            for( int i = 0; i < n; i++ )
                a[ i ] = E{ rand( ) % range, i };
            std::vector<E> b = a, c = a, d = a;
            auto cmp = []( const E& x, const E& y ) { return x.k < y.k; };
            std::sort( a.begin( ), a.end( ), cmp );
            ss::sort( b.data( ), (i64)n, cmp );
            for( int i = 0; i < n; i++ )
                if( a[ i ].tag != b[ i ].tag )
                {
                    fprintf( stderr, "sort mismatch n=%d\n", n );
                    return 1;
                }
            std::make_heap( c.begin( ), c.end( ), cmp );
            ss::make_heap( d.data( ), (i64)n, cmp );
            int m = n;
            while( m > 0 )
            {
                for( int i = 0; i < m; i++ )
                    if( c[ i ].tag != d[ i ].tag )
                    {
                        fprintf( stderr, "heap mismatch n=%d\n", n );
                        return 1;
                    }
                std::pop_heap( c.begin( ), c.begin( ) + m, cmp );
                ss::pop_heap( d.data( ), (i64)m, cmp );
                m--;
            }
        }
        // big arrays (heap-sort fallback / deep recursion)
        for( int it = 0; it < 30; it++ )
        {
            int n = 1000 + rand( ) % 50000;
            std::vector<E> a( n );
            int mode = it % 3;
            for( int i = 0; i < n; i++ )
                a[ i ] = E{ mode == 0 ? rand( ) % 7 : ( mode == 1 ? i / 3 : rand( ) ), i };
            std::vector<E> b = a;
            auto cmp = []( const E& x, const E& y ) { return x.k < y.k; };
            std::sort( a.begin( ), a.end( ), cmp );
            ss::sort( b.data( ), (i64)n, cmp );
            for( int i = 0; i < n; i++ )
                if( a[ i ].tag != b[ i ].tag )
                {
                    fprintf( stderr, "big sort mismatch n=%d\n", n );
                    return 1;
                }
        }
        printf( "sortcheck ok\n" );
        return 0;
    }
    if( std::string( argv[ 5 ] ) == "round3check" )
    {
        srand( atoi( argv[ 3 ] ) );
        // (1) K-mer keys by bit gathering (seed_center) == the byte loops they replace
        for( int it = 0; it < 200000; it++ )
        {
            const u32 K = 2 + rand( ) % 13;
            uint8_t b[ 16 ] = { 0 };
            for( u32 j = 0; j < K; j++ )
                b[ j ] = (uint8_t)( rand( ) & 3 );
            u64 x = 0, y = 0;
            for( u32 j = 0; j < 8; j++ )
                x |= (u64)b[ j ] << ( 8 * j ), y |= (u64)b[ 8 + j ] << ( 8 * j );
            const u32 le = kmer_gather( x, y );
            u32 keyR = 0, keyL = 0;
            for( u32 j = 0; j < K; j++ ) // right run: bases 0 .. K-1 complemented, first most significant
                keyR = ( keyR << 2 ) | ( ( 3u - b[ j ] ) & 3u );
            for( u32 j = 0; j < K; j++ ) // left run: the span's LAST base is q[centre], most significant
                keyL = ( keyL << 2 ) | ( b[ K - 1 - j ] & 3u );
            if( kmer_key_right( le, K ) != keyR || le != keyL )
            {
                fprintf( stderr, "kmer key mismatch K=%u\n", K );
                return 1;
            }
        }
        // (2) the window sweep with prefix sums == the sweep with running sums and re-added strips (ties, repeats, two contigs)
        IndexView X;
        u64 cstart[ 2 ] = { 0, 500000 }, clen[ 2 ] = { 500000, 400000 };
        X.cstart = cstart, X.clen = clen, X.n_contigs = 2, X.F = 900000, X.n = 1800000;
        ChainParams CP;
        CP.match = 2, CP.gap = 4, CP.extend = 2, CP.harm_score_min = 18, CP.harm_score_min_rel = 0.002, CP.soc_width = 0, CP.genome_size_disable = 0;
        for( int it = 0; it < 3000; it++ )
        {
            const u32 n = 1 + rand( ) % 400, qlen = 200 + rand( ) % 5000;
            std::vector<ma_seed> a( n );
            for( u32 i = 0; i < n; i++ )
            {
                a[ i ].q_start = rand( ) % qlen, a[ i ].len = 16 + rand( ) % 40;
                a[ i ].r_start = ( rand( ) % 3 == 0 ? 499000 : 20000 ) + rand( ) % ( it % 2 ? 3000 : 60000 );
                a[ i ].delta = a[ i ].r_start + qlen - a[ i ].q_start, a[ i ].ambiguity = 1 + rand( ) % 4, a[ i ].on_forward = 1;
            }
            ss::sort( a.data( ), (i64)n, SeedByDelta( ) );
            std::vector<ma_seed> b = a, tmp( n );
            std::vector<SoCEntry> m1( n + 1 ), m2( n + 1 );
            std::vector<RefMinMax> r1( n + 1 ), r2( n + 1 );
            std::vector<u64> pre( 2 * ( n + 2 ) );
            const u32 k1 = soc_windows( X, CP, a.data( ), n, qlen, m1.data( ), r1.data( ), tmp.data( ), true, nullptr );
            const u32 k2 = soc_windows( X, CP, b.data( ), n, qlen, m2.data( ), r2.data( ), tmp.data( ), true, pre.data( ) );
            bool same = k1 == k2;
            for( u32 k = 0; same && k < k1; k++ )
                same = m1[ k ].accLen == m2[ k ].accLen && m1[ k ].amb == m2[ k ].amb && m1[ k ].cnt == m2[ k ].cnt && m1[ k ].b == m2[ k ].b &&
                       m1[ k ].e == m2[ k ].e && r1[ k ].lo == r2[ k ].lo && r1[ k ].hi == r2[ k ].hi;
            if( !same )
            {
                fprintf( stderr, "soc prefix mismatch n=%u\n", n );
                return 1;
            }
        }
        printf( "round3check ok\n" );
        return 0;
    }
    CaseFile cs = readCase( argv[ 1 ] );
    std::vector<uint64_t> lens;
    std::vector<uint8_t> cat;
    for( auto& v : cs.contigs )
    {
        lens.push_back( v.size( ) );
        cat.insert( cat.end( ), v.begin( ), v.end( ) );
    }
    ma_or_index* ox = ma_or_index_build( (int32_t)lens.size( ), lens.data( ), cat.data( ) );
    IndexView X;
    std::vector<u64> cstart( lens.size( ) );
    {
        u64 o = 0;
        for( size_t i = 0; i < lens.size( ); i++ )
        {
            cstart[ i ] = o;
            o += lens[ i ];
        }
        u64 L2[ 5 ];
        i64 primary;
        u64 n;
        ma_or_index_meta( ox, L2, &primary, &n );
        X.bwt = ma_or_index_bwt( ox );
        X.sa = ma_or_index_sa( ox );
        X.pac = ma_or_index_pac( ox );
        X.cstart = cstart.data( );
        X.clen = lens.data( );
        X.n = n;
        X.F = n / 2;
        X.primary = primary;
        for( int i = 0; i < 5; i++ )
            X.L2[ i ] = L2[ i ];
        X.n_contigs = (i32)lens.size( );
    }
    ma_or_params OP;
    if( std::string( argv[ 2 ] ).compare( 0, 8, "illumina" ) == 0 )
        ma_or_params_illumina( &OP );
    else
        ma_or_params_default( &OP );
    if( std::string( argv[ 2 ] ).find( "+mems" ) != std::string::npos ) // the preset with the MEMs seeding technique
        OP.seeding_technique = 2;
    OP.srand_seed = (u32)atoi( argv[ 3 ] );
    SeedParams SP;
    SP.technique = OP.seeding_technique;
    SP.min_seed_len = OP.min_seed_len;
    SP.min_amb = OP.min_ambiguity;
    SP.max_amb = OP.max_ambiguity;
    SP.min_seed_size_drop = OP.min_seed_size_drop;
    SP.disable_heuristics = OP.disable_heuristics;
    SP.rel_min_seed_size_amount = OP.rel_min_seed_size_amount;
    SP.genome_size_disable = OP.genome_size_disable;
    if( const char* e = getenv( "MA_EMUL_SMEM_MERGE" ) ) // the kernels' default for uiMinAmbiguity == 0: twin list entries are not kept
        SP.smem_merge = atoi( e ) != 0 && SP.min_amb == 0 ? 1 : 0;
    if( const char* e = getenv( "MA_EMUL_SMEM_COMPACT" ) ) // the kernels' 16-byte list entries (one 22-bit length, the start shared by the list)
        SP.smem_compact = atoi( e ) != 0 ? 1 : 0;
    ChainParams CP;
    CP.max_num_soc = OP.max_num_soc;
    CP.min_num_soc = OP.min_num_soc;
    CP.harm_score_min = OP.harm_score_min;
    CP.max_score_lookahead = OP.max_score_lookahead;
    CP.switch_qlen = OP.switch_qlen;
    CP.min_delta_dist = OP.min_delta_dist;
    CP.sv_penalty = OP.sv_penalty;
    CP.match = OP.match;
    CP.gap = OP.gap;
    CP.extend = OP.extend;
    CP.disable_heuristics = OP.disable_heuristics;
    CP.soc_width = OP.soc_width;
    CP.genome_size_disable = OP.genome_size_disable;
    CP.harm_score_min_rel = OP.harm_score_min_rel;
    CP.soc_score_decrease_tol = OP.soc_score_decrease_tol;
    CP.score_diff_tol = OP.score_diff_tol;
    CP.max_delta_dist = OP.max_delta_dist;
    glibc_srand_ring( OP.srand_seed, CP.rng_ring );
    CP.libm_probe = 0;
    NwParams NP;
    NP.max_gap_area = OP.max_gap_area;
    NP.padding = OP.padding;
    NP.bandwidth_ext = OP.bandwidth_ext;
    NP.min_bandwidth_gap = OP.min_bandwidth_gap;
    NP.zdrop = OP.zdrop;
    NP.sv_penalty = OP.sv_penalty;
    NP.match = OP.match;
    NP.mismatch = OP.mismatch;
    NP.gap = OP.gap;
    NP.extend = OP.extend;
    NP.kq = (int8_t)OP.gap;
    NP.ke = (int8_t)OP.extend;
    NP.min_alignment_score = OP.min_alignment_score;
    NP.report_n_best = OP.report_n_best;
    NP.max_supplementary = OP.max_supplementary;
    NP.max_overlap_supplementary = OP.max_overlap_supplementary;

    FILE* f = fopen( argv[ 4 ], "w" );
    for( size_t ri = 0; ri < cs.reads.size( ); ri++ )
    {
        const std::vector<uint8_t>& q = cs.reads[ ri ];
        const u32 qlen = (u32)q.size( );
        fprintf( f, "R %zu %u\n", ri, qlen );
        // ---- seeding
        const u32 seg_cap = 6 * qlen + 8;
        std::vector<ma_segment> stage( seg_cap ), sa( qlen + 2 ), sb( qlen + 2 );
        std::vector<u32> seedStack( 2 * MA_SEED_STACK );
        SeedScratch SS{ stage.data( ), seg_cap, sa.data( ), sb.data( ), qlen + 2, SP.min_seed_size_drop, seedStack.data( ) };
        // round 6: the first n entries of each list in a separate array, as k_seed_tasks_smem keeps them in LDS (SeedScratch::lds).  The
        // array starts out as garbage: an entry that is read before it was written shows up in the dump
        std::vector<u64> heads;
        if( const char* e = getenv( "MA_EMUL_SMEM_LDS_HEADS" ) )
        {
            const u32 nh = std::min<u32>( (u32)std::max( 0, atoi( e ) ), qlen + 2 );
            heads.assign( (size_t)4 * nh + 2, ~0ull );
            SS.lds = heads.data( );
            SS.lds_n = nh;
            SS.lds_stride = 1;
        }
        SeedLane L;
        u32 nseg = 0;
        if( SP.technique == 2 )
        {
            // MEMs: one independent extension per start position (the kernel runs them as one lane each, k_mems)
            struct VecSink
            {
                std::vector<ma_segment>& v;
                void emit( u32 qs, u32 qsz, i64 sa, i64 san )
                {
                    ma_segment s;
                    s.q_start = qs, s.q_size = qsz, s.sa_start = sa, s.sa_start_rc = -1, s.sa_size = san;
                    v.push_back( s );
                }
            };
            stage.clear( );
            VecSink sink{ stage };
            u64 st = 0, bl = 0;
            for( u32 i = 0; i < qlen; i++ )
                mems_from( X, SP, q.data( ), qlen, i, sink, st, bl );
            nseg = (u32)stage.size( );
            if( !SP.disable_heuristics && SP.min_seed_size_drop != 0 ) // k_mems_finish
            {
                u64 sum = 0;
                for( auto& s : stage )
                    sum += (u64)s.q_size / (u64)SP.min_seed_size_drop;
                if( (double)sum < SP.rel_min_seed_size_amount * (double)qlen && SP.genome_size_disable < X.n )
                    nseg = 0;
            }
        }
        else
        {
            seed_begin_read( L, q.data( ), qlen );
            seed_read_serial( L, SP, SS, X );
            if( L.err )
            {
                fprintf( stderr, "seeding overflow %u\n", L.err );
                return 1;
            }
            nseg = seed_finish( L, SP, SS, X );
        }
        fprintf( f, "SEG %u\n", nseg );
        for( u32 k = 0; k < nseg; k++ )
            fprintf( f, "s %lld %lld %lld %lld %lld\n", (long long)stage[ k ].q_start, (long long)stage[ k ].q_size,
                     (long long)stage[ k ].sa_start, (long long)stage[ k ].sa_start_rc, (long long)stage[ k ].sa_size );
        // ---- extraction (k_seg_seed_counts + k_extract logic)
        std::vector<ma_seed> seeds;
        for( u32 k = 0; k < nseg; k++ )
        {
            const ma_segment& s = stage[ k ];
            if( (u64)s.q_size < (u64)OP.min_seed_len )
                continue;
            if( s.sa_size > (i64)OP.max_ambiguity && OP.max_ambiguity != 0 )
                continue;
            for( i64 row = s.sa_start; row < s.sa_start + s.sa_size; row++ )
            {
                u32 steps;
                u64 r = (u64)bwt_sa( X, row, steps );
                const bool fwd = r < X.n / 2;
                if( !fwd )
                    r = X.n - r - 1;
                ma_seed sd;
                sd.q_start = s.q_start;
                sd.len = s.q_size + 1;
                sd.r_start = (i64)r;
                sd.ambiguity = (u32)s.sa_size;
                sd.on_forward = fwd ? 1 : 0;
                u64 delta = r + ( qlen - (u64)s.q_start );
                delta += ( (u64)qlen + 1 ) * (u64)seq_id_for_position( X, r );
                sd.delta = (i64)delta;
                seeds.push_back( sd );
            }
        }
        fprintf( f, "SEED %zu\n", seeds.size( ) );
        for( auto& s : seeds )
            dumpSeed( f, "d", s );
        // ---- chaining
        const u32 n = (u32)seeds.size( );
        const size_t cap = n + 1;
        std::vector<ma_seed> work( seeds ), setA( cap ), setB( cap ), outA( cap ), hpool( 40 * cap + 64 );
        work.resize( cap );
        std::vector<SoCEntry> mx( cap );
        std::vector<RefMinMax> mm( cap );
        std::vector<Shadow> sh1( cap ), sh2( cap );
        std::vector<double> vX( 3 * cap ), vY( 3 * cap ), med( 6 * cap );
        std::vector<i32> inl( 3 * cap ), best( 3 * cap );
        ChainScratch C{ work.data( ), mx.data( ),  mm.data( ),  setA.data( ), setB.data( ), outA.data( ), sh1.data( ),
                        sh2.data( ),  vX.data( ),  vY.data( ),  med.data( ),  inl.data( ),  best.data( ) };
        std::vector<HSet> sets( 2 * OP.max_num_soc );
        unsigned long long used = 0;
        ChainOut O{ hpool.data( ), hpool.size( ), &used, sets.data( ), (u32)sets.size( ) };
        u32 err = 0;
        // SoC dump needs a private sweep
        {
            std::vector<ma_seed> w2( seeds );
            w2.resize( cap );
            std::vector<SoCEntry> mx2( cap );
            std::vector<RefMinMax> mm2( cap );
            u32 nmx = soc_sweep( X, CP, w2.data( ), n, qlen, mx2.data( ), mm2.data( ) );
            fprintf( f, "SOC %u\n", nmx );
            u32 idx = 0;
            while( nmx > 0 )
            {
                const SoCEntry e = mx2[ 0 ];
                u32 cnt = 0;
                for( u32 it = e.b; it != n && it != e.e; it++ )
                    cnt++;
                fprintf( f, "c %u %llu %u %u\n", idx++, (unsigned long long)e.accLen, e.amb, cnt );
                for( u32 it = e.b; it != n && it != e.e; it++ )
                    dumpSeed( f, "e", w2[ it ] );
                ss::pop_heap( mx2.data( ), (i64)nmx, SoCHeapOrder( ) );
                nmx--;
            }
        }
        const u32 nsets = chain_read( X, CP, C, n, qlen, O, err );
        if( err )
        {
            fprintf( stderr, "chain overflow %u\n", err );
            return 1;
        }
        fprintf( f, "HARM %u\n", nsets );
        for( u32 s = 0; s < nsets; s++ )
        {
            fprintf( f, "h %u %u\n", sets[ s ].soc, sets[ s ].cnt );
            for( u32 k = 0; k < sets[ s ].cnt; k++ )
                dumpSeed( f, "g", hpool[ sets[ s ].off + k ] );
        }
        // ---- DP: enumerate, answer with the oracle's ksw, stitch
        std::vector<AlnHeader> hdr( nsets );
        std::vector<std::vector<u64>> opsv( nsets );
        std::vector<u64> opsPool;
        std::vector<uint8_t> Q( q );
        for( u32 s = 0; s < nsets; s++ )
        {
            const ma_seed* S = hpool.data( ) + sets[ s ].off;
            AlnHeader h;
            memset( &h, 0, sizeof( h ) );
            h.soc_index = sets[ s ].soc;
            h.mapq = NAN;
            const NwWindow W = nw_window( X, NP, S, sets[ s ].cnt );
            std::vector<u64> ops( qlen + 100000 );
            h.ops_cap = (u32)ops.size( );
            if( W.valid )
            {
                std::vector<DpJob> jobs;
                HostSink sink{ &jobs, W.begin_ref };
                NwWalk<HostSink> walk{ X, NP, sink, Q.data( ), W.begin_ref, AlnBuilder{ nullptr, nullptr, nullptr } };
                walk.run( S, sets[ s ].cnt, qlen, W );
                std::vector<ma_or_ez> ez( jobs.size( ) );
                std::vector<std::vector<u32>> cig( jobs.size( ) );
                for( size_t j = 0; j < jobs.size( ); j++ )
                {
                    const DpJob& J = jobs[ j ];
                    std::vector<uint8_t> qq, tt;
                    for( u32 i = J.q_from; i < J.q_to; i++ )
                        qq.push_back( Q[ i ] );
                    for( u32 i = J.r_from; i < J.r_to; i++ )
                        tt.push_back( (uint8_t)text_base( X, J.win_begin + i ) );
                    if( J.rev )
                    {
                        std::reverse( qq.begin( ), qq.end( ) );
                        std::reverse( tt.begin( ), tt.end( ) );
                    }
                    cig[ j ].resize( qq.size( ) + tt.size( ) + 4 );
                    int nc = ma_or_ksw( (int)qq.size( ), qq.data( ), (int)tt.size( ), tt.data( ), J.w, J.zdrop, J.flag, &OP,
                                        &ez[ j ], cig[ j ].data( ), (int)cig[ j ].size( ) );
                    cig[ j ].resize( nc );
                }
                h.begin_ref = h.end_ref = W.begin_ref;
                HostStitch st{ &ez, &cig, 0 };
                NwWalk<HostStitch> w2{ X, NP, st, Q.data( ), W.begin_ref, AlnBuilder{ &h, ops.data( ), &err } };
                w2.run( S, sets[ s ].cnt, qlen, W );
            }
            h.ops_off = opsPool.size( );
            for( u32 k = 0; k < h.n_ops; k++ )
                opsPool.push_back( ops[ k ] );
            hdr[ s ] = h;
        }
        std::vector<u32> order( nsets + 1 ), mq( nsets + 1 );
        const u32 nmq = finish_read( NP, hdr.data( ), opsPool.data( ), nsets, qlen, order.data( ), mq.data( ) );
        fprintf( f, "ALN %u\n", nsets );
        for( u32 k = 0; k < nsets; k++ )
        {
            const AlnHeader& h = hdr[ order[ k ] ];
            fprintf( f, "a %llu %llu %llu %llu %lld %u %u", (unsigned long long)h.begin_ref,
                     (unsigned long long)h.end_ref, (unsigned long long)h.begin_q, (unsigned long long)h.end_q,
                     (long long)h.score, h.soc_index, h.n_ops );
            for( u32 j = 0; j < h.n_ops; j++ )
                fprintf( f, " %u:%llu", op_type( opsPool[ h.ops_off + j ] ),
                         (unsigned long long)op_len( opsPool[ h.ops_off + j ] ) );
            fprintf( f, "\n" );
        }
        fprintf( f, "MQ %u\n", nmq );
        for( u32 k = 0; k < nmq; k++ )
        {
            const AlnHeader& h = hdr[ mq[ k ] ];
            fprintf( f, "m %llu %llu %llu %llu %lld %d %d %.17g\n", (unsigned long long)h.begin_ref,
                     (unsigned long long)h.end_ref, (unsigned long long)h.begin_q, (unsigned long long)h.end_q,
                     (long long)h.score, (int)h.secondary, (int)h.supplementary, h.mapq );
        }
    }
    fclose( f );
    return 0;
}
