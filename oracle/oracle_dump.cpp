// TEST INFRASTRUCTURE ONLY -- CLI twin of ref_dump.cpp that drives OUR CPU restatement (ma_oracle)
// and writes the same text format, so `cmp` against the real reference's dump pins parity.
//   oracle_dump index <case> <out_prefix>
//   oracle_dump pipe  <case> <preset> <srand_seed> <out>
//   oracle_dump ext   <case> <out>
//   oracle_dump ksw   <kswcase> <out>
#include "dump_format.h"
#include "ma_oracle.h"
#include <cstdlib>

static ma_or_index* build( const CaseFile& c )
{
    std::vector<uint64_t> lens;
    std::vector<uint8_t> cat;
    for( auto& v : c.contigs )
    {
        lens.push_back( v.size( ) );
        cat.insert( cat.end( ), v.begin( ), v.end( ) );
    }
    return ma_or_index_build( (int32_t)lens.size( ), lens.data( ), cat.data( ) );
}

int main( int argc, char** argv )
{
    if( argc >= 4 && !strcmp( argv[ 1 ], "index" ) )
    {
        CaseFile c = readCase( argv[ 2 ] );
        ma_or_index* x = build( c );
        return ma_or_index_store( x, argv[ 3 ] );
    }
    if( argc >= 6 && !strcmp( argv[ 1 ], "pipe" ) )
    {
        CaseFile c = readCase( argv[ 2 ] );
        ma_or_index* x = build( c );
        ma_or_params P;
        if( !strncmp( argv[ 3 ], "illumina", 8 ) )
            ma_or_params_illumina( &P );
        else
            ma_or_params_default( &P );
        if( strstr( argv[ 3 ], "+mems" ) ) // the preset with the MEMs seeding technique (see ref_dump.cpp selectPreset)
            P.seeding_technique = 2;
        P.srand_seed = (uint32_t)atoi( argv[ 4 ] );
        if( argc >= 12 ) // pipe <case> <preset> <seed> <out> match mismatch gap extend gap2 extend2
        {
            P.match = atoi( argv[ 6 ] ), P.mismatch = atoi( argv[ 7 ] ), P.gap = atoi( argv[ 8 ] ), P.extend = atoi( argv[ 9 ] );
            P.gap2 = atoi( argv[ 10 ] ), P.extend2 = atoi( argv[ 11 ] );
        }
        std::vector<uint8_t> cat;
        std::vector<uint64_t> off{ 0 };
        for( auto& r : c.reads )
        {
            cat.insert( cat.end( ), r.begin( ), r.end( ) );
            off.push_back( cat.size( ) );
        }
        return ma_or_dump_pipe( x, &P, cat.data( ), off.data( ), c.reads.size( ), argv[ 5 ] );
    }
    if( argc >= 9 && !strcmp( argv[ 1 ], "f4" ) ) // f4 <case> <preset> <seed> <out> <inversions> <paired> <zdrop_inversion>
    {
        CaseFile c = readCase( argv[ 2 ] );
        ma_or_index* x = build( c );
        ma_or_params P;
        if( !strcmp( argv[ 3 ], "illumina" ) )
            ma_or_params_illumina( &P );
        else
            ma_or_params_default( &P );
        P.srand_seed = (uint32_t)atoi( argv[ 4 ] );
        P.search_inversions = atoi( argv[ 6 ] );
        P.use_paired_reads = atoi( argv[ 7 ] );
        P.zdrop_inversion = atoi( argv[ 8 ] );
        std::vector<uint8_t> cat;
        std::vector<uint64_t> off{ 0 };
        for( auto& r : c.reads )
        {
            cat.insert( cat.end( ), r.begin( ), r.end( ) );
            off.push_back( cat.size( ) );
        }
        cat.push_back( 0 );
        return ma_or_dump_f4( x, &P, cat.data( ), off.data( ), c.reads.size( ), argv[ 5 ] );
    }
    if( argc >= 4 && !strcmp( argv[ 1 ], "ext" ) )
    {
        CaseFile c = readCase( argv[ 2 ] );
        ma_or_index* x = build( c );
        uint64_t L2[ 5 ];
        int64_t primary;
        uint64_t n;
        ma_or_index_meta( x, L2, &primary, &n );
        FILE* f = fopen( argv[ 3 ], "w" );
        for( size_t i = 0; i < c.reads.size( ); i++ )
        {
            auto& q = c.reads[ i ];
            fprintf( f, "R %zu %zu\n", i, q.size( ) );
            if( q.empty( ) || q.back( ) >= 4 )
                continue;
            uint8_t c0 = q.back( );
            int64_t ik[ 3 ] = { (int64_t)L2[ c0 ] + 1, (int64_t)L2[ 3 - c0 ] + 1, (int64_t)( L2[ c0 + 1 ] - L2[ c0 ] ) };
            fprintf( f, "i %lld %lld %lld\n", (long long)ik[ 0 ], (long long)ik[ 1 ], (long long)ik[ 2 ] );
            for( size_t j = q.size( ) - 1; j-- > 0 && ik[ 2 ] > 0; )
            {
                int64_t ok[ 3 ];
                for( uint8_t cc = 0; cc < 5; cc++ )
                {
                    ma_or_extend_backward( x, ik, cc, ok );
                    fprintf( f, "x %d %lld %lld %lld\n", (int)cc, (long long)ok[ 0 ], (long long)ok[ 1 ],
                             (long long)ok[ 2 ] );
                }
                ma_or_extend_backward( x, ik, q[ j ], ok );
                memcpy( ik, ok, sizeof( ik ) );
            }
            if( ik[ 2 ] > 0 && ik[ 2 ] <= 64 )
                for( int64_t p = ik[ 0 ]; p < ik[ 0 ] + ik[ 2 ]; p++ )
                    fprintf( f, "p %lld %lld\n", (long long)p, (long long)ma_or_bwt_sa( x, p ) );
        }
        fclose( f );
        return 0;
    }
    if( argc >= 4 && !strcmp( argv[ 1 ], "ksw" ) )
    {
        std::vector<KswCase> v = readKswCases( argv[ 2 ] );
        ma_or_params P;
        ma_or_params_default( &P );
        if( argc >= 11 ) // ksw <cases> <out> <dirty|clean> match mismatch gap extend gap2 extend2
        {
            P.match = atoi( argv[ 5 ] ), P.mismatch = atoi( argv[ 6 ] ), P.gap = atoi( argv[ 7 ] ), P.extend = atoi( argv[ 8 ] );
            P.gap2 = atoi( argv[ 9 ] ), P.extend2 = atoi( argv[ 10 ] );
        }
        FILE* f = fopen( argv[ 3 ], "w" );
        std::vector<uint32_t> cig( 1 << 20 );
        for( size_t i = 0; i < v.size( ); i++ )
        {
            auto& k = v[ i ];
            ma_or_ez ez;
            int n = ma_or_ksw( (int)k.q.size( ), k.q.data( ), (int)k.t.size( ), k.t.data( ), k.w, k.zdrop, k.flag, &P, &ez,
                               cig.data( ), (int)cig.size( ) );
            fprintf( f, "k %zu %u %u %d %d %d %d %d %d %d %d %d", i, (unsigned)ez.max, (unsigned)ez.zdropped, ez.max_q,
                     ez.max_t, ez.mqe, ez.mqe_t, ez.mte, ez.mte_q, ez.score, ez.reach_end, ez.n_cigar );
            for( int j = 0; j < n; j++ )
                fprintf( f, " %u", cig[ j ] );
            fprintf( f, "\n" );
        }
        fclose( f );
        return 0;
    }
    fprintf( stderr, "usage: oracle_dump index|pipe|ext|ksw ...\n" );
    return 2;
}
