// SURVEY 8(f) f1, file side: builds the index of a case on the GPU, writes it with storeIndex as the reference's
// .bwt/.sa/.pac/.ann/.amb files, loads those files again with loadIndex and checks the round trip.
// usage: index_store_test <case> <prefix> [genome title -> <folder>/<title>.json]
//        index_store_test loadcheck <prefix>   (prints "ok" or "error: <what loadIndex threw>")
#include "../../ma_amd/host/ma_sam.h"
#include "../../oracle/dump_format.h"
#include <cstdio>
#include <cstring>

using namespace libMA;

int main( int argc, char** argv )
{
    if( argc < 3 )
        return 2;
    if( !strcmp( argv[ 1 ], "loadcheck" ) ) // index_store_test loadcheck <prefix>: loadIndex's validation of the files
    {
        try
        {
            std::shared_ptr<Pack> pPack;
            std::shared_ptr<FMIndex> pFM;
            loadIndex( argv[ 2 ], pPack, pFM );
            printf( "ok\n" );
        }
        catch( const std::runtime_error& e )
        {
            printf( "error: %s\n", e.what( ) );
        }
        return 0;
    }
    if( argc >= 5 && !strcmp( argv[ 4 ], "fasta" ) ) // index_store_test <genome.fa> <prefix> <title> fasta
    {
        try
        {
            std::shared_ptr<Pack> pPack;
            std::shared_ptr<FMIndex> pFM;
            srand( 12345 );
            buildIndexFromFasta( argv[ 1 ], pPack, pFM );
            storeIndex( argv[ 2 ], pPack, pFM, argv[ 3 ] );
        }
        catch( const std::runtime_error& e )
        {
            fprintf( stderr, "error: %s\n", e.what( ) );
            return 1;
        }
        return 0;
    }
    CaseFile c = readCase( argv[ 1 ] );
    std::vector<std::shared_ptr<NucSeq>> vContigs;
    for( size_t i = 0; i < c.contigs.size( ); i++ )
    {
        auto p = std::make_shared<NucSeq>( );
        p->xCodes = c.contigs[ i ];
        p->sName = c.names[ i ];
        vContigs.push_back( p );
    }
    try
    {
        std::shared_ptr<Pack> pPack, pPack2;
        std::shared_ptr<FMIndex> pFM, pFM2;
        buildIndex( vContigs, pPack, pFM );
        storeIndex( argv[ 2 ], pPack, pFM, argc >= 4 ? argv[ 3 ] : "" );
        // the reference's own index test (libs/ma/tests/index_generation.cpp): every substring of the genome is found by
        // backward search through the SuffixArrayInterface seam, and bwt_sa of its interval names its position
        {
            std::vector<uint8_t> vText;
            for( auto& pC : vContigs )
                vText.insert( vText.end( ), pC->xCodes.begin( ), pC->xCodes.end( ) );
            srand( 7 );
            for( int k = 0; k < 40; k++ )
            {
                const size_t uiLen = 25, uiPos = (size_t)rand( ) % ( vText.size( ) - uiLen );
                SAInterval ik = pFM->init_interval( vText[ uiPos + uiLen - 1 ] );
                for( size_t j = uiLen - 1; j-- > 0 && ik.size( ) > 0; )
                    ik = pFM->extend_backward( ik, vText[ uiPos + j ] );
                bool bFound = false;
                for( int64_t r = ik.start( ); r < ik.end( ) && r < ik.start( ) + 64; r++ )
                    bFound = bFound || pFM->bwt_sa( r ) == (int64_t)uiPos;
                if( ik.size( ) < 1 || !bFound )
                {
                    fprintf( stderr, "substring at %zu not found (interval size %lld)\n", uiPos, (long long)ik.size( ) );
                    return 1;
                }
            }
        }
        loadIndex( argv[ 2 ], pPack2, pFM2 );
        uint64_t a[ 3 ], b[ 3 ];
        int32_t na, nb;
        maCheck( ma_index_sizes( pFM->pDev->p, &a[ 0 ], &a[ 1 ], &a[ 2 ], &na ) );
        maCheck( ma_index_sizes( pFM2->pDev->p, &b[ 0 ], &b[ 1 ], &b[ 2 ], &nb ) );
        if( a[ 0 ] != b[ 0 ] || a[ 1 ] != b[ 1 ] || a[ 2 ] != b[ 2 ] || na != nb || pPack->vNames != pPack2->vNames ||
            pPack->vStarts != pPack2->vStarts || pPack->vLengths != pPack2->vLengths )
        {
            fprintf( stderr, "round trip differs\n" );
            return 1;
        }
        printf( "stored and reloaded: %llu bwt words, %llu sa samples, %d contigs\n", (unsigned long long)a[ 0 ],
                (unsigned long long)a[ 1 ], na );
    }
    catch( const std::runtime_error& e )
    {
        fprintf( stderr, "error: %s\n", e.what( ) );
        return 1;
    }
    return 0;
}
