// examples/ma_align.cpp -- reads in, SAM out, on the GPU, with the drop-in host layer (ma_amd/host): what a user of
// `maCMD -x <genome> -i <reads> [-m <mates>] -o <out.sam> -p <preset>` needs from the hot path.  Not a re-implementation
// of cmdMa.cpp: no option parsing beyond the four arguments, no thread pool (the batches are the parallelism).
//
//   ma_align <genome.fa | index prefix> <reads.fa|fq[.gz]> <out.sam|stdout> [preset] [mates.fa|fq[.gz]]
//
// build: g++ -std=c++17 -O2 [-DMA_WITH_ZLIB] -Iinclude -Ima_amd/host examples/ma_align.cpp -Lma_amd -lma_amd [-lz] -lpthread
#include "ma_sam.h"
#include <cstdio>

using namespace libMA;
typedef libMS::ContainerVector<std::shared_ptr<NucSeq>> ReadVec;

int main( int argc, char** argv )
{
    if( argc < 4 )
    {
        fprintf( stderr, "usage: ma_align <genome.fa | index prefix> <reads> <out.sam|stdout> [preset] [mates]\n" );
        return 2;
    }
    try
    {
        ParameterSetManager xParams;
        xParams.setSelected( argc >= 5 ? argv[ 4 ] : ( argc >= 6 ? "illuminapaired" : "default" ) );
        const bool bPaired = argc >= 6;
        std::shared_ptr<Pack> pPack;
        std::shared_ptr<FMIndex> pFM;
        const std::string sGenome = argv[ 1 ];
        if( std::ifstream( sGenome + ".bwt" ).good( ) )
            loadIndex( sGenome, pPack, pFM ); // the reference's own index files (or storeIndex output)
        else
        {
            srand( 1 );
            buildIndexFromFasta( sGenome, pPack, pFM ); // suffix sort on the GPU
        }
        BatchAligner xAligner( xParams );
        FileReader xReader( xParams );
        auto pIn = fileStreamFromPath( argv[ 2 ] );
        const size_t uiBatch = 1000000; // reads per device batch
        if( !bPaired )
        {
            FileWriter xWriter( xParams, std::string( argv[ 3 ] ), pPack );
            while( true )
            {
                auto pReads = std::make_shared<ReadVec>( );
                while( pReads->size( ) < uiBatch )
                {
                    auto pQ = xReader.execute( pIn );
                    if( pQ == nullptr )
                        break;
                    pReads->push_back( pQ );
                }
                if( pReads->empty( ) )
                    break;
                auto pRes = xAligner.execute( pFM, pReads );
                for( size_t i = 0; i < pReads->size( ); i++ )
                    xWriter.execute( ( *pReads )[ i ], ( *pRes )[ i ], pPack );
            }
        }
        else
        {
            PairedFileReader xPairedReader( xParams );
            auto pStreams = std::make_shared<PairedFileStream>( pIn, fileStreamFromPath( argv[ 5 ] ) );
            PairedFileWriter xWriter( xParams, std::string( argv[ 3 ] ), pPack );
            while( true )
            {
                auto pMates = std::make_shared<ReadVec>( );
                while( pMates->size( ) < uiBatch )
                {
                    auto pPair = xPairedReader.execute( pStreams );
                    if( pPair == nullptr )
                        break;
                    pMates->push_back( ( *pPair )[ 0 ] );
                    pMates->push_back( ( *pPair )[ 1 ] );
                }
                if( pMates->empty( ) )
                    break;
                auto pRes = xAligner.executePaired( pFM, pMates );
                for( size_t k = 0; k < pRes->size( ); k++ )
                    xWriter.execute( ( *pMates )[ 2 * k ], ( *pMates )[ 2 * k + 1 ], ( *pRes )[ k ], pPack );
            }
        }
    }
    catch( const std::runtime_error& e )
    {
        fprintf( stderr, "error: %s\n", e.what( ) );
        return 1;
    }
    return 0;
}
