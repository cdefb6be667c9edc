// CPU-only check of the FASTA/FASTQ reader of the host layer (ma_amd/host/ma_sam.h): dumps name, length and codes of
// every read exactly like `ref_dump read` does for the reference's FileReader.
// usage: reader_test <in.fa|fq> <out>
//        reader_test <in1> <out> <in2> <revcomp mate 0|1>     mate pairs of the PairedFileReader (`ref_dump readpair`)
//        reader_test <in> <out> batch <reads per batch> <block bytes>   the same dump through BatchFileReader (records cut out
//                                                               of the stream, blocks of that many bytes)
#include "ma_batch_nodes.h"

#include <cstdio>

using namespace libMA;

int main( int argc, char** argv )
{
    if( argc < 3 )
        return 2;
    ParameterSetManager xParams;
    FileReader xReader( xParams );
    FILE* f = fopen( argv[ 2 ], "w" );
    if( argc >= 5 && std::string( argv[ 3 ] ) != "batch" )
    {
        xParams.bRevCompPairedReadMates = atoi( argv[ 4 ] ) != 0;
        PairedFileReader xPairedReader( xParams );
        try
        {
            auto pStream = std::make_shared<PairedFileStream>( std::make_shared<StdFileStream>( argv[ 1 ] ),
                                                               std::make_shared<StdFileStream>( argv[ 3 ] ) );
            while( auto pPair = xPairedReader.execute( pStream ) )
                for( auto pQ : *pPair )
                {
                    fprintf( f, "%s %llu ", pQ->sName.c_str( ), (unsigned long long)pQ->length( ) );
                    for( uint8_t c : pQ->xCodes )
                        fputc( '0' + c, f );
                    fprintf( f, " %s\n", sam::fromToQual( *pQ, 0, pQ->length( ) ).c_str( ) );
                }
        }
        catch( const std::runtime_error& e )
        {
            fprintf( f, "ERROR %s\n", e.what( ) );
        }
        fclose( f );
        return 0;
    }
    if( argc >= 6 && std::string( argv[ 3 ] ) == "batch" )
    {
        try
        {
            auto pFile = std::make_shared<StdFileStream>( argv[ 1 ] );
            pFile->uiBlockBytes = (size_t)atoi( argv[ 5 ] );
            BatchFileReader xBatchReader( xParams );
            xBatchReader.uiBatchReads = (size_t)atoi( argv[ 4 ] );
            while( auto pBatch = xBatchReader.execute( pFile ) )
                for( auto pQ : *pBatch )
                {
                    fprintf( f, "%s %llu ", pQ->sName.c_str( ), (unsigned long long)pQ->length( ) );
                    for( uint8_t c : pQ->xCodes )
                        fputc( '0' + c, f );
                    fputc( '\n', f );
                }
        }
        catch( const std::runtime_error& e )
        {
            fprintf( f, "ERROR %s\n", e.what( ) );
        }
        fclose( f );
        return 0;
    }
    try
    {
        auto pStream = fileStreamFromPath( argv[ 1 ] );
        while( true )
        {
            auto pQ = xReader.execute( pStream );
            if( pQ == nullptr )
                break;
            fprintf( f, "%s %llu ", pQ->sName.c_str( ), (unsigned long long)pQ->length( ) );
            for( uint8_t c : pQ->xCodes )
                fputc( '0' + c, f );
            fputc( '\n', f );
        }
    }
    catch( const std::runtime_error& e )
    {
        fprintf( f, "ERROR %s\n", e.what( ) );
    }
    fclose( f );
    return 0;
}
