// Builds the computational graph exactly like libMA::setUpCompGraph (libs/ma/src/util/export.cpp:99-126)
// but with the MI355X modules of ma_amd/host/ma_modules.h, runs every read of a case file through it
// (volatile reader -> seeding -> SoC -> harmonization -> DP -> mapping quality) and writes the
// ALN / MQ records in the common dump format.  Without a GPU the first module must throw
// std::runtime_error (mode "nogpu").
#include "../../ma_amd/host/ma_sam.h"
#include "../../oracle/dump_format.h"
#include <cstdio>

using namespace libMA;
using namespace libMS;

class Reader : public Module<NucSeq, true> // stands in for FileReader (volatile source, nullptr = EOF)
{
  public:
    const CaseFile& c;
    size_t i = 0;
    Reader( const CaseFile& c ) : c( c )
    {}
    std::shared_ptr<NucSeq> execute( ) override
    {
        if( i >= c.reads.size( ) )
            return nullptr;
        auto p = std::make_shared<NucSeq>( );
        p->xCodes = c.reads[ i ];
        p->sName = "r" + std::to_string( i );
        i++;
        return p;
    }
};

class Writer : public Module<Container, false, NucSeq, ContainerVector<std::shared_ptr<Alignment>>,
                             ContainerVector<std::shared_ptr<Alignment>>>
{
  public:
    FILE* f;
    size_t n = 0;
    Writer( FILE* f ) : f( f )
    {}
    bool requiresLock( ) const override
    {
        return true;
    }
    std::shared_ptr<Container> execute( std::shared_ptr<NucSeq> pQ, std::shared_ptr<ContainerVector<std::shared_ptr<Alignment>>> pA,
                                        std::shared_ptr<ContainerVector<std::shared_ptr<Alignment>>> pM ) override
    {
        fprintf( f, "R %zu %llu\n", n++, (unsigned long long)pQ->length( ) );
        fprintf( f, "ALN %zu\n", pA->size( ) );
        for( auto& a : *pA )
        {
            fprintf( f, "a %llu %llu %llu %llu %lld %u %zu", (unsigned long long)a->uiBeginOnRef,
                     (unsigned long long)a->uiEndOnRef, (unsigned long long)a->uiBeginOnQuery,
                     (unsigned long long)a->uiEndOnQuery, (long long)a->iScore, a->index_of_strip, a->data.size( ) );
            for( auto& d : a->data )
                fprintf( f, " %d:%llu", (int)d.first, (unsigned long long)d.second );
            fprintf( f, "\n" );
        }
        fprintf( f, "MQ %zu\n", pM->size( ) );
        for( auto& a : *pM )
            fprintf( f, "m %llu %llu %llu %llu %lld %d %d %.17g\n", (unsigned long long)a->uiBeginOnRef,
                     (unsigned long long)a->uiEndOnRef, (unsigned long long)a->uiBeginOnQuery,
                     (unsigned long long)a->uiEndOnQuery, (long long)a->iScore, (int)a->bSecondary, (int)a->bSupplementary,
                     a->fMappingQuality );
        return std::make_shared<Container>( );
    }
};

// both writers must have seen the read before it is unlocked
struct Join2 : public Module<Container, false, Container, Container>
{
    std::shared_ptr<Container> execute( std::shared_ptr<Container>, std::shared_ptr<Container> ) override
    {
        return std::make_shared<Container>( );
    }
};

int main( int argc, char** argv )
{
    if( argc < 4 )
    {
        fprintf( stderr, "usage: host_graph_test <case> <preset> <out> [nogpu]\n" );
        return 2;
    }
    CaseFile c = readCase( argv[ 1 ] );
    ParameterSetManager xParams;
    xParams.setSelected( argv[ 2 ] );
    std::vector<std::shared_ptr<NucSeq>> vContigs;
    for( size_t i = 0; i < c.contigs.size( ); i++ )
    {
        auto p = std::make_shared<NucSeq>( );
        p->xCodes = c.contigs[ i ];
        p->sName = c.names[ i ];
        vContigs.push_back( p );
    }
    std::shared_ptr<Pack> pPackC;
    std::shared_ptr<FMIndex> pFmC;
    try
    {
        buildIndex( vContigs, pPackC, pFmC );
    }
    catch( const std::runtime_error& e )
    {
        if( argc >= 5 )
        {
            printf( "nogpu: got std::runtime_error as required: %s\n", e.what( ) );
            return 0;
        }
        fprintf( stderr, "error: %s\n", e.what( ) );
        return 1;
    }
    if( argc >= 5 )
    {
        fprintf( stderr, "expected a failure without a GPU\n" );
        return 1;
    }
    FILE* f = fopen( argv[ 3 ], "w" );
    // ---- graph set-up, cf. export.cpp:84-124
    auto pPack = std::make_shared<Pledge<Pack>>( );
    pPack->set( pPackC );
    auto pFMDIndex = std::make_shared<Pledge<FMIndex>>( );
    pFMDIndex->set( pFmC );
    auto pSai = std::make_shared<Pledge<SuffixArrayInterface>>( ); // Cast(pFMDIndex) of export.cpp:104
    pSai->set( pFmC );
    auto pReader = std::make_shared<Reader>( c );
    auto pSeeding = std::make_shared<BinarySeeding>( xParams );
    auto pSOC = std::make_shared<StripOfConsideration>( xParams );
    auto pHarmonization = std::make_shared<Harmonization>( xParams );
    auto pDP = std::make_shared<NeedlemanWunsch>( xParams );
    auto pMappingQual = std::make_shared<MappingQuality>( xParams );
    auto pWriter = std::make_shared<Writer>( f );
    auto pQueries = promiseMe( pReader );
    auto pQuery = promiseMe( std::make_shared<Lock<NucSeq>>( ), pQueries ); // export.cpp:102
    auto pSeeds = promiseMe( pSeeding, pSai, pQuery );
    auto pSOCs = promiseMe( pSOC, pSeeds, pQuery, pPack, pFMDIndex );
    auto pHarmonized = promiseMe( pHarmonization, pSOCs, pQuery, pFMDIndex );
    auto pAlignments = promiseMe( pDP, pHarmonized, pQuery, pPack );
    auto pAlignmentsWQuality = promiseMe( pMappingQual, pQuery, pAlignments );
    auto pWritten = promiseMe( pWriter, pQuery, pAlignments, pAlignmentsWQuality );
    // the SAM writer of export.cpp:109-117 on an in-memory stream (argv[3] + ".sam")
    auto pSamStream = std::make_shared<StringOutStream>( );
    auto pSamWriter = std::make_shared<FileWriter>( xParams, std::static_pointer_cast<OutStream>( pSamStream ), pPackC );
    auto pSamWritten = promiseMe( pSamWriter, pQuery, pAlignmentsWQuality, pPack );
    auto pBoth = promiseMe( std::make_shared<Join2>( ), pWritten, pSamWritten );
    auto pSink = promiseMe( std::make_shared<UnLock<Container>>( pQuery ), pBoth ); // export.cpp:122-124
    BasePledge::simultaneousGet( { pSink } );
    fclose( f );
    {
        FILE* fs = fopen( ( std::string( argv[ 3 ] ) + ".sam" ).c_str( ), "w" );
        fputs( pSamStream->sText.c_str( ), fs );
        fclose( fs );
    }
    return 0;
}
