// SURVEY 8(f) f4 on the GPU: builds the graphs of libMA::setUpCompGraph / setUpCompGraphPaired
// (libs/ma/src/util/export.cpp:72-202) with the MI355X modules incl. SmallInversions (DP through ma_ksw_batch),
// PairedReads and the (Paired)FileWriter, runs the reads of a case through it and writes the f4 dump + SAM text that
// `ref_dump f4` writes for the reference.  Reads 2k and 2k+1 of the case are the mates of pair k.
// usage: f4_graph_test <case> <preset> <srand seed> <out.f4> <inversions 0|1> <paired 0|1> <zdrop inversion> <out.sam> <sam options> [batch]
// with "batch" the reads go through BatchAligner::execute / executePaired (one device batch for all reads, one GPU launch
// for all inversion DP) instead of the per-read graph; the paired dump then holds the P / PAIR / p records only
#include "../../ma_amd/host/ma_sam.h"
#include "../../oracle/dump_format.h"
#include <cstdio>
#include <cstring>

using namespace libMA;
using namespace libMS;
typedef ContainerVector<std::shared_ptr<Alignment>> AlnVec;

class PairReader : public Module<PairedReadsContainer, true> // stands in for PairedFileReader on the case's reads
{
  public:
    const CaseFile& c;
    size_t i = 0;
    const size_t uiStep;
    PairReader( const CaseFile& c, size_t uiStep ) : c( c ), uiStep( uiStep )
    {}
    std::shared_ptr<PairedReadsContainer> execute( ) override
    {
        if( i + uiStep > c.reads.size( ) )
            return nullptr;
        auto pRet = std::make_shared<PairedReadsContainer>( );
        for( size_t k = 0; k < uiStep; k++ )
        {
            auto p = std::make_shared<NucSeq>( );
            p->xCodes = c.reads[ i + k ];
            p->sName = "r" + std::to_string( i + k );
            pRet->push_back( p );
        }
        i += uiStep;
        return pRet;
    }
};

static void dumpLine( FILE* f, const char* tag, const Alignment& a, int iOther )
{
    fprintf( f, "%s %d %d %llu %llu %llu %llu %lld %u %d %d %.17g %zu", tag, (int)a.xStats.bFirst, iOther,
             (unsigned long long)a.uiBeginOnRef, (unsigned long long)a.uiEndOnRef, (unsigned long long)a.uiBeginOnQuery,
             (unsigned long long)a.uiEndOnQuery, (long long)a.iScore, a.index_of_strip, (int)a.bSecondary, (int)a.bSupplementary,
             a.fMappingQuality, a.data.size( ) );
    for( auto& d : a.data )
        fprintf( f, " %d:%llu", (int)d.first, (unsigned long long)d.second );
    fprintf( f, "\n" );
}

// dump node of the single-read graph
class DumpOne : public Module<Container, false, NucSeq, AlnVec>
{
  public:
    FILE* f;
    size_t n = 0;
    DumpOne( FILE* f ) : f( f )
    {}
    std::shared_ptr<Container> execute( std::shared_ptr<NucSeq> pQ, std::shared_ptr<AlnVec> pFin ) override
    {
        fprintf( f, "R %zu %llu\nFIN 0 %zu\n", n++, (unsigned long long)pQ->length( ), pFin->size( ) );
        for( auto& a : *pFin )
            dumpLine( f, "f", *a, -1 );
        return std::make_shared<Container>( );
    }
};
// dump node of the paired graph (after PairedReads, like ref_dump)
class DumpPair : public Module<Container, false, NucSeq, NucSeq, AlnVec, AlnVec, AlnVec>
{
  public:
    FILE* f;
    size_t n = 0;
    DumpPair( FILE* f ) : f( f )
    {}
    std::shared_ptr<Container> execute( std::shared_ptr<NucSeq> pQ1, std::shared_ptr<NucSeq> pQ2, std::shared_ptr<AlnVec> pFin1,
                                        std::shared_ptr<AlnVec> pFin2, std::shared_ptr<AlnVec> pPair ) override
    {
        fprintf( f, "P %zu %llu %llu\n", n++, (unsigned long long)pQ1->length( ), (unsigned long long)pQ2->length( ) );
        fprintf( f, "FIN 0 %zu\n", pFin1->size( ) );
        for( auto& a : *pFin1 )
            dumpLine( f, "f", *a, -1 );
        fprintf( f, "FIN 1 %zu\n", pFin2->size( ) );
        for( auto& a : *pFin2 )
            dumpLine( f, "f", *a, -1 );
        fprintf( f, "PAIR %zu\n", pPair->size( ) );
        for( auto& a : *pPair )
        {
            int iOther = -1;
            auto pO = a->xStats.pOther.lock( );
            for( size_t j = 0; pO != nullptr && j < pPair->size( ); j++ )
                if( ( *pPair )[ j ] == pO )
                    iOther = (int)j;
            dumpLine( f, "p", *a, iOther );
        }
        return std::make_shared<Container>( );
    }
};
struct Join2 : public Module<Container, false, Container, Container>
{
    std::shared_ptr<Container> execute( std::shared_ptr<Container>, std::shared_ptr<Container> ) override
    {
        return std::make_shared<Container>( );
    }
};

int main( int argc, char** argv )
{
    if( argc < 10 )
    {
        fprintf( stderr, "usage: f4_graph_test <case> <preset> <seed> <out.f4> <inv> <paired> <zdrop inv> <out.sam> <sam options>\n" );
        return 2;
    }
    CaseFile c = readCase( argv[ 1 ] );
    const bool bInv = atoi( argv[ 5 ] ) != 0, bPaired = atoi( argv[ 6 ] ) != 0;
    const int iOptions = atoi( argv[ 9 ] );
    ParameterSetManager xParams;
    xParams.setSelected( argv[ 2 ] );
    xParams.getSelected( )->srand_seed = (uint32_t)atoi( argv[ 3 ] );
    xParams.getSelected( )->search_inversions = bInv;
    xParams.getSelected( )->zdrop_inversion = atoi( argv[ 7 ] );
    xParams.xSam.bSoftClip = ( iOptions & 1 ) != 0;
    xParams.xSam.bOutputMCigar = ( iOptions & 2 ) == 0;
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
        fprintf( stderr, "error: %s\n", e.what( ) );
        return 1;
    }
    FILE* f = fopen( argv[ 4 ], "w" );
    if( argc >= 11 && !strcmp( argv[ 10 ], "batch" ) )
    {
        auto pAll = std::make_shared<ContainerVector<std::shared_ptr<NucSeq>>>( );
        for( size_t i = 0; i < c.reads.size( ); i++ )
        {
            auto p = std::make_shared<NucSeq>( );
            p->xCodes = c.reads[ i ];
            p->sName = "r" + std::to_string( i );
            pAll->push_back( p );
        }
        if( bPaired && pAll->size( ) % 2 )
            pAll->pop_back( );
        BatchAligner xBatch( xParams );
        auto pSamStream = std::make_shared<StringOutStream>( );
        if( !bPaired )
        {
            auto pRes = xBatch.execute( pFmC, pAll );
            FileWriter xWriter( xParams, std::static_pointer_cast<OutStream>( pSamStream ), pPackC );
            for( size_t i = 0; i < pAll->size( ); i++ )
            {
                fprintf( f, "R %zu %llu\nFIN 0 %zu\n", i, (unsigned long long)( *pAll )[ i ]->length( ), ( *pRes )[ i ]->size( ) );
                for( auto& a : *( *pRes )[ i ] )
                    dumpLine( f, "f", *a, -1 );
                xWriter.execute( ( *pAll )[ i ], ( *pRes )[ i ], pPackC );
            }
        }
        else
        {
            auto pRes = xBatch.executePaired( pFmC, pAll );
            PairedFileWriter xWriter( xParams, std::static_pointer_cast<OutStream>( pSamStream ), pPackC );
            for( size_t k = 0; k < pRes->size( ); k++ )
            {
                auto pPair = ( *pRes )[ k ];
                fprintf( f, "P %zu %llu %llu\nPAIR %zu\n", k, (unsigned long long)( *pAll )[ 2 * k ]->length( ),
                         (unsigned long long)( *pAll )[ 2 * k + 1 ]->length( ), pPair->size( ) );
                for( auto& a : *pPair )
                {
                    int iOther = -1;
                    auto pO = a->xStats.pOther.lock( );
                    for( size_t j = 0; pO != nullptr && j < pPair->size( ); j++ )
                        if( ( *pPair )[ j ] == pO )
                            iOther = (int)j;
                    dumpLine( f, "p", *a, iOther );
                }
                xWriter.execute( ( *pAll )[ 2 * k ], ( *pAll )[ 2 * k + 1 ], pPair, pPackC );
            }
        }
        fclose( f );
        FILE* fs = fopen( argv[ 8 ], "w" );
        fputs( pSamStream->sText.c_str( ), fs );
        fclose( fs );
        return 0;
    }
    auto pPack = std::make_shared<Pledge<Pack>>( );
    pPack->set( pPackC );
    auto pFMDIndex = std::make_shared<Pledge<FMIndex>>( );
    pFMDIndex->set( pFmC );
    auto pSai = std::make_shared<Pledge<SuffixArrayInterface>>( );
    pSai->set( pFmC );
    auto pSeeding = std::make_shared<BinarySeeding>( xParams );
    auto pSOC = std::make_shared<StripOfConsideration>( xParams );
    auto pHarmonization = std::make_shared<Harmonization>( xParams );
    auto pDP = std::make_shared<NeedlemanWunsch>( xParams );
    auto pMappingQual = std::make_shared<MappingQuality>( xParams );
    auto pSmallInversions = std::make_shared<SmallInversions>( xParams );
    auto pPairedReads = std::make_shared<PairedReads>( xParams );
    auto pSamStream = std::make_shared<StringOutStream>( );
    // one mate's chain (export.cpp:167-186)
    auto fChain = [ & ]( auto pQuery ) {
        auto pSeeds = promiseMe( pSeeding, pSai, pQuery );
        auto pSOCs = promiseMe( pSOC, pSeeds, pQuery, pPack, pFMDIndex );
        auto pHarmonized = promiseMe( pHarmonization, pSOCs, pQuery, pFMDIndex );
        auto pAlignments = promiseMe( pDP, pHarmonized, pQuery, pPack );
        return promiseMe( pMappingQual, pQuery, pAlignments );
    };
    auto pTuples = promiseMe( std::make_shared<PairReader>( c, bPaired ? 2 : 1 ) );
    auto pTuple = promiseMe( std::make_shared<Lock<PairedReadsContainer>>( ), pTuples );
    auto pQueryA = promiseMe( std::make_shared<TupleGet<PairedReadsContainer, 0>>( ), pTuple );
    auto pMqA = fChain( pQueryA );
    std::shared_ptr<BasePledge> pSink;
    if( !bPaired )
    {
        auto fRest = [ & ]( auto pFinal ) {
            auto pDumped = promiseMe( std::make_shared<DumpOne>( f ), pQueryA, pFinal );
            auto pSamWriter = std::make_shared<FileWriter>( xParams, std::static_pointer_cast<OutStream>( pSamStream ), pPackC );
            auto pWritten = promiseMe( pSamWriter, pQueryA, pFinal, pPack );
            auto pBoth = promiseMe( std::make_shared<Join2>( ), pDumped, pWritten );
            pSink = promiseMe( std::make_shared<UnLock<Container>>( pTuple ), pBoth );
        };
        if( bInv )
            fRest( promiseMe( pSmallInversions, pMqA, pQueryA, pPack ) );
        else
            fRest( pMqA );
    }
    else
    {
        auto pQueryB = promiseMe( std::make_shared<TupleGet<PairedReadsContainer, 1>>( ), pTuple );
        auto pMqB = fChain( pQueryB );
        auto fRest = [ & ]( auto pFinA, auto pFinB ) {
            auto pPair = promiseMe( pPairedReads, pQueryA, pQueryB, pFinA, pFinB, pPack );
            auto pDumped = promiseMe( std::make_shared<DumpPair>( f ), pQueryA, pQueryB, pFinA, pFinB, pPair );
            auto pSamWriter = std::make_shared<PairedFileWriter>( xParams, std::static_pointer_cast<OutStream>( pSamStream ), pPackC );
            auto pWritten = promiseMe( pSamWriter, pQueryA, pQueryB, pPair, pPack );
            auto pBoth = promiseMe( std::make_shared<Join2>( ), pDumped, pWritten );
            pSink = promiseMe( std::make_shared<UnLock<Container>>( pTuple ), pBoth );
        };
        if( bInv )
            fRest( promiseMe( pSmallInversions, pMqA, pQueryA, pPack ), promiseMe( pSmallInversions, pMqB, pQueryB, pPack ) );
        else
            fRest( pMqA, pMqB );
    }
    BasePledge::simultaneousGet( { pSink } );
    fclose( f );
    FILE* fs = fopen( argv[ 8 ], "w" );
    fputs( pSamStream->sText.c_str( ), fs );
    fclose( fs );
    return 0;
}
