// TEST INFRASTRUCTURE ONLY -- the MI355X modules of ma_amd/host/ma_ref_binding.h inside the REAL reference: this program is
// compiled against the reference's own headers where they lie under /root/reference (oracle/Makefile.ref), links
// oracle/_ref/libma_ref.so (the reference compiled from its own sources) and ma_amd/libma_amd.so (the HIP engine), and builds
// the chain of libMA::setUpCompGraph (libs/ma/src/util/export.cpp:104-108) with the reference's own promiseMe / Pledge /
// simultaneousGet, its own containers, its own FileWriter -- with some or all of the five hot-path stages replaced by their
// ma_amd:: counterparts.  It travels to the GPU box as a binary under oracle/_ref/ like ref_dump; nothing in the product
// or in bench.py's timed region uses it.
//
// usage:
//   ref_graph_test pipe <case> <preset> <srand seed> <out> <stages>
//        per-read stage dump in the format of `ref_dump pipe`; <stages> = comma list out of seeding,soc,harm,dp,mq
//        ("all", "none"): the stages that run as ma_amd:: modules, the others are the reference's CPU modules
//   ref_graph_test sam  <case> <preset> <srand seed> <out.sam> <stages> <threads> [<sam options>]
//        <threads> copies of reader -> chain -> the reference's FileWriter over one shared source, driven by
//        BasePledge::simultaneousGet (module.h:268-378); prints device batch statistics as JSON on stdout
//   ref_graph_test nogpu <case>   -> constructs the modules and calls BinarySeeding::execute: must throw std::runtime_error
//                                    on a machine without a HIP device (no CPU fallback)
#include "ma_ref_binding.h"

#include "ma/module/binarySeeding.h"
#include "ma/module/fileReader.h"
#include "ma/module/fileWriter.h"
#include "ma/module/harmonization.h"
#include "ma/module/mappingQuality.h"
#include "ma/module/needlemanWunsch.h"
#include "ma/module/stripOfConsideration.h"
#include "ms/module/splitter.h"
#include "dump_format.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

using namespace libMA;
using namespace libMS;

static std::shared_ptr<NucSeq> mkSeq( const std::vector<uint8_t>& v )
{
    auto p = std::make_shared<NucSeq>( );
    if( !v.empty( ) )
        p->vAppend( v.data( ), v.size( ) );
    return p;
}

static void selectPreset( ParameterSetManager& xParams, const char* sPreset )
{
    std::string s( sPreset );
    const bool bMems = s.size( ) > 5 && s.compare( s.size( ) - 5, 5, "+mems" ) == 0;
    if( bMems )
        s.resize( s.size( ) - 5 );
    xParams.setSelected( s );
    if( bMems )
        xParams.getSelected( )->xSeedingTechnique->set( 2 );
}

static std::set<std::string> parseStages( const char* s )
{
    std::set<std::string> xRet;
    std::string sAll( s );
    if( sAll == "all" )
        return { "seeding", "soc", "harm", "dp", "mq" };
    if( sAll == "none" )
        return xRet;
    size_t uiPos = 0;
    while( uiPos <= sAll.size( ) )
    {
        size_t uiEnd = sAll.find( ',', uiPos );
        if( uiEnd == std::string::npos )
            uiEnd = sAll.size( );
        const std::string sOne = sAll.substr( uiPos, uiEnd - uiPos );
        if( sOne != "seeding" && sOne != "soc" && sOne != "harm" && sOne != "dp" && sOne != "mq" )
            throw std::runtime_error( "unknown stage " + sOne );
        xRet.insert( sOne );
        uiPos = uiEnd + 1;
    }
    return xRet;
}

// the reference's Harmonization draws from libc's rand(); the goldens were made with srand( seed ) in front of every read's
// Harmonization (ref_dump.cpp cmdPipe), the device restates exactly that
struct SeededHarmonization : public Harmonization
{
    unsigned uiSeed;
    SeededHarmonization( const ParameterSetManager& r, unsigned uiSeed ) : Harmonization( r ), uiSeed( uiSeed )
    {}
    virtual std::shared_ptr<ContainerVector<std::shared_ptr<Seeds>>> execute( std::shared_ptr<SoCPriorityQueue> pSoCIn, std::shared_ptr<NucSeq> pQuery,
                                                                             std::shared_ptr<FMIndex> pFM ) override
    {
        srand( uiSeed );
        return Harmonization::execute( pSoCIn, pQuery, pFM );
    }
};

// the five stages as shared_ptr<Module base>, either flavour
struct Stages
{
    std::shared_ptr<Module<SegmentVector, false, SuffixArrayInterface, NucSeq>> pSeeding;
    std::shared_ptr<Module<SoCPriorityQueue, false, SegmentVector, NucSeq, Pack, FMIndex>> pSoc;
    std::shared_ptr<Module<ContainerVector<std::shared_ptr<Seeds>>, false, SoCPriorityQueue, NucSeq, FMIndex>> pHarm;
    std::shared_ptr<Module<ContainerVector<std::shared_ptr<Alignment>>, false, ContainerVector<std::shared_ptr<Seeds>>, NucSeq, Pack>> pDp;
    std::shared_ptr<Module<ContainerVector<std::shared_ptr<Alignment>>, false, NucSeq, ContainerVector<std::shared_ptr<Alignment>>>> pMq;
    std::shared_ptr<ma_amd::BinarySeeding> pGpuSeeding;

    Stages( const ParameterSetManager& r, const std::set<std::string>& xGpu, unsigned uiSeed )
    {
        ma_amd::options( ).uiRansacSeed = uiSeed;
        if( xGpu.count( "seeding" ) )
            pSeeding = pGpuSeeding = std::make_shared<ma_amd::BinarySeeding>( r );
        else
            pSeeding = std::make_shared<BinarySeeding>( r );
        if( xGpu.count( "soc" ) )
            pSoc = std::make_shared<ma_amd::StripOfConsideration>( r );
        else
            pSoc = std::make_shared<StripOfConsideration>( r );
        if( xGpu.count( "harm" ) )
            pHarm = std::make_shared<ma_amd::Harmonization>( r );
        else
            pHarm = std::make_shared<SeededHarmonization>( r, uiSeed );
        if( xGpu.count( "dp" ) )
            pDp = std::make_shared<ma_amd::NeedlemanWunsch>( r );
        else
            pDp = std::make_shared<NeedlemanWunsch>( r );
        if( xGpu.count( "mq" ) )
            pMq = std::make_shared<ma_amd::MappingQuality>( r );
        else
            pMq = std::make_shared<MappingQuality>( r );
    }
};

struct RefIndex
{
    std::shared_ptr<Pack> pPack;
    std::shared_ptr<FMIndex> pFM;
};
static RefIndex buildIndex( const CaseFile& c )
{
    RefIndex r;
    r.pPack = std::make_shared<Pack>( );
    for( size_t i = 0; i < c.contigs.size( ); i++ )
        r.pPack->vAppendSequence( c.names[ i ], "", *mkSeq( c.contigs[ i ] ) );
    r.pFM = std::make_shared<FMIndex>( r.pPack );
    return r;
}

static void dumpSeeds( FILE* f, const char* tag, Seeds& s )
{
    for( auto& x : s )
        fprintf( f, "%s %llu %llu %llu %u %d %llu\n", tag, (unsigned long long)x.start( ), (unsigned long long)x.size( ),
                 (unsigned long long)x.start_ref( ), x.uiAmbiguity, (int)x.bOnForwStrand, (unsigned long long)x.uiDelta );
}

static int cmdPipe( const char* sCase, const char* sPreset, unsigned uiSeed, const char* sOut, const char* sStages )
{
    const CaseFile c = readCase( sCase );
    const RefIndex idx = buildIndex( c );
    ParameterSetManager xParams;
    selectPreset( xParams, sPreset );
    const auto xGpu = parseStages( sStages );
    if( !xGpu.empty( ) )
        ma_amd::attachIndex( idx.pPack, idx.pFM );
    ma_amd::options( ).xBatcher.bStages = true; // every container carries its content: CPU modules of the reference may follow
    Stages S( xParams, xGpu, uiSeed );
    ExtractSeeds xExtract( xParams ); // only to print the SEED section when the chain holds no GPU batch to print it from

    // the chain of export.cpp:104-108, built with the reference's own promiseMe on manually fulfilled input pledges
    auto pPack = std::make_shared<Pledge<Pack>>( );
    pPack->set( idx.pPack );
    auto pFMDIndex = std::make_shared<Pledge<FMIndex>>( );
    pFMDIndex->set( idx.pFM );
    auto pQuery = std::make_shared<Pledge<NucSeq>>( );
    auto pCast = std::make_shared<Cast<SuffixArrayInterface, FMIndex>>( xParams );
    auto pSeeds = promiseMe( S.pSeeding, promiseMe( pCast, pFMDIndex ), pQuery ); // :104
    auto pSOCs = promiseMe( S.pSoc, pSeeds, pQuery, pPack, pFMDIndex ); // :105
    auto pHarmonized = promiseMe( S.pHarm, pSOCs, pQuery, pFMDIndex ); // :106
    auto pAlignments = promiseMe( S.pDp, pHarmonized, pQuery, pPack ); // :107
    auto pAlignmentsWQuality = promiseMe( S.pMq, pQuery, pAlignments ); // :108

    FILE* f = fopen( sOut, "w" );
    for( size_t i = 0; i < c.reads.size( ); i++ )
    {
        auto pQ = mkSeq( c.reads[ i ] );
        pQ->sName = "r" + std::to_string( i );
        pQuery->set( pQ ); // invalidates the whole chain
        fprintf( f, "R %zu %zu\n", i, c.reads[ i ].size( ) );
        auto pSegs = pSeeds->get( );
        fprintf( f, "SEG %zu\n", pSegs->size( ) );
        for( auto& s : *pSegs )
            fprintf( f, "s %llu %llu %lld %lld %lld\n", (unsigned long long)s.start( ), (unsigned long long)s.size( ),
                     (long long)s.saInterval( ).start( ), (long long)s.saInterval( ).startRevComp( ), (long long)s.saInterval( ).size( ) );
        {
            // seeds in extraction order: out of the device batch when the read went through one
            const auto pTicketed = std::dynamic_pointer_cast<ma_amd::TicketedSegments>( pSegs );
            if( pTicketed != nullptr && pTicketed->xTicket )
            {
                const auto& R = *pTicketed->xTicket.pResult;
                const size_t r = pTicketed->xTicket.uiRead;
                fprintf( f, "SEED %zu\n", (size_t)( R.vSeedOff[ r + 1 ] - R.vSeedOff[ r ] ) );
                for( uint64_t k = R.vSeedOff[ r ]; k < R.vSeedOff[ r + 1 ]; k++ )
                    fprintf( f, "d %lld %lld %lld %u %d %lld\n", (long long)R.vSeeds[ k ].q_start, (long long)R.vSeeds[ k ].len,
                             (long long)R.vSeeds[ k ].r_start, R.vSeeds[ k ].ambiguity, (int)R.vSeeds[ k ].on_forward, (long long)R.vSeeds[ k ].delta );
            }
            else
            {
                auto pExtracted = xExtract.execute( pSegs, idx.pFM, pQ, idx.pPack );
                fprintf( f, "SEED %zu\n", pExtracted->size( ) );
                dumpSeeds( f, "d", *pExtracted );
            }
            // SoC pop order on a private queue: the stage module once more, popped with the reference's own pop()
            auto pPrivate = S.pSoc->execute( pSegs, pQ, idx.pPack, idx.pFM );
            fprintf( f, "SOC %zu\n", pPrivate->size( ) );
            while( !pPrivate->empty( ) )
            {
                auto uiScore = std::get<0>( pPrivate->vMaxima.front( ) ).uiAccumulativeLength;
                auto uiAmb = std::get<0>( pPrivate->vMaxima.front( ) ).uiSeedAmbiguity;
                auto p = pPrivate->pop( );
                fprintf( f, "c %u %llu %u %zu\n", p->xStats.index_of_strip, (unsigned long long)uiScore, uiAmb, p->size( ) );
                dumpSeeds( f, "e", *p );
            }
        }
        auto pHarm = pHarmonized->get( );
        fprintf( f, "HARM %zu\n", pHarm->size( ) );
        for( auto& pS : *pHarm )
        {
            fprintf( f, "h %u %zu\n", pS->xStats.index_of_strip, pS->size( ) );
            dumpSeeds( f, "g", *pS );
        }
        auto pAlns = pAlignments->get( );
        // printed before MappingQuality runs: the reference's re-sorts and re-flags these very objects
        fprintf( f, "ALN %zu\n", pAlns->size( ) );
        for( auto& pA : *pAlns )
        {
            fprintf( f, "a %llu %llu %llu %llu %lld %u %zu", (unsigned long long)pA->uiBeginOnRef, (unsigned long long)pA->uiEndOnRef,
                     (unsigned long long)pA->uiBeginOnQuery, (unsigned long long)pA->uiEndOnQuery, (long long)pA->iScore,
                     pA->xStats.index_of_strip, pA->data.size( ) );
            for( auto& d : pA->data )
                fprintf( f, " %d:%llu", (int)d.first, (unsigned long long)d.second );
            fprintf( f, "\n" );
        }
        auto pMq = pAlignmentsWQuality->get( );
        fprintf( f, "MQ %zu\n", pMq->size( ) );
        for( auto& pA : *pMq )
            fprintf( f, "m %llu %llu %llu %llu %lld %d %d %.17g\n", (unsigned long long)pA->uiBeginOnRef, (unsigned long long)pA->uiEndOnRef,
                     (unsigned long long)pA->uiBeginOnQuery, (unsigned long long)pA->uiEndOnQuery, (long long)pA->iScore,
                     (int)pA->bSecondary, (int)pA->bSupplementary, pA->fMappingQuality );
    }
    fclose( f );
    return 0;
}

// ---- the doAlign shape: shared source -> N graph copies -> the reference's FileWriter ---------------------------------
struct CaptureStream : public OutStream
{
    FILE* f;
    CaptureStream( FILE* f_ ) : f( f_ )
    {}
    OutStream& operator<<( std::string s )
    {
        fputs( s.c_str( ), f );
        return *this;
    }
};
// volatile source over the reads of a case (the role FileReader plays in export.cpp:101-103; one instance shared by all
// graph copies, every copy has its own pledge)
struct CaseSource : public Module<NucSeq, true>
{
    const CaseFile& rCase;
    std::mutex xMutex;
    size_t uiNext = 0;
    CaseSource( const CaseFile& rCase ) : rCase( rCase )
    {}
    virtual std::shared_ptr<NucSeq> execute( ) override
    {
        std::lock_guard<std::mutex> xGuard( xMutex );
        if( uiNext >= rCase.reads.size( ) )
            return nullptr; // EoF (module.h:688-695)
        auto pQ = mkSeq( rCase.reads[ uiNext ] );
        pQ->sName = "r" + std::to_string( uiNext );
        uiNext++;
        return pQ;
    }
};

static int cmdSam( const char* sCase, const char* sPreset, unsigned uiSeed, const char* sOut, const char* sStages, unsigned uiThreads,
                   int iOptions, const char* sReadsFile = nullptr )
{
    const CaseFile c = readCase( sCase );
    const RefIndex idx = buildIndex( c );
    ParameterSetManager xParams;
    selectPreset( xParams, sPreset );
    xParams.getSelected( )->xSoftClip->set( ( iOptions & 1 ) != 0 );
    xParams.getSelected( )->xOutputMCigar->set( ( iOptions & 2 ) == 0 );
    xParams.getSelected( )->xEmulateNgmlrTags->set( ( iOptions & 4 ) != 0 );
    const auto xGpu = parseStages( sStages );
    if( !xGpu.empty( ) )
    {
        auto pDev = ma_amd::attachIndex( idx.pPack, idx.pFM );
        // MA_TEST_REPLICAS=<n>: n - 1 further copies of the index ("virtual shards" on device 0); the prefetching reader
        // rotates its device batches over them, the graph is the same
        if( getenv( "MA_TEST_REPLICAS" ) && atoi( getenv( "MA_TEST_REPLICAS" ) ) > 1 && pDev->vReplicas.empty( ) )
            ma_amd::replicateIndex( pDev, std::vector<int>( (size_t)atoi( getenv( "MA_TEST_REPLICAS" ) ) - 1, 0 ) );
    }
    // all five on the GPU: the intermediate containers only pass the ticket on; a mixed chain needs their content
    ma_amd::options( ).xBatcher.bStages = xGpu.size( ) != 5;
    Stages S( xParams, xGpu, uiSeed );
    if( xGpu.count( "harm" ) == 0 && uiThreads > 1 )
        throw std::runtime_error( "the reference's Harmonization draws from the process-wide rand(): one thread only" );
    FILE* f = fopen( sOut, "w" );
    // option 32: the reference's FileWriter with its lock taken once per 64 KB (ma_amd::BufferedFileWriter); same bytes per read
    std::shared_ptr<FileWriter> pWriter;
    std::shared_ptr<ma_amd::BufferedFileWriter> pBuffered;
    if( ( iOptions & 32 ) != 0 )
        pWriter = pBuffered = std::make_shared<ma_amd::BufferedFileWriter>( xParams, std::make_shared<CaptureStream>( f ), idx.pPack );
    else
        pWriter = std::make_shared<FileWriter>( xParams, std::make_shared<CaptureStream>( f ), idx.pPack );
    auto pSource = std::make_shared<CaseSource>( c );
    auto pCast = std::make_shared<Cast<SuffixArrayInterface, FMIndex>>( xParams );
    auto pPack = std::make_shared<Pledge<Pack>>( );
    pPack->set( idx.pPack );
    auto pFMDIndex = std::make_shared<Pledge<FMIndex>>( );
    pFMDIndex->set( idx.pFM );
    std::vector<std::shared_ptr<BasePledge>> aSinks;
    auto pLock = std::make_shared<Lock<NucSeq>>( xParams );
    // option 8: the reader node wrapped into ma_amd::PrefetchReader (reads pulled ahead MA_PREFETCH_BATCH at a time and aligned
    // before any graph thread sees them); the rest of the graph is the same
    std::shared_ptr<libMS::Module<NucSeq, true>> pReaderNode = pSource;
    std::shared_ptr<ma_amd::PrefetchReader<>> pAhead;
    if( ( iOptions & 8 ) != 0 )
    {
        if( xGpu.size( ) != 5 )
            throw std::runtime_error( "the prefetching reader needs all five stages on the GPU" );
        ma_amd::options( ).xPrefetch.uiBatchReads = getenv( "MA_PREFETCH_BATCH" ) ? (size_t)atoi( getenv( "MA_PREFETCH_BATCH" ) ) : 64;
        ma_amd::options( ).xPrefetch.bStages = false;
        pAhead = std::make_shared<ma_amd::PrefetchReader<>>( xParams, pSource, idx.pFM );
        pReaderNode = pAhead;
    }
    // option 16: the reads come from a FASTA / FASTQ file through the reference's OWN FileReader (fileReader.cpp:37-203), wrapped
    // into ma_amd::PrefetchReader<FileStream> -- the line INTEGRATION.md adds at export.cpp:83
    std::shared_ptr<Pledge<FileStream>> pStreamPledge;
    std::shared_ptr<ma_amd::PrefetchReader<FileStream>> pAheadFile;
    if( ( iOptions & 16 ) != 0 )
    {
        if( xGpu.size( ) != 5 || sReadsFile == nullptr )
            throw std::runtime_error( "option 16 needs all five stages on the GPU and a reads file" );
        ma_amd::options( ).xPrefetch.uiBatchReads = getenv( "MA_PREFETCH_BATCH" ) ? (size_t)atoi( getenv( "MA_PREFETCH_BATCH" ) ) : 64;
        ma_amd::options( ).xPrefetch.bStages = false;
        pStreamPledge = std::make_shared<Pledge<FileStream>>( );
        pStreamPledge->set( std::make_shared<FileStreamFromPath>( std::string( sReadsFile ) ) );
        pAheadFile = std::make_shared<ma_amd::PrefetchReader<FileStream>>( xParams, std::make_shared<FileReader>( xParams ), idx.pFM );
    }
    BasePledge::parallelGraph( uiThreads, [ & ]( ) {
        // every graph copy its own handle of the shared constants (same objects behind a control block of the copy's own): a
        // constant pledge hands out a shared_ptr copy per get( ), and one control block for all threads is a contended cache line
        auto pPack = std::make_shared<Pledge<Pack>>( );
        pPack->set( std::shared_ptr<Pack>( idx.pPack.get( ), [ keep = idx.pPack ]( Pack* ) {} ) );
        auto pFMDIndex = std::make_shared<Pledge<FMIndex>>( );
        pFMDIndex->set( std::shared_ptr<FMIndex>( idx.pFM.get( ), [ keep = idx.pFM ]( FMIndex* ) {} ) );
        // the chain behind the reader node: export.cpp:102-124, whatever the reader node is
        auto chain = [ & ]( auto pQuery_ ) {
            auto pQuery = promiseMe( pLock, pQuery_ ); // the Lock / UnLock pair of export.cpp:101-124
            auto pSeeds = promiseMe( S.pSeeding, promiseMe( pCast, pFMDIndex ), pQuery );
            auto pSOCs = promiseMe( S.pSoc, pSeeds, pQuery, pPack, pFMDIndex );
            auto pHarmonized = promiseMe( S.pHarm, pSOCs, pQuery, pFMDIndex );
            auto pAlignments = promiseMe( S.pDp, pHarmonized, pQuery, pPack );
            auto pAlignmentsWQuality = promiseMe( S.pMq, pQuery, pAlignments );
            auto pEmptyContainer = promiseMe( pWriter, pQuery, pAlignmentsWQuality, pPack ); // export.cpp:120
            aSinks.push_back( promiseMe( std::make_shared<UnLock<libMS::Container>>( xParams, pQuery ), pEmptyContainer ) ); // :122-124
        };
        if( pAheadFile != nullptr )
            chain( promiseMe( pAheadFile, pStreamPledge ) ); // the reference's FileReader behind the prefetching reader
        else
            chain( promiseMe( pReaderNode ) ); // volatile source over the case's reads (plain, or wrapped: option 8)
    } );
    const auto tGraph = std::chrono::steady_clock::now( );
    BasePledge::simultaneousGet( aSinks, []( ) { return true; }, uiThreads );
    if( pBuffered != nullptr )
        pBuffered->flush( ); // (inside the timed region: what the threads' buffers still hold)
    const double fGraphSeconds = std::chrono::duration<double>( std::chrono::steady_clock::now( ) - tGraph ).count( );
    fclose( f );
    uint64_t uiBatches = 0, uiReads = 0;
    if( S.pGpuSeeding != nullptr )
        std::tie( uiBatches, uiReads ) = S.pGpuSeeding->batchStatistics( );
    uint64_t uiAheadBatches = 0, uiAheadReads = 0;
    if( pAhead != nullptr )
        std::tie( uiAheadBatches, uiAheadReads ) = pAhead->batchStatistics( );
    if( pAheadFile != nullptr )
        std::tie( uiAheadBatches, uiAheadReads ) = pAheadFile->batchStatistics( );
    printf( "{\"threads\": %u, \"reads\": %zu, \"device_batches\": %llu, \"reads_in_batches\": %llu, \"prefetched_batches\": %llu, "
            "\"prefetched_reads\": %llu, \"graph_seconds\": %.4f, \"reads_per_s\": %.1f}\n", uiThreads, c.reads.size( ),
            (unsigned long long)uiBatches, (unsigned long long)uiReads, (unsigned long long)uiAheadBatches, (unsigned long long)uiAheadReads,
            fGraphSeconds, c.reads.size( ) / fGraphSeconds );
    return 0;
}

static int cmdNoGpu( const char* sCase )
{
    const CaseFile c = readCase( sCase );
    const RefIndex idx = buildIndex( c );
    ParameterSetManager xParams;
    try
    {
        ma_amd::attachIndex( idx.pPack, idx.pFM );
        ma_amd::BinarySeeding xSeeding( xParams );
        auto pQ = mkSeq( c.reads[ 0 ] );
        xSeeding.execute( idx.pFM, pQ );
    }
    catch( const std::runtime_error& rE )
    {
        printf( "std::runtime_error: %s\n", rE.what( ) );
        return 0;
    }
    printf( "no exception\n" );
    return 0;
}

int main( int argc, char** argv )
{
    try
    {
        if( argc >= 7 && !strcmp( argv[ 1 ], "pipe" ) )
            return cmdPipe( argv[ 2 ], argv[ 3 ], (unsigned)atoi( argv[ 4 ] ), argv[ 5 ], argv[ 6 ] );
        if( argc >= 8 && !strcmp( argv[ 1 ], "sam" ) )
            return cmdSam( argv[ 2 ], argv[ 3 ], (unsigned)atoi( argv[ 4 ] ), argv[ 5 ], argv[ 6 ], (unsigned)atoi( argv[ 7 ] ),
                           argc > 8 ? atoi( argv[ 8 ] ) : 0, argc > 9 ? argv[ 9 ] : nullptr );
        if( argc >= 3 && !strcmp( argv[ 1 ], "nogpu" ) )
            return cmdNoGpu( argv[ 2 ] );
    }
    catch( const std::exception& rE )
    {
        fprintf( stderr, "ref_graph_test: %s\n", rE.what( ) );
        return 1;
    }
    fprintf( stderr, "usage: see the head of oracle/ref_graph_test.cpp\n" );
    return 2;
}
