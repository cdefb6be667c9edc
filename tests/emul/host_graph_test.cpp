// Builds the computational graph exactly like libMA::setUpCompGraph (libs/ma/src/util/export.cpp:99-126)
// but with the MI355X modules of ma_amd/host/ma_modules.h, runs every read of a case file through it
// (volatile reader -> seeding -> SoC -> harmonization -> DP -> mapping quality) and writes the
// ALN / MQ records in the common dump format.  Without a GPU the first module must throw
// std::runtime_error (mode "nogpu").  Further modes (argv[4]):
//   threads <N> [repeat]  N graph copies over one shared reader, like BasePledge::parallelGraph + simultaneousGet with N
//                         threads (export.cpp:84-126): the per-read execute() calls funnel into device batches; the records
//                         are written in read order afterwards; with repeat the read set is cycled <repeat> times and only
//                         the rate is printed
//   mixed                 every module gets a container WITHOUT a ticket (rebuilt from the previous module's output), i.e.
//                         each stage runs on its own through the stage inputs of the C ABI, the way a maintainer replaces
//                         the reference's modules one at a time
//   socs                  pops every read's SoC queue and writes the SOC records of the common dump format
//   multi <shards>        MultiDeviceAligner over <shards> index replicas on device 0 (virtual shards)
//   multiflat <shards>    MultiDeviceAligner::executeFlat over <shards> virtual shards, run TWICE: the second run must not
//                         create an engine (persistent engines) and must give the same records
// MA_TEST_REPLICAS=<n>: n - 1 further copies of the index are attached to the FMIndex / Pack of the graph
// (libMA::replicateIndex, "virtual shards" on device 0): prefetch, batchgraph and the BatchAligner forms then rotate their
// device batches over n replicas while the graph stays as it is.
#include "../../ma_amd/host/ma_batch_nodes.h"
#include "../../oracle/dump_format.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace libMA;
using namespace libMS;

class Reader : public Module<NucSeq, true> // stands in for FileReader (volatile source, nullptr = EOF)
{
  public:
    const CaseFile& c;
    size_t i = 0, uiTotal;
    std::mutex xMutex; // the reference's FileReader locks its stream the same way (fileReader.cpp:37-45)
    Reader( const CaseFile& c, size_t uiRepeat = 1 ) : c( c ), uiTotal( c.reads.size( ) * uiRepeat )
    {}
    std::shared_ptr<NucSeq> execute( ) override
    {
        std::lock_guard<std::mutex> xGuard( xMutex );
        if( i >= uiTotal )
            return nullptr;
        auto p = std::make_shared<NucSeq>( );
        p->xCodes = c.reads[ i % c.reads.size( ) ];
        p->sName = "r" + std::to_string( i );
        i++;
        return p;
    }
};

// collects the per-read results of a multi-threaded run; index = number in the read's name
struct Collected
{
    std::mutex xMutex;
    std::vector<std::shared_ptr<NucSeq>> vQ;
    std::vector<std::shared_ptr<ContainerVector<std::shared_ptr<Alignment>>>> vA, vM;
};
class Collector : public Module<Container, false, NucSeq, ContainerVector<std::shared_ptr<Alignment>>,
                                ContainerVector<std::shared_ptr<Alignment>>>
{
  public:
    Collected& r;
    const bool bKeep;
    std::atomic<size_t> uiSeen{ 0 }, uiAligned{ 0 };
    Collector( Collected& r, bool bKeep ) : r( r ), bKeep( bKeep )
    {}
    std::shared_ptr<Container> execute( std::shared_ptr<NucSeq> pQ, std::shared_ptr<ContainerVector<std::shared_ptr<Alignment>>> pA,
                                        std::shared_ptr<ContainerVector<std::shared_ptr<Alignment>>> pM ) override
    {
        uiSeen++;
        if( !pM->empty( ) )
            uiAligned++;
        if( bKeep )
        {
            const size_t k = (size_t)atoll( pQ->sName.c_str( ) + 1 );
            std::lock_guard<std::mutex> xGuard( r.xMutex );
            if( r.vQ.size( ) <= k )
                r.vQ.resize( k + 1 ), r.vA.resize( k + 1 ), r.vM.resize( k + 1 );
            r.vQ[ k ] = pQ, r.vA[ k ] = pA, r.vM[ k ] = pM;
        }
        return std::make_shared<Container>( );
    }
};

// "mixed" mode: hands a module's output on as a plain container, as if a module of the reference had produced it
template <typename TP> struct Detach : public Module<TP, false, TP>
{
    const bool bActive; // false: pass the container on as it is (the normal graph)
    Detach( bool bActive ) : bActive( bActive )
    {}
    std::shared_ptr<TP> execute( std::shared_ptr<TP> pIn ) override;
};
template <> std::shared_ptr<SegmentVector> Detach<SegmentVector>::execute( std::shared_ptr<SegmentVector> pIn )
{
    if( !bActive )
        return pIn;
    auto p = std::make_shared<SegmentVector>( );
    for( const Segment& s : *pIn )
        p->push_back( s );
    return p;
}
template <> std::shared_ptr<SoCPriorityQueue> Detach<SoCPriorityQueue>::execute( std::shared_ptr<SoCPriorityQueue> pIn )
{
    if( !bActive )
        return pIn;
    auto p = std::make_shared<SoCPriorityQueue>( );
    p->pSeeds = std::make_shared<Seeds>( *pIn->pSeeds );
    return p;
}
template <> std::shared_ptr<SeedsSetVector> Detach<SeedsSetVector>::execute( std::shared_ptr<SeedsSetVector> pIn )
{
    if( !bActive )
        return pIn;
    auto p = std::make_shared<SeedsSetVector>( );
    for( auto& pS : *pIn )
        p->push_back( std::make_shared<Seeds>( *pS ) );
    return p;
}

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

static void writeRecords( FILE* f, size_t n, const NucSeq& rQ, const ContainerVector<std::shared_ptr<Alignment>>& rA,
                          const ContainerVector<std::shared_ptr<Alignment>>* pM )
{
    fprintf( f, "R %zu %llu\n", n, (unsigned long long)rQ.length( ) );
    if( pM != nullptr )
    {
        fprintf( f, "ALN %zu\n", rA.size( ) );
        for( auto& a : rA )
        {
            fprintf( f, "a %llu %llu %llu %llu %lld %u %zu", (unsigned long long)a->uiBeginOnRef, (unsigned long long)a->uiEndOnRef,
                     (unsigned long long)a->uiBeginOnQuery, (unsigned long long)a->uiEndOnQuery, (long long)a->iScore,
                     a->index_of_strip, a->data.size( ) );
            for( auto& d : a->data )
                fprintf( f, " %d:%llu", (int)d.first, (unsigned long long)d.second );
            fprintf( f, "\n" );
        }
    }
    const auto& rM = pM != nullptr ? *pM : rA;
    fprintf( f, "MQ %zu\n", rM.size( ) );
    for( auto& a : rM )
        fprintf( f, "m %llu %llu %llu %llu %lld %d %d %.17g\n", (unsigned long long)a->uiBeginOnRef, (unsigned long long)a->uiEndOnRef,
                 (unsigned long long)a->uiBeginOnQuery, (unsigned long long)a->uiEndOnQuery, (long long)a->iScore, (int)a->bSecondary,
                 (int)a->bSupplementary, a->fMappingQuality );
}

// N graph copies over one shared reader (export.cpp:84-126: parallelGraph + simultaneousGet)
static int runThreads( const CaseFile& c, const ParameterSetManager& xParams, std::shared_ptr<Pack> pPackC, std::shared_ptr<FMIndex> pFmC,
                       const char* sOut, int iThreads, int iRepeat, size_t uiPrefetchBatch = 0 )
{
    auto pPack = std::make_shared<Pledge<Pack>>( );
    pPack->set( pPackC );
    auto pFMDIndex = std::make_shared<Pledge<FMIndex>>( );
    pFMDIndex->set( pFmC );
    auto pSai = std::make_shared<Pledge<SuffixArrayInterface>>( );
    pSai->set( pFmC );
    auto pPlainReader = std::make_shared<Reader>( c, iRepeat > 0 ? (size_t)iRepeat : 1 );
    // mode "prefetch": the reader node wrapped into a PrefetchReader (reads pulled ahead uiPrefetchBatch at a time, through all
    // stages on the GPU before any graph thread sees them); the rest of the graph is the same
    std::shared_ptr<libMS::Module<NucSeq, true>> pReader = pPlainReader;
    std::shared_ptr<PrefetchReader<>> pAhead;
    if( uiPrefetchBatch != 0 )
    {
        detail::PrefetchOptions xPO;
        xPO.uiBatchReads = uiPrefetchBatch;
        xPO.uiDepth = 2;
        xPO.bStages = defaultBatcherOptions( ).bStages;
        xPO.bSocQueues = defaultBatcherOptions( ).bSocQueues;
        pAhead = std::make_shared<PrefetchReader<>>( xParams, pPlainReader, pFmC, xPO );
        pReader = pAhead;
    }
    auto pSeeding = std::make_shared<BinarySeeding>( xParams );
    auto pSOC = std::make_shared<StripOfConsideration>( xParams );
    auto pHarmonization = std::make_shared<Harmonization>( xParams );
    auto pDP = std::make_shared<NeedlemanWunsch>( xParams );
    auto pMappingQual = std::make_shared<MappingQuality>( xParams );
    Collected xAll;
    auto pCollector = std::make_shared<Collector>( xAll, iRepeat == 0 );
    std::vector<std::shared_ptr<BasePledge>> vSinks;
    for( int t = 0; t < iThreads; t++ ) // one copy of the graph per thread, modules shared
    {
        auto pQueries = promiseMe( pReader );
        auto pQuery = promiseMe( std::make_shared<Lock<NucSeq>>( ), pQueries );
        auto pSeeds = promiseMe( pSeeding, pSai, pQuery );
        auto pSOCs = promiseMe( pSOC, pSeeds, pQuery, pPack, pFMDIndex );
        auto pHarmonized = promiseMe( pHarmonization, pSOCs, pQuery, pFMDIndex );
        auto pAlignments = promiseMe( pDP, pHarmonized, pQuery, pPack );
        auto pAlignmentsWQuality = promiseMe( pMappingQual, pQuery, pAlignments );
        auto pCollected = promiseMe( pCollector, pQuery, pAlignments, pAlignmentsWQuality );
        vSinks.push_back( promiseMe( std::make_shared<UnLock<Container>>( pQuery ), pCollected ) );
    }
    const auto t0 = std::chrono::steady_clock::now( );
    BasePledge::simultaneousGet( vSinks );
    const double fSec = std::chrono::duration<double>( std::chrono::steady_clock::now( ) - t0 ).count( );
    auto xStat = pSeeding->batchStatistics( );
    const unsigned long long uiFunnelled = xStat.second;
    if( pAhead != nullptr )
    {
        double fRun = 0, fPull = 0;
        pAhead->stats( xStat.first, xStat.second, fRun, fPull );
        if( uiFunnelled != 0 )
        {
            fprintf( stderr, "prefetch mode: %llu reads went through the per-read funnel\n", uiFunnelled );
            return 1;
        }
    }
    printf( "{\"graph_threads\": %d, \"reads\": %zu, \"aligned_reads\": %zu, \"seconds\": %.4f, \"reads_per_s\": %.1f, "
            "\"device_batches\": %llu, \"mean_reads_per_device_batch\": %.1f}\n",
            iThreads, pCollector->uiSeen.load( ), pCollector->uiAligned.load( ), fSec, pCollector->uiSeen.load( ) / fSec,
            (unsigned long long)xStat.first, xStat.first ? (double)xStat.second / xStat.first : 0.0 );
    if( iRepeat == 0 )
    {
        FILE* f = fopen( sOut, "w" );
        for( size_t k = 0; k < xAll.vQ.size( ); k++ )
            writeRecords( f, k, *xAll.vQ[ k ], *xAll.vA[ k ], xAll.vM[ k ].get( ) );
        fclose( f );
    }
    return 0;
}

// SoCPriorityQueue::pop across the boundary: SOC records like oracle/ref_dump.cpp writes them
static int runSocs( const CaseFile& c, const ParameterSetManager& xParams, std::shared_ptr<Pack> pPackC, std::shared_ptr<FMIndex> pFmC,
                    const char* sOut )
{
    BinarySeeding xSeeding( xParams );
    StripOfConsideration xSoc( xParams );
    FILE* f = fopen( sOut, "w" );
    for( size_t i = 0; i < c.reads.size( ); i++ )
    {
        auto pQ = std::make_shared<NucSeq>( );
        pQ->xCodes = c.reads[ i ];
        auto pSegs = xSeeding.execute( pFmC, pQ );
        auto pQueue = xSoc.execute( pSegs, pQ, pPackC, pFmC );
        fprintf( f, "R %zu %zu\n", i, c.reads[ i ].size( ) );
        fprintf( f, "SOC %zu\n", pQueue->size( ) );
        while( !pQueue->empty( ) )
        {
            auto xFront = pQueue->front( );
            auto p = pQueue->pop( );
            fprintf( f, "c %u %llu %u %zu\n", p->index_of_strip, (unsigned long long)xFront.first, xFront.second, p->size( ) );
            for( auto& x : *p )
                fprintf( f, "e %llu %llu %llu %u %d %llu\n", (unsigned long long)x.start( ), (unsigned long long)x.size( ),
                         (unsigned long long)x.start_ref( ), x.uiAmbiguity, (int)x.bOnForwStrand, (unsigned long long)x.uiDelta );
        }
    }
    fclose( f );
    return 0;
}

// MultiDeviceAligner with <shards> replicas of the index on device 0
static int runMulti( const CaseFile& c, const ParameterSetManager& xParams, std::shared_ptr<Pack> pPackC, std::shared_ptr<FMIndex> pFmC,
                     const char* sOut, int iShards )
{
    auto vReplicas = MultiDeviceAligner::replicate( pFmC, std::vector<int>( (size_t)iShards, 0 ), 0 );
    MultiDeviceAligner xAligner( xParams, vReplicas );
    xAligner.uiBatchReads = 37; // several ragged device batches per shard
    auto pQueries = std::make_shared<ContainerVector<std::shared_ptr<NucSeq>>>( );
    for( size_t i = 0; i < c.reads.size( ); i++ )
    {
        auto pQ = std::make_shared<NucSeq>( );
        pQ->xCodes = c.reads[ i ];
        pQ->sName = "r" + std::to_string( i );
        pQueries->push_back( pQ );
    }
    auto pRes = xAligner.execute( pQueries );
    FILE* f = fopen( sOut, "w" );
    for( size_t i = 0; i < c.reads.size( ); i++ )
        writeRecords( f, i, *( *pQueries )[ i ], *( *pRes )[ i ], nullptr );
    fclose( f );
    size_t uiReads = 0, uiBatches = 0, uiShardsUsed = 0;
    for( auto& t : xAligner.vLast )
        uiReads += t.uiReads, uiBatches += t.uiBatches, uiShardsUsed += t.uiBatches != 0;
    printf( "{\"shards\": %d, \"reads\": %zu, \"device_batches\": %zu, \"shards_used\": %zu}\n", iShards, uiReads, uiBatches, uiShardsUsed );
    return 0;
}

// MultiDeviceAligner::executeFlat (persistent engines): two runs, records of the second one
static int runMultiFlat( const CaseFile& c, const ParameterSetManager& xParams, std::shared_ptr<Pack> pPackC, std::shared_ptr<FMIndex> pFmC,
                         const char* sOut, int iShards )
{
    auto vReplicas = MultiDeviceAligner::replicate( pFmC, std::vector<int>( (size_t)iShards, 0 ), 0 );
    MultiDeviceAligner xAligner( xParams, vReplicas );
    xAligner.uiBatchReads = 37;
    xAligner.uiInflight = 2;
    auto pQueries = std::make_shared<ReadVector>( );
    for( size_t i = 0; i < c.reads.size( ); i++ )
    {
        auto pQ = std::make_shared<NucSeq>( );
        pQ->xCodes = c.reads[ i ];
        pQ->sName = "r" + std::to_string( i );
        pQueries->push_back( pQ );
    }
    auto pFirst = xAligner.executeFlat( pQueries );
    const uint64_t uiEnginesAfterFirst = detail::Engine::created( ).load( );
    const size_t uiBatchesFirst = pFirst->size( );
    pFirst.reset( );
    auto pFlat = xAligner.executeFlat( pQueries );
    const uint64_t uiEnginesAfterSecond = detail::Engine::created( ).load( );
    FILE* f = fopen( sOut, "w" );
    size_t uiAt = 0;
    for( const auto& pBatch : *pFlat )
    {
        if( pBatch == nullptr || pBatch->uiFirst != uiAt )
        {
            fprintf( stderr, "multiflat: the flat batches are not in input order\n" );
            return 1;
        }
        for( size_t i = 0; i < pBatch->size( ); i++ )
            writeRecords( f, uiAt + i, *pBatch->read( i ), *pBatch->alignmentsOf( i ), nullptr );
        uiAt += pBatch->size( );
    }
    fclose( f );
    size_t uiShardsUsed = 0;
    for( auto& t : xAligner.vLast )
        uiShardsUsed += t.uiBatches != 0;
    printf( "{\"shards\": %d, \"reads\": %zu, \"device_batches\": %zu, \"device_batches_first_run\": %zu, \"shards_used\": %zu, "
            "\"engines_after_first_run\": %llu, \"engines_after_second_run\": %llu}\n",
            iShards, uiAt, pFlat->size( ), uiBatchesFirst, uiShardsUsed, (unsigned long long)uiEnginesAfterFirst,
            (unsigned long long)uiEnginesAfterSecond );
    return uiAt == c.reads.size( ) && uiEnginesAfterFirst == uiEnginesAfterSecond ? 0 : 1;
}

// The throughput form as graph nodes (ma_batch_nodes.h): BatchFileReader (a FASTQ text of the case's reads on an in-memory
// stream, <batch> reads per call) -> BatchAlign -> BatchFileWriter, <threads> graph copies under simultaneousGet; the SAM
// text goes to <out>.  <options>: bit 0 soft clip, bit 1 =/X cigars, bit 2 NGMLR tags (the writer's per-read fall-back).
static int runBatchGraph( const CaseFile& c, ParameterSetManager xParams, std::shared_ptr<Pack> pPackC, std::shared_ptr<FMIndex> pFmC,
                          const char* sOut, int iThreads, size_t uiBatch, int iOptions )
{
    xParams.xSam.bSoftClip = ( iOptions & 1 ) != 0;
    xParams.xSam.bOutputMCigar = ( iOptions & 2 ) == 0;
    xParams.xSam.bEmulateNgmlrTags = ( iOptions & 4 ) != 0;
    std::string sFastq;
    for( size_t i = 0; i < c.reads.size( ); i++ )
    {
        sFastq += "@r" + std::to_string( i ) + " the description is dropped\n";
        for( uint8_t b : c.reads[ i ] )
            sFastq.push_back( "ACGTN"[ b < 4 ? b : 4 ] );
        sFastq += "\n+\n" + std::string( c.reads[ i ].size( ), 'I' ) + "\n";
    }
    auto pStream = std::make_shared<Pledge<FileStream>>( );
    pStream->set( std::make_shared<StringStream>( sFastq ) );
    auto pOut = std::make_shared<StringOutStream>( );
    auto pReader = std::make_shared<BatchFileReader>( xParams );
    pReader->uiBatchReads = uiBatch;
    auto pAlign = std::make_shared<BatchAlign>( xParams );
    auto pWriter = std::make_shared<BatchFileWriter>( xParams, std::static_pointer_cast<OutStream>( pOut ), pPackC );
    pWriter->uiFormatThreads = 3;
    auto pPack = std::make_shared<Pledge<Pack>>( );
    pPack->set( pPackC );
    auto pFMDIndex = std::make_shared<Pledge<FMIndex>>( );
    pFMDIndex->set( pFmC );
    std::vector<std::shared_ptr<BasePledge>> vSinks;
    for( int t = 0; t < iThreads; t++ )
    {
        auto pBatch = promiseMe( std::make_shared<Lock<ReadVector>>( ), promiseMe( pReader, pStream ) );
        auto pAligned = promiseMe( pAlign, pFMDIndex, pBatch );
        auto pWritten = promiseMe( pWriter, pBatch, pAligned, pPack );
        vSinks.push_back( promiseMe( std::make_shared<UnLock<Container>>( pBatch ), pWritten ) );
    }
    BasePledge::simultaneousGet( vSinks );
    FILE* f = fopen( sOut, "w" );
    fputs( pOut->sText.c_str( ), f );
    fclose( f );
    printf( "{\"graph_threads\": %d, \"reads\": %llu, \"device_batches\": %llu, \"aligned_reads\": %llu, \"sam_bytes\": %llu}\n", iThreads,
            (unsigned long long)pAlign->uiReads.load( ), (unsigned long long)pAlign->uiBatches.load( ),
            (unsigned long long)pAlign->uiAligned.load( ), (unsigned long long)pWriter->uiBytes.load( ) );
    return 0;
}

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
        if( argc >= 5 && !strcmp( argv[ 4 ], "nogpu" ) )
        {
            printf( "nogpu: got std::runtime_error as required: %s\n", e.what( ) );
            return 0;
        }
        fprintf( stderr, "error: %s\n", e.what( ) );
        return 1;
    }
    if( const char* e = getenv( "MA_TEST_REPLICAS" ) ) // virtual shards attached to the graph's one index (see the header)
        if( atoi( e ) > 1 )
            replicateIndex( pFmC->pDev, std::vector<int>( (size_t)atoi( e ) - 1, 0 ) );
    const std::string sMode = argc >= 5 ? argv[ 4 ] : "";
    if( sMode == "nogpu" )
    {
        fprintf( stderr, "expected a failure without a GPU\n" );
        return 1;
    }
    try
    {
        if( sMode == "threads" )
            return runThreads( c, xParams, pPackC, pFmC, argv[ 3 ], argc >= 6 ? atoi( argv[ 5 ] ) : 8, argc >= 7 ? atoi( argv[ 6 ] ) : 0 );
        if( sMode == "prefetch" ) // prefetch <threads> <reads per device batch pulled ahead>
            return runThreads( c, xParams, pPackC, pFmC, argv[ 3 ], argc >= 6 ? atoi( argv[ 5 ] ) : 8, 0, argc >= 7 ? (size_t)atoi( argv[ 6 ] ) : 64 );
        if( sMode == "socs" )
            return runSocs( c, xParams, pPackC, pFmC, argv[ 3 ] );
        if( sMode == "multi" )
            return runMulti( c, xParams, pPackC, pFmC, argv[ 3 ], argc >= 6 ? atoi( argv[ 5 ] ) : 2 );
        if( sMode == "multiflat" )
            return runMultiFlat( c, xParams, pPackC, pFmC, argv[ 3 ], argc >= 6 ? atoi( argv[ 5 ] ) : 2 );
        if( sMode == "batchgraph" )
            return runBatchGraph( c, xParams, pPackC, pFmC, argv[ 3 ], argc >= 6 ? atoi( argv[ 5 ] ) : 2, argc >= 7 ? (size_t)atoi( argv[ 6 ] ) : 50,
                                  argc >= 8 ? atoi( argv[ 7 ] ) : 0 );
    }
    catch( const std::exception& e )
    {
        fprintf( stderr, "error: %s\n", e.what( ) );
        return 1;
    }
    const bool bMixed = sMode == "mixed";
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
    auto pSeeds0 = promiseMe( pSeeding, pSai, pQuery );
    auto pSeeds = promiseMe( std::make_shared<Detach<SegmentVector>>( bMixed ), pSeeds0 );
    auto pSOCs0 = promiseMe( pSOC, pSeeds, pQuery, pPack, pFMDIndex );
    auto pSOCs = promiseMe( std::make_shared<Detach<SoCPriorityQueue>>( bMixed ), pSOCs0 );
    auto pHarmonized0 = promiseMe( pHarmonization, pSOCs, pQuery, pFMDIndex );
    auto pHarmonized = promiseMe( std::make_shared<Detach<SeedsSetVector>>( bMixed ), pHarmonized0 );
    auto pAlignments = promiseMe( pDP, pHarmonized, pQuery, pPack );
    auto pAlignmentsWQuality = promiseMe( pMappingQual, pQuery, pAlignments );
    auto pWritten = promiseMe( pWriter, pQuery, pAlignments, pAlignmentsWQuality );
    // the SAM writer of export.cpp:109-117 on an in-memory stream (argv[3] + ".sam")
    auto pSamStream = std::make_shared<StringOutStream>( );
    auto pSamWriter = std::make_shared<FileWriter>( xParams, std::static_pointer_cast<OutStream>( pSamStream ), pPackC );
    auto pSamWritten = promiseMe( pSamWriter, pQuery, pAlignmentsWQuality, pPack );
    auto pBoth = promiseMe( std::make_shared<Join2>( ), pWritten, pSamWritten );
    auto pSink = promiseMe( std::make_shared<UnLock<Container>>( pQuery ), pBoth ); // export.cpp:122-124
    BasePledge::simultaneousGet( std::vector<std::shared_ptr<BasePledge>>{ pSink } );
    fclose( f );
    {
        FILE* fs = fopen( ( std::string( argv[ 3 ] ) + ".sam" ).c_str( ), "w" );
        fputs( pSamStream->sText.c_str( ), fs );
        fclose( fs );
    }
    return 0;
}
