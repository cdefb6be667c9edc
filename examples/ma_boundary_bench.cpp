// examples/ma_boundary_bench.cpp -- host-fed end-to-end throughput through the drop-in boundary (what
// ExecutionContext::doAlign, libs/ma/inc/ma/util/execution-context.h:291-406, does with the reference's modules):
// reads in HOST memory -> H2D -> all stages on the GPU -> D2H -> Alignment containers -> SAM text.
// Three legs over the same synthetic 150 bp reads of the indexed genome:
//   batch_aligner   BatchAligner::execute, device batches of 256 k reads, 1 and 2 in flight; wall + the time of each phase
//   sam             FileWriter::execute for every read on T host threads into a counting sink
//   graph           the UNCHANGED per-read graph of setUpCompGraph (export.cpp:99-126), T graph threads over one shared
//                   reader, per-read execute() calls funnelled into device batches by the DeviceBatcher
// Prints one JSON line.  bench.py runs it after its timed regions and reports it as config.boundary (never as `value`).
//
//   ma_boundary_bench <index prefix> <reads> <read length> <preset> <device> [graph threads]
//
// build: g++ -std=c++17 -O2 -Iinclude -Ima_amd/host examples/ma_boundary_bench.cpp -Lma_amd -lma_amd -lpthread
#include "ma_batch_nodes.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace libMA;
using namespace libMS;
typedef ContainerVector<std::shared_ptr<NucSeq>> ReadVec;

static double now( )
{
    return std::chrono::duration<double>( std::chrono::steady_clock::now( ).time_since_epoch( ) ).count( );
}

struct CountingSink : public OutStream
{
    std::atomic<uint64_t> uiBytes{ 0 };
    void put( const char*, size_t n ) override
    {
        uiBytes += n;
    }
    void write( const char*, size_t n ) override
    {
        uiBytes += n;
    }
};

class VecReader : public Module<NucSeq, true>
{
  public:
    const ReadVec& r;
    std::atomic<size_t> i{ 0 };
    VecReader( const ReadVec& r ) : r( r )
    {}
    std::shared_ptr<NucSeq> execute( ) override
    {
        const size_t k = i++;
        return k < r.size( ) ? r[ k ] : nullptr;
    }
};

// volatile source that hands out a NEW NucSeq per call, as a file reader does (FileReader::execute, fileReader.cpp:37-203)
class FreshReader : public Module<NucSeq, true>
{
  public:
    const ReadVec& r;
    std::atomic<size_t> i{ 0 };
    FreshReader( const ReadVec& r ) : r( r )
    {}
    std::shared_ptr<NucSeq> execute( ) override
    {
        const size_t k = i++;
        return k < r.size( ) ? std::make_shared<NucSeq>( *r[ k ] ) : nullptr;
    }
};

// execute( ) of a module, timed (thread time summed over all graph threads): where the host side of the per-read graph goes
template <class TP_MODULE> struct Timed : public TP_MODULE
{
    std::atomic<uint64_t> uiNs{ 0 }, uiCalls{ 0 };
    template <typename... A> Timed( A&&... a ) : TP_MODULE( std::forward<A>( a )... )
    {}
    template <typename F> auto timed( F&& f ) -> decltype( f( ) )
    {
        const auto t0 = std::chrono::steady_clock::now( );
        auto r = f( );
        uiNs += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>( std::chrono::steady_clock::now( ) - t0 ).count( );
        uiCalls++;
        return r;
    }
    double usPerCall( ) const
    {
        return uiCalls ? uiNs / 1e3 / uiCalls : 0.0;
    }
};
struct TReader : public Timed<PrefetchReader<>>
{
    using Timed<PrefetchReader<>>::Timed;
    std::shared_ptr<NucSeq> execute( ) override
    {
        return timed( [ & ]( ) { return PrefetchReader<>::execute( ); } );
    }
};
struct TSeeding : public Timed<BinarySeeding>
{
    using Timed<BinarySeeding>::Timed;
    std::shared_ptr<SegmentVector> execute( std::shared_ptr<SuffixArrayInterface> a, std::shared_ptr<NucSeq> b ) override
    {
        return timed( [ & ]( ) { return BinarySeeding::execute( a, b ); } );
    }
};
struct TSoc : public Timed<StripOfConsideration>
{
    using Timed<StripOfConsideration>::Timed;
    std::shared_ptr<SoCPriorityQueue> execute( std::shared_ptr<SegmentVector> a, std::shared_ptr<NucSeq> b, std::shared_ptr<Pack> c,
                                               std::shared_ptr<FMIndex> d ) override
    {
        return timed( [ & ]( ) { return StripOfConsideration::execute( a, b, c, d ); } );
    }
};
struct THarm : public Timed<Harmonization>
{
    using Timed<Harmonization>::Timed;
    std::shared_ptr<SeedsSetVector> execute( std::shared_ptr<SoCPriorityQueue> a, std::shared_ptr<NucSeq> b, std::shared_ptr<FMIndex> c ) override
    {
        return timed( [ & ]( ) { return Harmonization::execute( a, b, c ); } );
    }
};
typedef libMS::ContainerVector<std::shared_ptr<Alignment>> AlnVec;
struct TDp : public Timed<NeedlemanWunsch>
{
    using Timed<NeedlemanWunsch>::Timed;
    std::shared_ptr<AlnVec> execute( std::shared_ptr<SeedsSetVector> a, std::shared_ptr<NucSeq> b, std::shared_ptr<Pack> c ) override
    {
        return timed( [ & ]( ) { return NeedlemanWunsch::execute( a, b, c ); } );
    }
};
struct TMq : public Timed<MappingQuality>
{
    using Timed<MappingQuality>::Timed;
    std::shared_ptr<AlnVec> execute( std::shared_ptr<NucSeq> a, std::shared_ptr<AlnVec> b ) override
    {
        return timed( [ & ]( ) { return MappingQuality::execute( a, b ); } );
    }
};
struct TWriter : public Timed<FileWriter>
{
    using Timed<FileWriter>::Timed;
    std::shared_ptr<libMS::Container> execute( std::shared_ptr<NucSeq> a, std::shared_ptr<AlnVec> b, std::shared_ptr<Pack> c ) override
    {
        return timed( [ & ]( ) { return FileWriter::execute( a, b, c ); } );
    }
};

// ---- leg 4: the per-read graph of export.cpp:99-126 with the reader node wrapped into a PrefetchReader: reads are pulled ahead
// a device batch at a time and every graph thread gets reads that are already aligned -- a few dozen graph threads
static std::string prefetchLeg( const ParameterSetManager& xParams, std::shared_ptr<ReadVec> pReads, std::shared_ptr<FMIndex> pFM,
                                std::shared_ptr<Pack> pPack, uint64_t uiExpectedSamBytes, double& fPrefetchBest, int& iPrefetchBestThreads )
{
    const size_t n = pReads->size( );
    auto now = []( ) { return std::chrono::duration<double>( std::chrono::steady_clock::now( ).time_since_epoch( ) ).count( ); };
    // Every graph copy gets its own handles of the shared constants (Pack, FMIndex): same objects, but behind a control block
    // of the copy's own.  A constant pledge hands out a shared_ptr copy per get( ) -- six per read -- and with ONE control block
    // for all threads those reference counts bounce a cache line between the cores (measured: ~15 us of thread time per read).
    auto own = []( auto p ) {
        typedef typename decltype( p )::element_type T;
        return std::shared_ptr<T>( p.get( ), [ p ]( T* ) {} );
    };
    double t0 = 0;
    std::string sPrefetch;
    for( int iT : { 8, 16, 32 } )
    {
        detail::PrefetchOptions xPO;
        xPO.uiBatchReads = 1u << 16;
        xPO.uiDepth = 2;
        // (MA_BOUNDARY_SHARED_READS=1: the source hands out the caller's own objects, which PrefetchReader then has to copy)
        std::shared_ptr<Module<NucSeq, true>> pSource;
        if( getenv( "MA_BOUNDARY_SHARED_READS" ) )
            pSource = std::make_shared<VecReader>( *pReads );
        else
            pSource = std::make_shared<FreshReader>( *pReads );
        if( getenv( "MA_PREFETCH_BATCH" ) )
            xPO.uiBatchReads = (size_t)atoi( getenv( "MA_PREFETCH_BATCH" ) );
        if( getenv( "MA_PREFETCH_DEPTH" ) )
            xPO.uiDepth = (size_t)atoi( getenv( "MA_PREFETCH_DEPTH" ) );
        auto pAhead = std::make_shared<TReader>( xParams, pSource, pFM, xPO );
        auto pSeeding4 = std::make_shared<TSeeding>( xParams );
        auto pSoc4 = std::make_shared<TSoc>( xParams );
        auto pHarm4 = std::make_shared<THarm>( xParams );
        auto pDp4 = std::make_shared<TDp>( xParams );
        auto pMq4 = std::make_shared<TMq>( xParams );
        auto pSink4 = std::make_shared<CountingSink>( );
        auto pWriter4 = std::make_shared<TWriter>( xParams, std::static_pointer_cast<OutStream>( pSink4 ), pPack );
        if( !getenv( "MA_BOUNDARY_WRITER_UNBUFFERED" ) )
            pWriter4->uiBufferBytes = 1u << 16; // per-thread buffers: the writer's lock once per 64 KB instead of once per read
        std::vector<std::shared_ptr<BasePledge>> vSinks4;
        for( int t = 0; t < iT; t++ )
        {
            auto pPackP = std::make_shared<Pledge<Pack>>( );
            pPackP->set( own( pPack ) );
            auto pFmP = std::make_shared<Pledge<FMIndex>>( );
            pFmP->set( own( pFM ) );
            auto pSai = std::make_shared<Pledge<SuffixArrayInterface>>( );
            pSai->set( own( std::static_pointer_cast<SuffixArrayInterface>( pFM ) ) );
            auto pQuery = promiseMe( std::make_shared<Lock<NucSeq>>( ), promiseMe( std::static_pointer_cast<PrefetchReader<>>( pAhead ) ) );
            auto pSeeds = promiseMe( std::static_pointer_cast<BinarySeeding>( pSeeding4 ), pSai, pQuery );
            auto pSOCs = promiseMe( std::static_pointer_cast<StripOfConsideration>( pSoc4 ), pSeeds, pQuery, pPackP, pFmP );
            auto pHarmonized = promiseMe( std::static_pointer_cast<Harmonization>( pHarm4 ), pSOCs, pQuery, pFmP );
            auto pAlignments = promiseMe( std::static_pointer_cast<NeedlemanWunsch>( pDp4 ), pHarmonized, pQuery, pPackP );
            auto pWithQuality = promiseMe( std::static_pointer_cast<MappingQuality>( pMq4 ), pQuery, pAlignments );
            auto pWritten = promiseMe( std::static_pointer_cast<FileWriter>( pWriter4 ), pQuery, pWithQuality, pPackP );
            vSinks4.push_back( promiseMe( std::make_shared<UnLock<Container>>( pQuery ), pWritten ) );
        }
        t0 = now( );
        BasePledge::simultaneousGet( vSinks4 );
        pWriter4->flush( );
        const double f4 = now( ) - t0;
        uint64_t uiB = 0, uiR = 0;
        double fRun4 = 0, fPull4 = 0;
        pAhead->stats( uiB, uiR, fRun4, fPull4 );
        if( pSeeding4->batcher( ) != nullptr )
            throw std::runtime_error( "prefetch leg: a read went through the per-read funnel" );
        char buf[ 640 ];
        snprintf( buf, sizeof( buf ), "%s\"graph_threads_%d\": {\"reads_per_s\": %.1f, \"wall_s\": %.4f, \"device_batches\": %llu, "
                                      "\"device_batch_s\": %.4f, \"pull_s\": %.4f, \"sam_bytes\": %llu, \"us_per_read\": {\"thread_time\": %.2f, "
                                      "\"reader\": %.2f, \"seeding\": %.2f, \"soc\": %.2f, \"harmonization\": %.2f, \"dp\": %.2f, \"mapping_quality\": %.2f, "
                                      "\"writer\": %.2f}}",
                  sPrefetch.empty( ) ? "" : ", ", iT, n / f4, f4, (unsigned long long)uiB, fRun4, fPull4, (unsigned long long)pSink4->uiBytes.load( ),
                  f4 * iT * 1e6 / n, pAhead->usPerCall( ), pSeeding4->usPerCall( ), pSoc4->usPerCall( ), pHarm4->usPerCall( ), pDp4->usPerCall( ),
                  pMq4->usPerCall( ), pWriter4->usPerCall( ) );
        sPrefetch += buf;
        if( uiExpectedSamBytes != 0 && pSink4->uiBytes.load( ) != uiExpectedSamBytes )
            throw std::runtime_error( "prefetch leg: SAM bytes differ from the funnel leg's" );
        if( n / f4 > fPrefetchBest )
            fPrefetchBest = n / f4, iPrefetchBestThreads = iT;
    }
    return sPrefetch;
}

int main( int argc, char** argv )
{
    if( argc < 6 )
    {
        fprintf( stderr, "usage: ma_boundary_bench <index prefix> <reads> <read length> <preset> <device> [graph threads]\n" );
        return 2;
    }
    try
    {
        const size_t n = (size_t)atoll( argv[ 2 ] ), uiLen = (size_t)atoll( argv[ 3 ] );
        ParameterSetManager xParams;
        xParams.setSelected( argv[ 4 ] );
        maCheck( ma_set_device( atoi( argv[ 5 ] ) ) );
        maCheck( ma_host_bind_thread( atoi( argv[ 5 ] ), 0, nullptr ) ); // this thread and all it starts: the CPUs next to the GPU
        const unsigned uiHw = std::max( 1u, std::thread::hardware_concurrency( ) );
        const int iGraphThreads = argc >= 7 ? atoi( argv[ 6 ] ) : (int)std::min( 2048u, 8 * uiHw );
        std::shared_ptr<Pack> pPack;
        std::shared_ptr<FMIndex> pFM;
        double t0 = now( );
        loadIndex( argv[ 1 ], pPack, pFM );
        const double fLoad = now( ) - t0;
        // ---- reads: sampled on the host from the packed forward strand, 0.5 % substitutions, every second one reverse
        uint64_t uiN = 0;
        maCheck( ma_index_sizes( pFM->pDev->p, nullptr, nullptr, &uiN, nullptr ) );
        const uint64_t uiF = uiN / 2;
        std::vector<uint8_t> vPac( ( uiF + 3 ) / 4 + 1 );
        maCheck( ma_index_download( pFM->pDev->p, nullptr, nullptr, nullptr, nullptr, vPac.data( ), nullptr, nullptr ) );
        auto pReads = std::make_shared<ReadVec>( );
        uint64_t uiState = 0x9E3779B97F4A7C15ull;
        auto rnd = [ & ]( ) {
            uiState ^= uiState << 13, uiState ^= uiState >> 7, uiState ^= uiState << 17;
            return uiState;
        };
        for( size_t i = 0; i < n; i++ )
        {
            auto pQ = std::make_shared<NucSeq>( );
            pQ->sName = "r" + std::to_string( i );
            pQ->xCodes.resize( uiLen );
            const uint64_t uiPos = rnd( ) % ( uiF - uiLen );
            for( size_t j = 0; j < uiLen; j++ )
            {
                const uint64_t p = uiPos + j;
                uint8_t b = ( vPac[ p >> 2 ] >> ( ( ~p & 3 ) << 1 ) ) & 3;
                if( rnd( ) % 200 == 0 )
                    b = ( b + 1 + rnd( ) % 3 ) & 3;
                pQ->xCodes[ j ] = b;
            }
            if( i & 1 )
            {
                std::reverse( pQ->xCodes.begin( ), pQ->xCodes.end( ) );
                for( auto& b : pQ->xCodes )
                    b = 3 - b;
            }
            pReads->push_back( pQ );
        }
        std::vector<uint8_t>( ).swap( vPac );
        if( getenv( "MA_BOUNDARY_ONLY_GRAPH" ) ) // diagnostics: the per-read graph with the prefetching reader alone
        {
            defaultBatcherOptions( ).bStages = false;
            double fBest = 0;
            int iBest = 0;
            const std::string sOnly = prefetchLeg( xParams, pReads, pFM, pPack, 0, fBest, iBest );
            printf( "{\"graph\": {\"reads_per_s\": %.1f, \"threads\": %d, %s}}\n", fBest, iBest, sOnly.c_str( ) );
            return 0;
        }
        // ---- leg 1: BatchAligner
        std::string sBatch;
        std::shared_ptr<BatchAligner::TP_RESULT> pRes;
        for( size_t uiInflight : { (size_t)1, (size_t)2 } )
        {
            BatchAligner xAligner( xParams );
            xAligner.uiInflight = uiInflight;
            xAligner.execute( pFM, pReads ); // warm-up: every engine allocates its device pools and page-locked staging once
            pRes = xAligner.execute( pFM, pReads );
            const AlignerTiming& T = xAligner.xLast;
            char buf[ 512 ];
            snprintf( buf, sizeof( buf ),
                      "%s\"inflight_%zu\": {\"reads_per_s\": %.1f, \"wall_s\": %.4f, \"h2d_s\": %.4f, \"kernels_s\": %.4f, \"d2h_s\": %.4f, "
                      "\"containers_s\": %.4f, \"device_batches\": %llu, \"aligned_reads\": %llu}",
                      sBatch.empty( ) ? "" : ", ", uiInflight, n / T.fWall, T.fWall, T.fH2D, T.fKernels, T.fD2H, T.fContainers,
                      (unsigned long long)T.uiBatches, (unsigned long long)T.uiAlignedReads );
            sBatch += buf;
        }
        // ---- leg 1b: the same with the results left flat (no Alignment containers), 1 .. 4 device batches in flight
        std::string sFlat;
        std::shared_ptr<BatchAligner::TP_FLAT> pFlat;
        for( size_t uiInflight : { (size_t)1, (size_t)2, (size_t)3, (size_t)4 } )
        {
            BatchAligner xAligner( xParams );
            xAligner.uiInflight = uiInflight;
            xAligner.executeFlat( pFM, pReads ); // warm-up: engines size their buffers, page-locked arenas are touched
            pFlat = xAligner.executeFlat( pFM, pReads );
            const AlignerTiming& T = xAligner.xLast;
            char buf[ 512 ];
            snprintf( buf, sizeof( buf ),
                      "%s\"inflight_%zu\": {\"reads_per_s\": %.1f, \"wall_s\": %.4f, \"gather_s\": %.4f, \"h2d_s\": %.4f, \"kernels_s\": %.4f, "
                      "\"d2h_s\": %.4f, \"device_batches\": %llu, \"aligned_reads\": %llu, \"slowest_batch_over_median\": %.2f}",
                      sFlat.empty( ) ? "" : ", ", uiInflight, n / T.fWall, T.fWall, T.fPack, T.fH2D, T.fKernels, T.fD2H,
                      (unsigned long long)T.uiBatches, (unsigned long long)T.uiAlignedReads, T.maxOverMedian( ) );
            sFlat += buf;
            // engines before admission: no batch of a timed leg may pay an engine's allocations (round 3: 2.9 s inside a 0.06 s leg)
            if( T.maxOverMedian( ) > 3.0 )
                fprintf( stderr, "WARNING: flat leg with %zu in flight: slowest device batch %.1fx the median\n", uiInflight, T.maxOverMedian( ) );
        }
        // ---- leg 1c: ONE process, several index replicas: MultiDeviceAligner::executeFlat with persistent engines.
        // On a node every replica is another GPU; on a one-GPU box the second replica is a "virtual shard" on the same device,
        // which must cost nothing: 2 replicas x 2 batches in flight against 1 replica x 4 in flight.
        std::string sMulti;
        {
            int nDev = 1;
            maCheck( ma_device_count( &nDev ) );
            const int iHere = atoi( argv[ 5 ] );
            double fOne = 0;
            for( int iReplicas : { 1, 2 } )
            {
                std::vector<int> vDevices;
                for( int g = 0; g < iReplicas; g++ )
                    vDevices.push_back( nDev >= iReplicas ? ( iHere + g ) % nDev : iHere );
                auto vReplicas = MultiDeviceAligner::replicate( pFM, vDevices, iHere );
                MultiDeviceAligner xMulti( xParams, vReplicas );
                xMulti.uiInflight = iReplicas == 1 ? 4 : 2;
                xMulti.warmUp( pReads );
                xMulti.executeFlat( pReads );
                const uint64_t uiEngines = detail::Engine::created( ).load( );
                auto pOut = xMulti.executeFlat( pReads );
                const AlignerTiming& T = xMulti.xLast;
                if( iReplicas == 1 )
                    fOne = n / T.fWall;
                char buf[ 512 ];
                snprintf( buf, sizeof( buf ),
                          "%s\"replicas_%d\": {\"reads_per_s\": %.1f, \"wall_s\": %.4f, \"inflight_per_replica\": %zu, \"devices\": \"%s\", "
                          "\"device_batches\": %llu, \"batches_of_last_replica\": %llu, \"engines_created_by_the_timed_run\": %llu, "
                          "\"over_one_replica\": %.3f, \"slowest_batch_over_median\": %.2f}",
                          sMulti.empty( ) ? "" : ", ", iReplicas, n / T.fWall, T.fWall, xMulti.uiInflight,
                          nDev >= iReplicas ? "distinct" : "virtual shards on one device", (unsigned long long)T.uiBatches,
                          (unsigned long long)xMulti.vLast.back( ).uiBatches,
                          (unsigned long long)( detail::Engine::created( ).load( ) - uiEngines ), fOne > 0 ? ( n / T.fWall ) / fOne : 0.0,
                          T.maxOverMedian( ) );
                sMulti += buf;
            }
        }
        // ---- leg 2b: SAM text of the flat batches (BatchFileWriter: arenas, one write per batch)
        double fSamFlat = 0;
        uint64_t uiSamFlatBytes = 0;
        {
            auto pSinkF = std::make_shared<CountingSink>( );
            BatchFileWriter xBatchWriter( xParams, std::static_pointer_cast<OutStream>( pSinkF ), pPack );
            xBatchWriter.uiFormatThreads = std::min( 32u, uiHw );
            t0 = now( );
            for( auto& pB : *pFlat )
            {
                auto pSlice = std::make_shared<ReadVec>( pReads->begin( ) + pB->uiFirst, pReads->begin( ) + pB->uiFirst + pB->size( ) );
                xBatchWriter.execute( pSlice, pB, pPack );
            }
            fSamFlat = now( ) - t0;
            uiSamFlatBytes = pSinkF->uiBytes.load( );
        }
        pFlat.reset( );
        // ---- leg 2c: the throughput path AS GRAPH NODES: source -> BatchAlign -> BatchFileWriter, T graph threads under
        // simultaneousGet (each thread = one device batch in flight), reads in host memory -> SAM bytes in the sink
        std::string sBatchGraph;
        for( int iT : { 2, 3, 4 } )
        {
            auto pSinkG = std::make_shared<CountingSink>( );
            auto pSource = std::make_shared<BatchSource>( pReads );
            pSource->uiBatchReads = 1u << 17;
            auto pAlign = std::make_shared<BatchAlign>( xParams );
            auto pBatchWriter = std::make_shared<BatchFileWriter>( xParams, std::static_pointer_cast<OutStream>( pSinkG ), pPack );
            pBatchWriter->uiFormatThreads = std::min( 16u, uiHw );
            auto pPackP = std::make_shared<Pledge<Pack>>( );
            pPackP->set( pPack );
            auto pFmP = std::make_shared<Pledge<FMIndex>>( );
            pFmP->set( pFM );
            std::vector<std::shared_ptr<BasePledge>> vSinks;
            for( int t = 0; t < iT; t++ )
            {
                auto pBatch = promiseMe( std::make_shared<Lock<ReadVec>>( ), promiseMe( pSource ) );
                auto pAligned = promiseMe( pAlign, pFmP, pBatch );
                auto pWritten = promiseMe( pBatchWriter, pBatch, pAligned, pPackP );
                vSinks.push_back( promiseMe( std::make_shared<UnLock<Container>>( pBatch ), pWritten ) );
            }
            {
                // warm-up: iT engines at once, each allocates its device pools and page-locked staging (they stay with pAlign)
                std::vector<std::thread> vWarm;
                for( int t = 0; t < iT; t++ )
                    vWarm.emplace_back( [ & ]( ) {
                        pAlign->execute( pFM, std::make_shared<ReadVec>( pReads->begin( ), pReads->begin( ) + std::min<size_t>( n, 1u << 17 ) ) );
                    } );
                for( auto& rT : vWarm )
                    rT.join( );
                pAlign->uiBatches = 0;
            }
            t0 = now( );
            BasePledge::simultaneousGet( vSinks );
            const double f = now( ) - t0;
            char buf[ 384 ];
            snprintf( buf, sizeof( buf ), "%s\"graph_threads_%d\": {\"reads_per_s\": %.1f, \"wall_s\": %.4f, \"device_batches\": %llu, \"sam_bytes\": %llu}",
                      sBatchGraph.empty( ) ? "" : ", ", iT, n / f, f, (unsigned long long)pAlign->uiBatches.load( ),
                      (unsigned long long)pSinkG->uiBytes.load( ) );
            sBatchGraph += buf;
        }
        // ---- leg 2d: the doAlign shape (execution-context.h:291-406) end to end: FASTQ TEXT -> BatchFileReader -> BatchAlign ->
        // BatchFileWriter -> SAM bytes, graph threads under simultaneousGet.  The text lives in memory (no disk in the timing).
        std::string sFastqGraph;
        {
            std::string sFastq;
            sFastq.reserve( n * ( 2 * uiLen + 24 ) );
            for( size_t i = 0; i < n; i++ )
            {
                const NucSeq& rQ = *( *pReads )[ i ];
                sFastq.push_back( '@' );
                sFastq += rQ.sName;
                sFastq.push_back( '\n' );
                for( uint8_t b : rQ.xCodes )
                    sFastq.push_back( "ACGTN"[ b < 4 ? b : 4 ] );
                sFastq += "\n+\n";
                sFastq.append( rQ.xCodes.size( ), 'F' );
                sFastq.push_back( '\n' );
            }
            for( int iT : { 3, 4, 6 } )
            {
                auto pSinkG = std::make_shared<CountingSink>( );
                auto pStream = std::make_shared<Pledge<FileStream>>( );
                pStream->set( std::make_shared<StringStream>( sFastq ) );
                auto pBatchReader = std::make_shared<BatchFileReader>( xParams );
                pBatchReader->uiBatchReads = 1u << 17;
                auto pAlign = std::make_shared<BatchAlign>( xParams );
                auto pBatchWriter = std::make_shared<BatchFileWriter>( xParams, std::static_pointer_cast<OutStream>( pSinkG ), pPack );
                pBatchWriter->uiFormatThreads = std::min( 16u, uiHw );
                auto pPackP = std::make_shared<Pledge<Pack>>( );
                pPackP->set( pPack );
                auto pFmP = std::make_shared<Pledge<FMIndex>>( );
                pFmP->set( pFM );
                std::vector<std::shared_ptr<BasePledge>> vSinks;
                for( int t = 0; t < iT; t++ )
                {
                    auto pBatch = promiseMe( std::make_shared<Lock<ReadVec>>( ), promiseMe( pBatchReader, pStream ) );
                    auto pAligned = promiseMe( pAlign, pFmP, pBatch );
                    auto pWritten = promiseMe( pBatchWriter, pBatch, pAligned, pPackP );
                    vSinks.push_back( promiseMe( std::make_shared<UnLock<Container>>( pBatch ), pWritten ) );
                }
                {
                    std::vector<std::thread> vWarm;
                    for( int t = 0; t < iT; t++ )
                        vWarm.emplace_back( [ & ]( ) {
                            pAlign->execute( pFM, std::make_shared<ReadVec>( pReads->begin( ), pReads->begin( ) + std::min<size_t>( n, 1u << 17 ) ) );
                        } );
                    for( auto& rT : vWarm )
                        rT.join( );
                    pAlign->uiBatches = 0;
                }
                t0 = now( );
                BasePledge::simultaneousGet( vSinks );
                const double f = now( ) - t0;
                char buf[ 384 ];
                snprintf( buf, sizeof( buf ), "%s\"graph_threads_%d\": {\"reads_per_s\": %.1f, \"wall_s\": %.4f, \"device_batches\": %llu, \"fastq_bytes\": %zu, \"sam_bytes\": %llu}",
                          sFastqGraph.empty( ) ? "" : ", ", iT, n / f, f, (unsigned long long)pAlign->uiBatches.load( ), sFastq.size( ),
                          (unsigned long long)pSinkG->uiBytes.load( ) );
                sFastqGraph += buf;
            }
        }
        // ---- leg 2: SAM text of every read
        auto pSink = std::make_shared<CountingSink>( );
        FileWriter xWriter( xParams, std::static_pointer_cast<OutStream>( pSink ), pPack );
        const unsigned uiSamThreads = std::min( 64u, uiHw );
        t0 = now( );
        {
            std::atomic<size_t> uiNext{ 0 };
            std::vector<std::thread> vT;
            for( unsigned t = 0; t < uiSamThreads; t++ )
                vT.emplace_back( [ & ]( ) {
                    for( size_t i = uiNext++; i < n; i = uiNext++ )
                        xWriter.execute( ( *pReads )[ i ], ( *pRes )[ i ], pPack );
                } );
            for( auto& t : vT )
                t.join( );
        }
        const double fSam = now( ) - t0;
        pRes.reset( );
        // ---- leg 3: the unchanged per-read graph with many graph threads
        auto pPackP = std::make_shared<Pledge<Pack>>( );
        pPackP->set( pPack );
        auto pFmP = std::make_shared<Pledge<FMIndex>>( );
        pFmP->set( pFM );
        auto pSai = std::make_shared<Pledge<SuffixArrayInterface>>( );
        pSai->set( pFM );
        // the chain of export.cpp:104-108 only hands its intermediate containers from one MI355X module to the next
        defaultBatcherOptions( ).bStages = false;
        auto pReader = std::make_shared<VecReader>( *pReads );
        auto pSeeding = std::make_shared<BinarySeeding>( xParams );
        auto pSOC = std::make_shared<StripOfConsideration>( xParams );
        auto pHarm = std::make_shared<Harmonization>( xParams );
        auto pDP = std::make_shared<NeedlemanWunsch>( xParams );
        auto pMq = std::make_shared<MappingQuality>( xParams );
        auto pSink2 = std::make_shared<CountingSink>( );
        auto pWriter = std::make_shared<FileWriter>( xParams, std::static_pointer_cast<OutStream>( pSink2 ), pPack );
        std::vector<std::shared_ptr<BasePledge>> vSinks;
        for( int t = 0; t < iGraphThreads; t++ )
        {
            auto pQuery = promiseMe( std::make_shared<Lock<NucSeq>>( ), promiseMe( pReader ) );
            auto pSeeds = promiseMe( pSeeding, pSai, pQuery );
            auto pSOCs = promiseMe( pSOC, pSeeds, pQuery, pPackP, pFmP );
            auto pHarmonized = promiseMe( pHarm, pSOCs, pQuery, pFmP );
            auto pAlignments = promiseMe( pDP, pHarmonized, pQuery, pPackP );
            auto pWithQuality = promiseMe( pMq, pQuery, pAlignments );
            auto pWritten = promiseMe( pWriter, pQuery, pWithQuality, pPackP );
            vSinks.push_back( promiseMe( std::make_shared<UnLock<Container>>( pQuery ), pWritten ) );
        }
        t0 = now( );
        BasePledge::simultaneousGet( vSinks );
        const double fGraph = now( ) - t0;
        auto xStat = pSeeding->batchStatistics( );
        double fRun = 0, fUp = 0, fKern = 0, fDown = 0;
        if( pSeeding->batcher( ) != nullptr )
            pSeeding->batcher( )->phaseSeconds( fRun, fUp, fKern, fDown );
        double aStage[ 4 ] = { 0, 0, 0, 0 };
        if( pSeeding->batcher( ) != nullptr )
            pSeeding->batcher( )->stageSeconds( aStage );
        fprintf( stderr, "graph leg: %.3f s wall; device batches: run %.3f s (h2d %.3f, kernels %.3f = seed %.3f + extract %.3f + chain %.3f + dp %.3f, d2h %.3f) summed over %llu batches\n",
                 fGraph, fRun, fUp, fKern, aStage[ 0 ], aStage[ 1 ], aStage[ 2 ], aStage[ 3 ], fDown, (unsigned long long)xStat.first );
        double fPrefetchBest = 0;
        int iPrefetchBestThreads = 0;
        const std::string sPrefetch = prefetchLeg( xParams, pReads, pFM, pPack, pSink2->uiBytes.load( ), fPrefetchBest, iPrefetchBestThreads );
        printf( "{\"reads\": %zu, \"read_len\": %zu, \"host_threads\": %u, \"index_load_s\": %.2f, "
                "\"batch_aligner\": {%s, \"what\": \"reads in host memory -> BatchAligner::execute (H2D, all stages, D2H, Alignment "
                "containers); 256 k reads per device batch; phase times summed over the batches\"}, "
                "\"batch_aligner_flat\": {%s, \"what\": \"BatchAligner::executeFlat: the same without Alignment containers -- the result "
                "of a device batch stays one header array + one ops array in page-locked memory\"}, "
                "\"multi_device_flat\": {%s, \"what\": \"MultiDeviceAligner::executeFlat: one process, device batches rotating over the "
                "replicas of the index, engines persistent (second run timed)\"}, "
                "\"sam_flat\": {\"reads_per_s\": %.1f, \"bytes\": %llu, \"what\": \"BatchFileWriter::execute on the flat batches: formatted by "
                "up to 32 threads into byte arenas, one write per batch\"}, "
                "\"batch_graph\": {%s, \"what\": \"BatchSource -> BatchAlign -> BatchFileWriter as graph nodes under promiseMe / "
                "simultaneousGet, 128 k reads per device batch, one device batch in flight per graph thread: reads in host memory -> SAM bytes\"}, "
                "\"fastq_to_sam_graph\": {%s, \"what\": \"FASTQ text in memory -> BatchFileReader (records cut out of the stream under its lock, "
                "reads built outside) -> BatchAlign -> BatchFileWriter -> SAM bytes: the shape of ExecutionContext::doAlign as batch graph nodes\"}, "
                "\"sam\": {\"reads_per_s\": %.1f, \"threads\": %u, \"bytes\": %llu, \"what\": \"FileWriter::execute per read into a "
                "counting sink\"}, "
                "\"graph\": {\"reads_per_s\": %.1f, \"threads\": %d, %s, \"what\": \"the per-read graph of export.cpp:99-126 (reader -> "
                "BinarySeeding -> StripOfConsideration -> Harmonization -> NeedlemanWunsch -> MappingQuality -> FileWriter), the reader "
                "node wrapped into PrefetchReader: it pulls 64 k reads ahead, sends them through all stages on the GPU and hands "
                "every graph thread reads that are aligned already; reads_per_s = the best of the thread counts\"}, "
                "\"graph_funnel\": {\"reads_per_s\": %.1f, \"graph_threads\": %d, \"device_batches\": %llu, \"mean_reads_per_device_batch\": %.1f, "
                "\"what\": \"the same graph with the plain reader: per-read execute() calls of that many graph threads funnelled into "
                "device batches (DeviceBatcher)\"}}\n",
                n, uiLen, uiHw, fLoad, sBatch.c_str( ), sFlat.c_str( ), sMulti.c_str( ), n / fSamFlat, (unsigned long long)uiSamFlatBytes, sBatchGraph.c_str( ),
                sFastqGraph.c_str( ), n / fSam, uiSamThreads, (unsigned long long)pSink->uiBytes.load( ), fPrefetchBest, iPrefetchBestThreads,
                sPrefetch.c_str( ), n / fGraph, iGraphThreads, (unsigned long long)xStat.first, xStat.first ? (double)xStat.second / xStat.first : 0.0 );
    }
    catch( const std::exception& e )
    {
        fprintf( stderr, "error: %s\n", e.what( ) );
        return 1;
    }
    return 0;
}
