// ma_engine.h -- the host-side machinery between MA's per-read Module graph and the batch-oriented C ABI:
//   Engine        one device batch (ma_batch) + one HIP stream on the device of its index: reads in host memory ->
//                 H2D -> all four stages -> D2H of every stage's records (BatchResult), with the time of each phase
//   PrefetchQueue the funnel turned round (round 4): a volatile SOURCE of the graph reads AHEAD -- it pulls a device batch
//                 worth of reads from the reader it wraps, sends them through all stages and hands every graph thread one
//                 read of a FINISHED batch with its ticket, so that no graph thread ever blocks per read and a few dozen
//                 graph threads keep the GPU busy (the per-read funnel needs a thousand)
//   DeviceBatcher thread-safe funnel: the graph threads of libMA::setUpCompGraph (export.cpp:84-126, one graph copy per
//                 thread, every copy calling BinarySeeding::execute concurrently and lock-free, module.h:303-369) hand in
//                 ONE read each and block; the reads that arrive while the GPU is busy form the next device batch, whose
//                 results every waiting thread then picks its slice from (SURVEY 8(b) "Threading")
// Nothing here computes: every stage runs on the GPU behind include/ma_amd.h; a non-zero status becomes
// std::runtime_error like in the module wrappers.
#pragma once
#include "../../include/ma_amd.h"
#include <algorithm>
#include <atomic>
#include <climits>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace ma_amd
{
namespace engine
{
// a read in host memory: codes A0 C1 G2 T3 N4; the owner keeps it alive while its batch is in flight
struct ReadRef
{
    const uint8_t* pCodes = nullptr;
    size_t uiLength = 0;
    ReadRef( )
    {}
    ReadRef( const uint8_t* pCodes, size_t uiLength ) : pCodes( pCodes ), uiLength( uiLength )
    {}
    ReadRef( const std::vector<uint8_t>& rCodes ) : pCodes( rCodes.data( ) ), uiLength( rCodes.size( ) )
    {}
};

inline void engineCheck( int rc )
{
    if( rc != 0 )
        throw std::runtime_error( ma_last_error( ) );
}
inline double secondsSince( const std::chrono::steady_clock::time_point& t0 )
{
    return std::chrono::duration<double>( std::chrono::steady_clock::now( ) - t0 ).count( );
}

// Copies of an index on the devices of vDevices (the index is replicated per GPU, 11.9 GB of 288 GB; the arrays are
// downloaded once and uploaded per device, each upload on a thread bound to its device).  The caller owns the handles
// (ma_index_destroy).  A device may be named more than once and may be the original's own ("virtual shards" in tests).
inline std::vector<ma_index*> replicateOnDevices( const ma_index* pOriginal, const std::vector<int>& vDevices )
{
    std::vector<ma_index*> vNew;
    if( vDevices.empty( ) )
        return vNew;
    uint64_t nWords = 0, nSa = 0, uiN = 0;
    int32_t nContigs = 0;
    engineCheck( ma_index_sizes( pOriginal, &nWords, &nSa, &uiN, &nContigs ) );
    std::vector<uint32_t> vBwt( nWords );
    std::vector<int64_t> vSa( nSa );
    std::vector<uint8_t> vPac( ( uiN / 2 + 3 ) / 4 + 1 );
    std::vector<uint64_t> vStarts( nContigs ), vLens( nContigs );
    uint64_t L2[ 5 ];
    int64_t primary = 0;
    engineCheck( ma_index_download( pOriginal, vBwt.data( ), vSa.data( ), L2, &primary, vPac.data( ), vStarts.data( ), vLens.data( ) ) );
    for( int iDev : vDevices )
    {
        ma_index* pCopy = nullptr;
        std::string sFailure;
        std::thread xCreator( [ & ]( ) { // the upload binds its own thread to the target device
            if( ma_set_device( iDev ) != 0 ||
                ma_index_create( vBwt.data( ), nWords, vSa.data( ), nSa, L2, primary, uiN, vPac.data( ), nContigs, vStarts.data( ),
                                 vLens.data( ), &pCopy ) != 0 )
                sFailure = ma_last_error( );
        } );
        xCreator.join( );
        if( !sFailure.empty( ) )
        {
            for( ma_index* p : vNew )
                ma_index_destroy( p );
            throw std::runtime_error( "replicateOnDevices: device " + std::to_string( iDev ) + ": " + sFailure );
        }
        vNew.push_back( pCopy );
    }
    return vNew;
}

// Page-locked host array that only grows (ma_host_alloc): the arrays a device batch is uploaded from and downloaded into.
// Not value-initialised: a download overwrites what it needs.
template <typename T> class HostBuf
{
    T* p = nullptr;
    size_t uiCap = 0;
    bool bPinned = false;

    void release( )
    {
        if( p != nullptr && bPinned )
            ma_host_free( p );
        else if( p != nullptr )
            free( p );
        p = nullptr;
        uiCap = 0;
    }

  public:
    HostBuf( )
    {}
    HostBuf( const HostBuf& ) = delete;
    HostBuf& operator=( const HostBuf& ) = delete;
    ~HostBuf( )
    {
        release( );
    }
    T* need( size_t n ) // at least n elements (contents are lost when it grows)
    {
        if( n > uiCap )
        {
            // the new block first, into locals: a failed (page-locked) allocation throws and must leave the buffer as it was,
            // not with p == nullptr and the new capacity (the engine is reused after a failed batch)
            const size_t uiNewCap = n + n / 4 + 64;
            // page-locking pays for the MB-sized arrays of a throughput batch; the few KB of a small batch of the per-read
            // funnel stay ordinary memory (page-locking and unlocking go through the driver and stall other streams)
            const bool bNewPinned = uiNewCap * sizeof( T ) >= ( 1u << 20 );
            void* q = nullptr;
            if( bNewPinned )
                engineCheck( ma_host_alloc( uiNewCap * sizeof( T ), &q ) );
            else if( ( q = malloc( uiNewCap * sizeof( T ) ) ) == nullptr )
                throw std::bad_alloc( );
            release( );
            p = static_cast<T*>( q );
            uiCap = uiNewCap;
            bPinned = bNewPinned;
        }
        return p;
    }
    T* data( )
    {
        return p;
    }
    const T* data( ) const
    {
        return p;
    }
    const T& operator[]( size_t i ) const
    {
        return p[ i ];
    }
};

// Host copies of the records of one device batch, CSR by read (offset arrays have n + 1 entries).
struct BatchResult
{
    size_t uiReads = 0;
    bool bStages = false; // segments / seeds / harmonized sets / unsorted-quality alignments present?
    std::vector<uint64_t> vSegOff, vSeedOff, vHsetOff, vHseedOff, vAlnOff;
    std::vector<ma_segment> vSegs;
    std::vector<ma_seed> vSeeds, vHseeds;
    std::vector<uint32_t> vHsetSoc;
    // the SoC queue of every read as the sweep leaves it (ma_batch_get_soc_heap): strips CSR by vSocOff, the read's seeds
    // re-sorted by reference position at vSeedOff (what a binding on the reference's own SoCPriorityQueue fills it with)
    std::vector<uint64_t> vSocOff;
    std::vector<ma_soc> vSocHeap;
    std::vector<ma_seed> vSortedSeeds;
    bool bSocQueues = false;
    std::vector<ma_alignment> vAlns; // NeedlemanWunsch output (bStages)
    std::vector<uint64_t> vAlnOps; // (type, length) pairs
    // MappingQuality output = the result of the path: a FLAT view, one header array + one ops array per device batch in
    // page-locked memory that the engine recycles; Alignment containers are built from it only where somebody asks for them
    HostBuf<uint64_t> vMqOff; // n + 1
    HostBuf<ma_alignment> vMq;
    HostBuf<uint64_t> vMqOps; // (type, length) pairs
    uint64_t uiMqAlignments = 0, uiMqOps = 0;
    uint64_t uiAlignedReads = 0;
    double fPack = 0, fH2D = 0, fKernels = 0, fD2H = 0; // seconds: gathering the reads, upload, all stages, download
    float aStageMs[ 8 ] = { 0, 0, 0, 0, 0, 0, 0, 0 }; // host wall time of seed / extract / chain / dp (ma_batch_host_ms)
};

class Engine
{
    const ma_index* pIndex;
    const ma_params xP;
    ma_batch* pBatch = nullptr;
    void* pStream = nullptr;
    uint64_t uiCapReads = 0, uiCapBases = 0;
    HostBuf<uint8_t> vCodes; // page-locked staging of the reads
    HostBuf<uint64_t> vOff;
    const bool bBlocking; // waits sleep instead of spinning (hosts that run far more threads than cores)
    // results are handed out as shared_ptr; one that nobody holds any more is reused (its page-locked arrays are kept)
    std::vector<std::shared_ptr<BatchResult>> vPool;

    std::shared_ptr<BatchResult> freshResult( )
    {
        for( auto& pR : vPool )
            if( pR.use_count( ) == 1 )
                return pR;
        if( vPool.size( ) < 64 )
        {
            vPool.push_back( std::make_shared<BatchResult>( ) );
            return vPool.back( );
        }
        return std::make_shared<BatchResult>( ); // the caller keeps many results alive: not pooled
    }

  public:
    bool bFetchSocQueues = false; // run(): with bStages also fetch every read's SoC queue (one extra kernel per batch)
    // batches this engine has completed.  Its FIRST batch allocates the device pools (GBs) and page-locks the staging
    // arrays, which stalls every stream of the process: callers run first batches one at a time, before they admit
    // concurrent work (primed( ), BatchAligner::alignRange, BatchAlign::execute, DeviceBatcher's constructor)
    uint64_t uiRuns = 0;
    bool primed( uint64_t uiReads ) const
    {
        return uiRuns != 0 && pBatch != nullptr && uiReads <= uiCapReads;
    }

  private:
    void fit( uint64_t uiReads, uint64_t uiBases )
    {
        if( pBatch != nullptr && uiReads <= uiCapReads && uiBases <= uiCapBases )
            return;
        if( pBatch != nullptr )
            ma_batch_destroy( pBatch );
        pBatch = nullptr;
        uiCapReads = std::max<uint64_t>( uiReads + uiReads / 4, 64 );
        uiCapBases = std::max<uint64_t>( uiBases + uiBases / 4, 4096 );
        engineCheck( ma_batch_create( pIndex, &xP, uiCapReads, uiCapBases + 64, &pBatch ) );
        engineCheck( ma_batch_set_stream( pBatch, pStream ) );
        engineCheck( ma_batch_set_blocking_sync( pBatch, bBlocking ? 1 : 0 ) );
    }

  public:
    Engine( const ma_index* pIndex, const ma_params& rP, bool bBlocking = false ) : pIndex( pIndex ), xP( rP ), bBlocking( bBlocking )
    {
        engineCheck( ma_stream_create( pIndex, &pStream ) );
        created( )++;
    }
    // engines constructed by this process so far (tests: a second run of a persistent aligner must not create any)
    static std::atomic<uint64_t>& created( )
    {
        static std::atomic<uint64_t> uiCreated{ 0 };
        return uiCreated;
    }
    const ma_index* index( ) const
    {
        return pIndex;
    }
    Engine( const Engine& ) = delete;
    Engine& operator=( const Engine& ) = delete;
    ~Engine( )
    {
        if( pBatch != nullptr )
            ma_batch_destroy( pBatch );
        ma_stream_destroy( pIndex, pStream );
    }

    // device batch and pools for batches of up to that size, now (instead of inside the first run)
    void reserve( uint64_t uiReads, uint64_t uiBases )
    {
        fit( uiReads, uiBases );
    }

    // vReads[i] = codes of read i (A0 C1 G2 T3 N4).  bStages: also fetch the records of the intermediate stages.
    std::shared_ptr<BatchResult> run( const std::vector<const std::vector<uint8_t>*>& vReads, bool bStages )
    {
        std::vector<ReadRef> vRefs;
        vRefs.reserve( vReads.size( ) );
        for( const std::vector<uint8_t>* pRead : vReads )
            vRefs.emplace_back( *pRead );
        return run( vRefs, bStages );
    }
    std::shared_ptr<BatchResult> run( const std::vector<ReadRef>& vReads, bool bStages )
    {
        const size_t n = vReads.size( );
        auto tPack = std::chrono::steady_clock::now( );
        uint64_t* pOff = vOff.need( n + 1 );
        pOff[ 0 ] = 0;
        for( size_t i = 0; i < n; i++ )
            pOff[ i + 1 ] = pOff[ i ] + vReads[ i ].uiLength;
        uint8_t* pCodes = vCodes.need( pOff[ n ] + 1 );
        // gather the reads into the staging array; large batches on a few threads (10^6 scattered 150-byte copies)
        auto gather = [ & ]( size_t lo, size_t hi ) {
            for( size_t i = lo; i < hi; i++ )
                if( vReads[ i ].uiLength != 0 )
                    memcpy( pCodes + pOff[ i ], vReads[ i ].pCodes, vReads[ i ].uiLength );
        };
        const size_t uiThreads = n >= ( 1u << 16 ) ? 4 : 1;
        if( uiThreads == 1 )
            gather( 0, n );
        else
        {
            std::vector<std::thread> vT;
            for( size_t t = 1; t < uiThreads; t++ )
                vT.emplace_back( gather, n * t / uiThreads, n * ( t + 1 ) / uiThreads );
            gather( 0, n / uiThreads );
            for( auto& rT : vT )
                rT.join( );
        }
        const double fPack = secondsSince( tPack );
        auto pRes = runFlat( pCodes, pOff, n, bStages );
        pRes->fPack = fPack;
        return pRes;
    }
    // reads that already are one array of codes + CSR offsets (n + 1), e.g. in page-locked memory the caller filled
    std::shared_ptr<BatchResult> runFlat( const uint8_t* pCodes, const uint64_t* pOff, size_t n, bool bStages )
    {
        auto pRes = freshResult( );
        BatchResult& R = *pRes;
        R.uiReads = n;
        R.bStages = bStages;
        R.bSocQueues = false;
        R.fPack = 0;
        fit( n, pOff[ n ] );
        auto t0 = std::chrono::steady_clock::now( );
        engineCheck( ma_batch_set_reads( pBatch, pCodes, pOff, n ) );
        R.fH2D = secondsSince( t0 );
        t0 = std::chrono::steady_clock::now( );
        engineCheck( ma_align_batch( pBatch ) );
        engineCheck( ma_batch_sync( pBatch ) );
        R.fKernels = secondsSince( t0 );
        engineCheck( ma_batch_host_ms( pBatch, R.aStageMs ) );
        t0 = std::chrono::steady_clock::now( );
        uint64_t nSeg = 0, nSeed = 0, nHset = 0, nHseed = 0, nAln = 0, nOps = 0;
        engineCheck( ma_batch_counts( pBatch, &nSeg, &nSeed, &nHset, &nHseed, &nAln, &nOps, &R.uiAlignedReads ) );
        if( bStages )
        {
            R.vSegOff.resize( n + 1 );
            R.vSegs.resize( nSeg + 1 );
            engineCheck( ma_batch_get_segments( pBatch, R.vSegOff.data( ), R.vSegs.data( ) ) );
            R.vSeedOff.resize( n + 1 );
            R.vSeeds.resize( nSeed + 1 );
            engineCheck( ma_batch_get_seeds( pBatch, R.vSeedOff.data( ), R.vSeeds.data( ) ) );
            R.vHsetOff.resize( n + 1 );
            R.vHseedOff.resize( nHset + 1 );
            R.vHsetSoc.resize( nHset + 1 );
            R.vHseeds.resize( nHseed + 1 );
            engineCheck( ma_batch_get_hsets( pBatch, R.vHsetOff.data( ), R.vHseedOff.data( ), R.vHsetSoc.data( ), R.vHseeds.data( ) ) );
            if( bFetchSocQueues )
            {
                uint64_t nSocs = 0;
                engineCheck( ma_batch_get_soc_heap( pBatch, &nSocs, nullptr, nullptr, nullptr, nullptr ) );
                R.vSocOff.resize( n + 1 );
                R.vSocHeap.resize( nSocs + 1 );
                R.vSortedSeeds.resize( nSeed + 1 );
                std::vector<uint64_t> vSeedOffAgain( n + 1 );
                engineCheck( ma_batch_get_soc_heap( pBatch, &nSocs, R.vSocOff.data( ), R.vSocHeap.data( ), vSeedOffAgain.data( ),
                                                    R.vSortedSeeds.data( ) ) );
                R.bSocQueues = true;
            }
            R.vAlnOff.resize( n + 1 );
            R.vAlns.resize( nAln + 1 );
            R.vAlnOps.resize( 2 * nOps + 2 );
            engineCheck( ma_batch_get_alignments( pBatch, R.vAlnOff.data( ), R.vAlns.data( ), R.vAlnOps.data( ) ) );
        }
        uint64_t* pMqOff = R.vMqOff.need( n + 1 );
        engineCheck( ma_batch_get_mapq_alignments( pBatch, pMqOff, R.vMq.need( nAln + 1 ), R.vMqOps.need( 2 * nOps + 2 ) ) );
        R.uiMqAlignments = n ? pMqOff[ n ] : 0;
        R.uiMqOps = 0;
        if( R.uiMqAlignments )
        {
            const ma_alignment& rLast = R.vMq[ R.uiMqAlignments - 1 ];
            R.uiMqOps = rLast.ops_off + rLast.n_ops;
        }
        R.fD2H = secondsSince( t0 );
        uiRuns++;
        if( getenv( "MA_ENGINE_TRACE" ) ) // diagnostics: the phases of every device batch
            fprintf( stderr, "engine %p: %zu reads, h2d %.4f s, stages %.4f s (seed %.1f extract %.1f chain %.1f dp %.1f ms), d2h %.4f s\n",
                     (void*)this, n, R.fH2D, R.fKernels, R.aStageMs[ 0 ], R.aStageMs[ 1 ], R.aStageMs[ 2 ], R.aStageMs[ 3 ], R.fD2H );
        return pRes;
    }
};

// One read's place in a finished device batch.
struct Ticket
{
    std::shared_ptr<const BatchResult> pResult;
    size_t uiRead = 0;
    explicit operator bool( ) const
    {
        return pResult != nullptr;
    }
};

struct BatcherOptions
{
    size_t uiMaxBatch = 1u << 18; // reads per device batch at most
    size_t uiEngines = 2; // device batches in flight (own stream each).  Measured with 2048 graph threads: 2 -> 109 k reads/s, 4 -> 63 k: with
                          // an engine always free a batch is sealed after xGather, while it is small, and small batches pay the fixed cost
    std::chrono::microseconds xGather{ 40 }; // an idle GPU still waits this long for more reads to arrive
    std::chrono::microseconds xMaxWait{ 20000 }; // a read never waits longer than this for its batch to be sealed
    std::chrono::microseconds xFill{ 3000 }; // while another batch runs, a batch smaller than half the last one waits this long for more
    // true: every stage's records are fetched, so SegmentVector / SoCPriorityQueue / seed sets / NeedlemanWunsch's
    // alignments carry their content (what a graph needs whose nodes are not all MI355X modules).  false: only the
    // MappingQuality-annotated alignments are fetched; the intermediate containers are empty shells that just pass the
    // ticket on -- enough for the chain of export.cpp:104-108 and several times cheaper per read.
    bool bStages = true;
    // with bStages: also fetch every read's SoC queue (needed by bindings whose SoCPriorityQueue is the reference's own
    // type and must therefore be filled eagerly; the mirror types of ma_modules.h compute it on demand)
    bool bSocQueues = false;
    // bases per read the engines' device pools are sized for when the batcher is constructed (a funnel batch holds at most
    // min( uiMaxBatch, 4096 ) reads); a long-read graph sets its expected read length
    size_t uiReserveBasesPerRead = 512;
};

// One-shot gate many threads sleep at until it opens.  A condition variable makes every woken thread re-acquire the mutex it
// waited with: the ~600 graph threads of one device batch then leave one by one (and fight the threads that are entering the
// next batch for the same mutex) -- measured as ~45 ms per cycle of a graph thread, against 5 ms on the GPU.  A futex wakes
// them all at once and none of them needs a lock afterwards.
class Gate
{
    std::atomic<int> iOpen{ 0 };

  public:
    void wait( )
    {
        while( iOpen.load( std::memory_order_acquire ) == 0 )
            syscall( SYS_futex, reinterpret_cast<int*>( &iOpen ), FUTEX_WAIT_PRIVATE, 0, nullptr, nullptr, 0 );
    }
    void open( )
    {
        iOpen.store( 1, std::memory_order_release );
        syscall( SYS_futex, reinterpret_cast<int*>( &iOpen ), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0 );
    }
};

class DeviceBatcher
{
    struct Slot
    {
        std::vector<ReadRef> vReads;
        std::shared_ptr<const BatchResult> pResult;
        std::string sError;
        bool bSealed = false;
        Gate xDone; // only this batch's readers wait here: a finished batch wakes nobody else
    };
    const ma_index* pIndex;
    const ma_params xP;
    const BatcherOptions xOpt;
    std::mutex xMutex;
    std::condition_variable xChanged;
    std::shared_ptr<Slot> pOpen;
    std::vector<std::unique_ptr<Engine>> vIdle; // engines not running a batch
    size_t uiEnginesMade = 0, uiRunning = 0, uiLastSealed = 0;
    uint64_t uiBatches = 0, uiReadsTotal = 0;
    double fSumH2D = 0, fSumKernels = 0, fSumD2H = 0, fSumRun = 0; // seconds over all batches
    double aSumStageMs[ 4 ] = { 0, 0, 0, 0 };

    // called with the lock held by the thread that sealed the slot; releases the lock while the GPU works
    void runSealed( std::unique_lock<std::mutex>& rLock, const std::shared_ptr<Slot>& pSlot )
    {
        xChanged.wait( rLock, [ & ]( ) { return !vIdle.empty( ) || uiEnginesMade < xOpt.uiEngines; } );
        std::unique_ptr<Engine> pEngine;
        if( !vIdle.empty( ) )
        {
            pEngine = std::move( vIdle.back( ) );
            vIdle.pop_back( );
        }
        else
            uiEnginesMade++; // constructed below, outside the lock
        uiRunning++;
        rLock.unlock( );
        const auto tRun = std::chrono::steady_clock::now( );
        try
        {
            if( pEngine == nullptr )
            {
                pEngine.reset( new Engine( pIndex, xP, true ) );
                pEngine->bFetchSocQueues = xOpt.bSocQueues;
            }
            pSlot->pResult = pEngine->run( pSlot->vReads, xOpt.bStages );
        }
        catch( const std::exception& rE )
        {
            pSlot->sError = rE.what( );
            if( pSlot->sError.empty( ) )
                pSlot->sError = "device batch failed";
        }
        rLock.lock( );
        if( pEngine != nullptr )
            vIdle.push_back( std::move( pEngine ) );
        else
            uiEnginesMade--;
        uiRunning--;
        uiBatches++;
        uiReadsTotal += pSlot->vReads.size( );
        fSumRun += secondsSince( tRun );
        if( pSlot->pResult != nullptr )
        {
            fSumH2D += pSlot->pResult->fH2D, fSumKernels += pSlot->pResult->fKernels, fSumD2H += pSlot->pResult->fD2H;
            for( int k = 0; k < 4; k++ )
                aSumStageMs[ k ] += pSlot->pResult->aStageMs[ k ];
        }
        xChanged.notify_all( ); // leaders of open batches and sealers waiting for a device slot
        pSlot->xDone.open( ); // result and error of the slot are published by the release store inside
    }

  public:
    // Engines before admission: all uiEngines engines (stream + device batch sized for the batches a funnel sees) exist when
    // the constructor returns; none is created while reads are being aligned.  Constructing a batcher therefore allocates
    // device memory and streams even if no read ever arrives (the per-read modules construct theirs on the first execute( ) of a
    // graph thread, BinarySeeding::batcherFor); a device that is short of memory fails HERE, with a message that says so.
    DeviceBatcher( const ma_index* pIndex, const ma_params& rP, const BatcherOptions& rOpt = BatcherOptions( ) )
        : pIndex( pIndex ), xP( rP ), xOpt( rOpt )
    {
        const uint64_t uiReads = std::min<uint64_t>( std::max<uint64_t>( xOpt.uiMaxBatch, 1 ), 4096 );
        for( size_t k = 0; k < xOpt.uiEngines; k++ )
        {
            try
            {
                std::unique_ptr<Engine> pEngine( new Engine( pIndex, xP, true ) );
                pEngine->bFetchSocQueues = xOpt.bSocQueues;
                pEngine->reserve( uiReads, uiReads * std::max<size_t>( xOpt.uiReserveBasesPerRead, 1 ) );
                vIdle.push_back( std::move( pEngine ) );
                uiEnginesMade++;
            }
            catch( const std::exception& rE )
            {
                throw std::runtime_error( "DeviceBatcher: engine " + std::to_string( k ) + " of " + std::to_string( xOpt.uiEngines ) + " (" +
                                          std::to_string( uiReads ) + " reads x " + std::to_string( xOpt.uiReserveBasesPerRead ) +
                                          " bases): " + rE.what( ) );
            }
        }
    }
    DeviceBatcher( const DeviceBatcher& ) = delete;

    // Blocks until the batch that contains this read has been through all stages.  rCodes must stay alive meanwhile
    // (the caller holds the NucSeq).  Re-entrant: called concurrently by all graph threads.
    Ticket align( const std::vector<uint8_t>& rCodes )
    {
        return align( ReadRef( rCodes ) );
    }
    Ticket align( const ReadRef& rRead )
    {
        std::unique_lock<std::mutex> xLock( xMutex );
        if( pOpen == nullptr )
            pOpen = std::make_shared<Slot>( );
        std::shared_ptr<Slot> pSlot = pOpen;
        const size_t uiMine = pSlot->vReads.size( );
        pSlot->vReads.push_back( rRead );
        auto seal = [ & ]( ) {
            pSlot->bSealed = true;
            uiLastSealed = pSlot->vReads.size( );
            if( pOpen == pSlot )
                pOpen = nullptr;
        };
        if( pSlot->vReads.size( ) >= xOpt.uiMaxBatch )
        {
            seal( );
            runSealed( xLock, pSlot );
        }
        else if( uiMine == 0 )
        {
            // the first read of a batch leads it: it seals the batch once a device slot is free and the arrivals have had
            // a moment to gather (while all device slots are busy the batch simply keeps growing), or after xMaxWait
            // While ANOTHER batch is on the device a free engine is no reason to leave with a handful of reads: the readers of
            // the batch that just finished are only now coming back, and a batch of 30 reads costs the same ~4 ms as one of 600
            // (measured: the batches alternated between ~1100 and ~40 reads, and less than one of them was in flight on average).
            // Such a batch waits until it holds half of what the last sealed batch held, or xFill.
            const auto tOpened = std::chrono::steady_clock::now( );
            while( !pSlot->bSealed )
            {
                const auto tNow = std::chrono::steady_clock::now( );
                const bool bGathered = tNow - tOpened >= xOpt.xGather;
                const bool bFilled = uiRunning == 0 || pSlot->vReads.size( ) >= uiLastSealed / 2 || tNow - tOpened >= xOpt.xFill;
                if( ( bGathered && bFilled && uiRunning < xOpt.uiEngines ) || tNow - tOpened >= xOpt.xMaxWait )
                    break;
                xChanged.wait_for( xLock, bGathered ? std::chrono::microseconds( 200 ) : xOpt.xGather );
            }
            if( !pSlot->bSealed )
            {
                seal( );
                runSealed( xLock, pSlot );
            }
        }
        xLock.unlock( );
        pSlot->xDone.wait( );
        if( !pSlot->sError.empty( ) )
            throw std::runtime_error( pSlot->sError );
        Ticket xT;
        xT.pResult = pSlot->pResult;
        xT.uiRead = uiMine;
        return xT;
    }

    // statistics: device batches run so far, reads they carried
    void stats( uint64_t& rBatches, uint64_t& rReads )
    {
        std::lock_guard<std::mutex> xLock( xMutex );
        rBatches = uiBatches;
        rReads = uiReadsTotal;
    }
    // seconds summed over all device batches: whole run() calls, and inside them upload / kernels / download
    void phaseSeconds( double& rRun, double& rH2D, double& rKernels, double& rD2H )
    {
        std::lock_guard<std::mutex> xLock( xMutex );
        rRun = fSumRun, rH2D = fSumH2D, rKernels = fSumKernels, rD2H = fSumD2H;
    }
    void stageSeconds( double aOut[ 4 ] ) // host wall time of the four stage calls, summed over all batches
    {
        std::lock_guard<std::mutex> xLock( xMutex );
        for( int k = 0; k < 4; k++ )
            aOut[ k ] = aSumStageMs[ k ] / 1e3;
    }
};

struct PrefetchOptions
{
    size_t uiBatchReads = 1u << 16; // reads per device batch pulled ahead
    size_t uiDepth = 2; // device batches ahead of the graph threads (in flight or finished and not yet handed out): one engine each
    // reads of a finished batch a graph thread takes at a time: the queue's lock is taken once per slice, not once per read
    // (with more graph threads than the host grants cores, a thread that is descheduled while it holds a lock stalls them all)
    size_t uiSlice = 128;
    bool bStages = false, bSocQueues = false; // as in BatcherOptions
    // bases per read the engines' device pools and staging arrays are sized for up front (short reads: 256; a long-read graph
    // sets its expected read length, otherwise the first batch re-allocates inside the run)
    size_t uiReserveBasesPerRead = 256;
};

// Reads pulled AHEAD of the graph threads.  TP_ITEM is the read handle of the host layer (shared_ptr to its NucSeq type).
//   next( out, pull, ref ): the next read of a finished device batch with its ticket; false = the source is exhausted and
//   everything pulled has been handed out.  `pull` returns the next read of the wrapped source or an empty handle at its
//   end; it is called by ONE thread at a time (the wrapped reader need not be re-entrant).  A caller that finds fewer than
//   uiDepth batches ahead pulls and runs the next batch itself (it blocks for one device batch, the others keep being
//   served from the finished ones), so there is no thread of our own and nothing runs when nobody asks.
// The reads a batch pulled stay with the batch until its last slice is done and are then released together by ONE thread: the
// host layer hands the graph threads COPIES (made, and later freed, by the consuming thread).  Handing out the pulled objects
// themselves was measured at a third of the rate: they are allocated by the thread that pulled the batch and would be freed
// by 8-16 others, all contending for that one thread's malloc arena (13.4 vs 5.1 us of thread time per read).
// Order: reads are handed out batch by batch in pull order, within a batch in pull order; which graph thread gets which
// read is as arbitrary as in the reference (every graph thread takes the next read from the shared reader, export.cpp:99-126).
// TP_ENGINE: Engine; the CPU test of the queue's bookkeeping (tests/emul/prefetch_queue_test.cpp) puts a stand-in there.
template <typename TP_ITEM, typename TP_ENGINE = Engine> class PrefetchQueue
{
    struct Batch
    {
        std::vector<TP_ITEM> vItems;
        std::shared_ptr<const BatchResult> pResult;
        size_t uiNext = 0;
    };
    const ma_params xP;
    const PrefetchOptions xOpt;
    std::mutex xMutex; // state below
    std::condition_variable xChanged;
    std::vector<std::shared_ptr<Batch>> vReady; // finished batches, oldest first
    // engines not running a batch.  With several index replicas (one per GPU of the node) every replica has uiDepth
    // engines of its own and the device batches rotate over them: whoever pulls the next batch takes the engine that has been
    // idle longest, so G GPUs are fed by the unchanged graph of export.cpp:99-126 -- tickets carry their batch's result, the
    // modules downstream never ask which device computed it
    std::vector<std::unique_ptr<TP_ENGINE>> vIdle;
    size_t uiDepthTotal = 1; // uiDepth per replica
    size_t uiLoading = 0; // batches being pulled or on the device
    bool bEof = false;
    std::string sError;
    std::mutex xPullMutex; // the wrapped source is read by one thread at a time, batches are pulled in order
    bool bSourceEnded = false; // (under xPullMutex) the source has returned its end marker: it is not asked again
    // what a thread's unfinished slice checks before it touches its queue's batch: a queue that was destroyed (an aborted
    // graph) leaves stale slices behind in the threads that served it; they are dropped the next time the thread looks
    const std::shared_ptr<int> pAlive = std::make_shared<int>( 0 );
    static uint64_t nextId( )
    {
        static std::atomic<uint64_t> uiNext{ 1 };
        return uiNext++;
    }
    const uint64_t uiId = nextId( ); // what a thread's slice remembers of its queue
    uint64_t uiBatches = 0, uiReadsTotal = 0;
    double fSumRun = 0, fSumPull = 0;

  public:
    // Constructing a queue creates uiDepth engines per index replica -- a stream, device pools for uiBatchReads reads of
    // uiReserveBasesPerRead bases (GBs for long reads) and page-locked staging each: before the first read is asked for, not
    // inside the first timed batch.  A device that is short of memory makes the constructor throw (with the replica's number).
    PrefetchQueue( const std::vector<const ma_index*>& vIndices, const ma_params& rP, const PrefetchOptions& rOpt = PrefetchOptions( ) )
        : xP( rP ), xOpt( rOpt )
    {
        if( vIndices.empty( ) )
            throw std::runtime_error( "PrefetchQueue: no index" );
        uiDepthTotal = std::max<size_t>( xOpt.uiDepth, 1 ) * vIndices.size( );
        // round k holds the k-th engine of every replica: vIdle is served from its front, so consecutive batches go to
        // different devices
        for( size_t k = 0; k < std::max<size_t>( xOpt.uiDepth, 1 ); k++ )
            for( size_t g = 0; g < vIndices.size( ); g++ )
            {
                try
                {
                    std::unique_ptr<TP_ENGINE> pEngine( new TP_ENGINE( vIndices[ g ], xP, true ) );
                    pEngine->bFetchSocQueues = xOpt.bSocQueues;
                    pEngine->reserve( xOpt.uiBatchReads, xOpt.uiBatchReads * std::max<size_t>( xOpt.uiReserveBasesPerRead, 1 ) );
                    vIdle.push_back( std::move( pEngine ) );
                }
                catch( const std::exception& rE )
                {
                    throw std::runtime_error( "PrefetchQueue: engine " + std::to_string( k ) + " of index replica " + std::to_string( g ) +
                                              " (" + std::to_string( xOpt.uiBatchReads ) + " reads x " +
                                              std::to_string( xOpt.uiReserveBasesPerRead ) + " bases): " + rE.what( ) );
                }
            }
    }
    PrefetchQueue( const ma_index* pIndex, const ma_params& rP, const PrefetchOptions& rOpt = PrefetchOptions( ) )
        : PrefetchQueue( std::vector<const ma_index*>( 1, pIndex ), rP, rOpt )
    {}
    PrefetchQueue( const PrefetchQueue& ) = delete;

    // the calling thread's slice of a finished batch (thread-local: no lock while it lasts)
    struct Slice
    {
        uint64_t uiOwner = 0; // id of the queue the slice belongs to (not its address: a later queue may be allocated there)
        std::weak_ptr<int> pOwnerAlive;
        std::shared_ptr<Batch> pBatch;
        // the batch's result behind a control block of THIS thread's own: the tickets of a slice are copied a dozen times per
        // read (query -> segments -> ... -> alignments), and reference counts that 16 threads keep bumping on ONE control
        // block bounce its cache line between the cores (measured: half of a graph thread's time per read)
        std::shared_ptr<const BatchResult> pResult;
        size_t uiNext = 0, uiEnd = 0;
        void take( const std::shared_ptr<Batch>& pB, size_t uiFrom, size_t uiTo, uint64_t uiQueue, const std::shared_ptr<int>& pAliveOfQueue )
        {
            uiOwner = uiQueue, pOwnerAlive = pAliveOfQueue, pBatch = pB, uiNext = uiFrom, uiEnd = uiTo;
            std::shared_ptr<const BatchResult> pKeep = pB->pResult;
            pResult = std::shared_ptr<const BatchResult>( pKeep.get( ), [ pKeep ]( const BatchResult* ) {} );
        }
        void drop( )
        {
            pBatch.reset( );
            pResult.reset( );
            uiNext = uiEnd = 0;
        }
        bool empty( ) const
        {
            return uiNext >= uiEnd;
        }
    };
    // One slice PER QUEUE the thread serves: a graph thread that alternates between two PrefetchReaders -- two
    // wrapped files, two aligners in one process -- keeps its unfinished slice of queue A while it takes reads of queue B; with
    // one slice per thread the reads of A's slice, already taken out of A's batch, were lost.  Finished slices and slices of
    // queues that no longer exist are removed on the way.
    static Slice& mySlice( uint64_t uiQueue )
    {
        static thread_local std::vector<Slice> vSlices;
        Slice* pFound = nullptr;
        for( size_t k = 0; k < vSlices.size( ); )
        {
            if( vSlices[ k ].uiOwner == uiQueue )
                pFound = &vSlices[ k ];
            else if( vSlices[ k ].empty( ) || vSlices[ k ].pOwnerAlive.expired( ) )
            {
                if( pFound == &vSlices.back( ) )
                    pFound = &vSlices[ k ];
                vSlices[ k ] = std::move( vSlices.back( ) );
                vSlices.pop_back( );
                continue;
            }
            k++;
        }
        if( pFound != nullptr )
            return *pFound;
        vSlices.emplace_back( );
        vSlices.back( ).uiOwner = uiQueue;
        return vSlices.back( );
    }

    template <typename TP_PULL, typename TP_REF> bool next( TP_ITEM& rItem, Ticket& rTicket, TP_PULL&& fPull, TP_REF&& fRef )
    {
        Slice& rMine = mySlice( uiId );
        if( rMine.uiNext < rMine.uiEnd )
        {
            const size_t k = rMine.uiNext++;
            rItem = rMine.pBatch->vItems[ k ];
            rTicket.pResult = rMine.pResult;
            rTicket.uiRead = k;
            if( rMine.uiNext >= rMine.uiEnd )
                rMine.drop( );
            return true;
        }
        std::unique_lock<std::mutex> xLock( xMutex );
        for( ;; )
        {
            if( !sError.empty( ) )
                throw std::runtime_error( sError );
            // keep uiDepth batches ahead: this caller pulls and runs the next one, the others are served meanwhile
            if( !bEof && uiLoading + vReady.size( ) < uiDepthTotal && !vIdle.empty( ) )
            {
                uiLoading++;
                std::unique_ptr<TP_ENGINE> pEngine = std::move( vIdle.front( ) ); // idle longest: batches rotate over the replicas
                vIdle.erase( vIdle.begin( ) );
                xLock.unlock( );
                auto pBatch = std::make_shared<Batch>( );
                std::string sFailed;
                bool bEnd = false;
                const auto tPull = std::chrono::steady_clock::now( );
                double fPullS = 0, fRun = 0;
                try
                {
                    {
                        std::lock_guard<std::mutex> xPull( xPullMutex );
                        // a source that has ended is not asked again (it need not be idempotent at its end): a second puller
                        // may have been waiting for this lock while the first one met the end
                        bEnd = bSourceEnded;
                        if( !bEnd )
                            pBatch->vItems.reserve( xOpt.uiBatchReads );
                        while( !bEnd && pBatch->vItems.size( ) < xOpt.uiBatchReads )
                        {
                            TP_ITEM xItem = fPull( );
                            if( !xItem )
                                bEnd = bSourceEnded = true;
                            else
                                pBatch->vItems.push_back( std::move( xItem ) );
                        }
                    }
                    fPullS = secondsSince( tPull );
                    if( !pBatch->vItems.empty( ) )
                    {
                        const auto tRun = std::chrono::steady_clock::now( );
                        std::vector<ReadRef> vRefs;
                        vRefs.reserve( pBatch->vItems.size( ) );
                        for( const TP_ITEM& rI : pBatch->vItems )
                            vRefs.push_back( fRef( rI ) );
                        pBatch->pResult = pEngine->run( vRefs, xOpt.bStages );
                        fRun = secondsSince( tRun );
                    }
                }
                catch( const std::exception& rE )
                {
                    sFailed = rE.what( );
                    if( sFailed.empty( ) )
                        sFailed = "device batch failed";
                }
                catch( ... ) // whatever the wrapped source throws: the queue's state is restored and every waiter sees the failure
                {
                    sFailed = "PrefetchQueue: the wrapped source or the device batch threw an exception that is not a std::exception";
                }
                xLock.lock( );
                vIdle.push_back( std::move( pEngine ) );
                uiLoading--;
                if( bEnd )
                    bEof = true;
                if( !sFailed.empty( ) )
                {
                    bEof = true;
                    if( sError.empty( ) )
                        sError = sFailed;
                }
                else if( !pBatch->vItems.empty( ) )
                {
                    // (with several batches ahead they may finish, and are handed out, in another order than they were pulled)
                    vReady.push_back( pBatch );
                    uiBatches++, uiReadsTotal += pBatch->vItems.size( );
                    fSumRun += fRun, fSumPull += fPullS;
                }
                xChanged.notify_all( );
                continue;
            }
            if( !vReady.empty( ) )
            {
                std::shared_ptr<Batch> pB = vReady.front( );
                const size_t k = pB->uiNext;
                pB->uiNext = std::min( pB->vItems.size( ), k + std::max<size_t>( xOpt.uiSlice, 1 ) );
                const size_t uiEnd = pB->uiNext;
                if( pB->uiNext >= pB->vItems.size( ) )
                {
                    vReady.erase( vReady.begin( ) );
                    xChanged.notify_all( ); // room for another batch ahead
                }
                xLock.unlock( );
                rMine.take( pB, k + 1, uiEnd, uiId, pAlive );
                rItem = pB->vItems[ k ];
                rTicket.pResult = rMine.pResult;
                rTicket.uiRead = k;
                if( rMine.uiNext >= rMine.uiEnd )
                    rMine.drop( );
                return true;
            }
            if( bEof && uiLoading == 0 )
                return false;
            xChanged.wait( xLock );
        }
    }
    void stats( uint64_t& rBatches, uint64_t& rReads, double& rRunSeconds, double& rPullSeconds )
    {
        std::lock_guard<std::mutex> xLock( xMutex );
        rBatches = uiBatches, rReads = uiReadsTotal, rRunSeconds = fSumRun, rPullSeconds = fSumPull;
    }
};
} // namespace engine
} // namespace ma_amd
