// CPU test of PrefetchQueue's bookkeeping (ma_amd/host/ma_engine.h) with a stand-in engine: no GPU, no libma_amd.so.
//   two     two queues over two sources, served by the SAME threads which alternate between them after every read (a graph
//           thread that serves two PrefetchReaders): every read of both sources must be handed out exactly once, with the
//           ticket of its own batch (one thread-local slice per thread lost the unfinished slice of the
//           other queue)
//   abort   a queue is destroyed while threads still hold unfinished slices of it; the same threads then serve a new
//           queue: no stale slice may be handed out, and the old batch's memory is released
//   throw   the wrapped source throws something that is not a std::exception: every caller sees a failure, nobody hangs
//   eof     a source that must not be asked again after its end marker (it aborts if it is)
//   replicas  three "index replicas": batches rotate over their engines
#include "../../ma_amd/host/ma_engine.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>

// the C ABI symbols ma_engine.h refers to are never called here (the stand-in engine replaces Engine); they only have to link
extern "C" {
const char* ma_last_error( void ) { return "stub"; }
int ma_host_alloc( uint64_t, void** ) { return 1; }
int ma_host_free( void* ) { return 0; }
int ma_batch_create( const ma_index*, const ma_params*, uint64_t, uint64_t, ma_batch** ) { return 1; }
int ma_batch_destroy( ma_batch* ) { return 0; }
int ma_batch_set_stream( ma_batch*, void* ) { return 1; }
int ma_batch_set_blocking_sync( ma_batch*, int ) { return 1; }
int ma_batch_set_reads( ma_batch*, const uint8_t*, const uint64_t*, uint64_t ) { return 1; }
int ma_align_batch( ma_batch* ) { return 1; }
int ma_batch_sync( ma_batch* ) { return 1; }
int ma_batch_host_ms( ma_batch*, float* ) { return 1; }
int ma_batch_counts( ma_batch*, uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint64_t* ) { return 1; }
int ma_batch_get_segments( ma_batch*, uint64_t*, ma_segment* ) { return 1; }
int ma_batch_get_seeds( ma_batch*, uint64_t*, ma_seed* ) { return 1; }
int ma_batch_get_hsets( ma_batch*, uint64_t*, uint64_t*, uint32_t*, ma_seed* ) { return 1; }
int ma_batch_get_soc_heap( ma_batch*, uint64_t*, uint64_t*, ma_soc*, uint64_t*, ma_seed* ) { return 1; }
int ma_batch_get_alignments( ma_batch*, uint64_t*, ma_alignment*, uint64_t* ) { return 1; }
int ma_batch_get_mapq_alignments( ma_batch*, uint64_t*, ma_alignment*, uint64_t* ) { return 1; }
int ma_stream_create( const ma_index*, void** ) { return 1; }
int ma_stream_destroy( const ma_index*, void* ) { return 0; }
}

using namespace ma_amd::engine;

static std::atomic<long> g_liveResults{ 0 };
struct CountedResult : BatchResult
{
    CountedResult( ) { g_liveResults++; }
    ~CountedResult( ) { g_liveResults--; }
};
// stand-in for Engine: the "device batch" records the first code of every read (the test puts the read's number there)
struct FakeEngine
{
    const ma_index* pIndex;
    bool bFetchSocQueues = false;
    static std::atomic<int>& runsOf( size_t g )
    {
        static std::atomic<int> a[ 8 ];
        return a[ g ];
    }
    FakeEngine( const ma_index* pIndex, const ma_params&, bool ) : pIndex( pIndex ) {}
    void reserve( uint64_t, uint64_t ) {}
    std::shared_ptr<BatchResult> run( const std::vector<ReadRef>& vReads, bool )
    {
        auto p = std::make_shared<CountedResult>( );
        p->uiReads = vReads.size( );
        p->vSegOff.resize( vReads.size( ) );
        for( size_t i = 0; i < vReads.size( ); i++ )
            memcpy( &p->vSegOff[ i ], vReads[ i ].pCodes, 8 ); // the read's identity, to be found again through the ticket
        runsOf( (size_t)( (uintptr_t)pIndex - 1 ) )++;
        std::this_thread::sleep_for( std::chrono::microseconds( 300 ) );
        return p;
    }
};
typedef std::shared_ptr<std::vector<uint8_t>> Item;
typedef PrefetchQueue<Item, FakeEngine> Queue;

struct Source
{
    uint64_t uiTag, n, i = 0;
    bool bEnded = false, bStrict = false, bThrow = false;
    Item pull( )
    {
        if( bEnded && bStrict )
        {
            fprintf( stderr, "the source was asked again after its end marker\n" );
            abort( );
        }
        if( bThrow && i == n / 2 )
            throw 42;
        if( i >= n )
        {
            bEnded = true;
            return nullptr;
        }
        auto p = std::make_shared<std::vector<uint8_t>>( 8 );
        const uint64_t v = uiTag << 32 | i++;
        memcpy( p->data( ), &v, 8 );
        return p;
    }
};
static const ma_index* fakeIndex( size_t g )
{
    return (const ma_index*)( (uintptr_t)g + 1 );
}

int main( int argc, char** argv )
{
    const std::string sMode = argc > 1 ? argv[ 1 ] : "two";
    ma_params P;
    memset( &P, 0, sizeof( P ) );
    PrefetchOptions O;
    O.uiBatchReads = 97;
    O.uiDepth = 2;
    O.uiSlice = 16;
    const int nThreads = 6;
    if( sMode == "two" || sMode == "replicas" || sMode == "eof" )
    {
        const size_t G = sMode == "replicas" ? 3 : 1;
        std::vector<const ma_index*> vIdx;
        for( size_t g = 0; g < G; g++ )
            vIdx.push_back( fakeIndex( g ) );
        Queue xA( vIdx, P, O ), xB( vIdx, P, O );
        Source sA{ 1, 5000 }, sB{ 2, 3333 };
        sA.bStrict = sB.bStrict = true;
        std::mutex xSeenMutex;
        std::map<uint64_t, int> xSeen;
        std::atomic<int> iBad{ 0 };
        auto worker = [ & ]( int t ) {
            bool bA = true, bB = true;
            int k = t;
            while( bA || bB )
            {
                const bool bUseA = bA && ( !bB || ( k++ & 1 ) == 0 ); // alternate after every read
                Queue& rQ = bUseA ? xA : xB;
                Source& rS = bUseA ? sA : sB;
                Item pItem;
                Ticket xT;
                const bool bGot = rQ.next( pItem, xT, [ & ]( ) { return rS.pull( ); }, []( const Item& p ) { return ReadRef( *p ); } );
                if( !bGot )
                {
                    ( bUseA ? bA : bB ) = false;
                    continue;
                }
                uint64_t v;
                memcpy( &v, pItem->data( ), 8 );
                if( xT.pResult == nullptr || xT.uiRead >= xT.pResult->uiReads || xT.pResult->vSegOff[ xT.uiRead ] != v )
                    iBad++; // the ticket does not lead to this read's record
                std::lock_guard<std::mutex> xG( xSeenMutex );
                xSeen[ v ]++;
            }
        };
        std::vector<std::thread> vT;
        for( int t = 0; t < nThreads; t++ )
            vT.emplace_back( worker, t );
        for( auto& r : vT )
            r.join( );
        size_t uiOnce = 0;
        for( auto& kv : xSeen )
            uiOnce += kv.second == 1;
        printf( "{\"mode\": \"%s\", \"reads\": %zu, \"seen_once\": %zu, \"bad_tickets\": %d, \"runs_per_replica\": [%d, %d, %d]}\n", sMode.c_str( ),
                (size_t)( sA.n + sB.n ), uiOnce, iBad.load( ), FakeEngine::runsOf( 0 ).load( ), FakeEngine::runsOf( 1 ).load( ),
                FakeEngine::runsOf( 2 ).load( ) );
        bool bOk = uiOnce == sA.n + sB.n && xSeen.size( ) == sA.n + sB.n && iBad == 0;
        if( sMode == "replicas" ) // batches rotate: every replica ran about a third of them
        {
            const int tot = FakeEngine::runsOf( 0 ) + FakeEngine::runsOf( 1 ) + FakeEngine::runsOf( 2 );
            for( size_t g = 0; g < 3; g++ )
                bOk = bOk && FakeEngine::runsOf( g ) * 5 >= tot;
        }
        return bOk ? 0 : 1;
    }
    if( sMode == "abort" )
    {
        std::vector<std::thread> vT;
        std::atomic<int> iStale{ 0 }, iPhase{ 0 }, iArrived{ 0 };
        auto pOld = std::make_shared<Queue>( fakeIndex( 0 ), P, O );
        std::shared_ptr<Queue> pNew;
        Source sOld{ 7, 4000 }, sNew{ 8, 1500 };
        std::mutex xSeenMutex;
        std::set<uint64_t> xSeenNew;
        auto worker = [ & ]( ) {
            // phase 0: every thread takes ONE read of the old queue: it now holds an unfinished slice (15 more reads)
            Item pItem;
            Ticket xT;
            pOld->next( pItem, xT, [ & ]( ) { return sOld.pull( ); }, []( const Item& p ) { return ReadRef( *p ); } );
            xT = Ticket( );
            iArrived++;
            while( iPhase.load( ) == 0 )
                std::this_thread::yield( );
            // phase 1: the old queue is gone; serve the new one to its end
            while( pNew->next( pItem, xT, [ & ]( ) { return sNew.pull( ); }, []( const Item& p ) { return ReadRef( *p ); } ) )
            {
                uint64_t v;
                memcpy( &v, pItem->data( ), 8 );
                if( ( v >> 32 ) != 8 )
                    iStale++;
                std::lock_guard<std::mutex> xG( xSeenMutex );
                xSeenNew.insert( v );
            }
        };
        for( int t = 0; t < nThreads; t++ )
            vT.emplace_back( worker );
        while( iArrived.load( ) < nThreads )
            std::this_thread::yield( );
        pOld.reset( ); // the graph was aborted
        pNew = std::make_shared<Queue>( fakeIndex( 0 ), P, O );
        iPhase = 1;
        for( auto& r : vT )
            r.join( );
        pNew.reset( );
        // the threads have exited: every slice (thread-local) is gone, so is every batch result
        printf( "{\"mode\": \"abort\", \"new_reads_seen\": %zu, \"stale\": %d, \"live_results\": %ld}\n", xSeenNew.size( ), iStale.load( ),
                g_liveResults.load( ) );
        return xSeenNew.size( ) == sNew.n && iStale == 0 && g_liveResults == 0 ? 0 : 1;
    }
    if( sMode == "throw" )
    {
        Queue xQ( fakeIndex( 0 ), P, O );
        Source s{ 3, 1000 };
        s.bThrow = true;
        std::atomic<int> iFailed{ 0 }, iServed{ 0 };
        auto worker = [ & ]( ) {
            try
            {
                Item pItem;
                Ticket xT;
                while( xQ.next( pItem, xT, [ & ]( ) { return s.pull( ); }, []( const Item& p ) { return ReadRef( *p ); } ) )
                    iServed++;
            }
            catch( const std::runtime_error& )
            {
                iFailed++;
            }
        };
        std::vector<std::thread> vT;
        for( int t = 0; t < nThreads; t++ )
            vT.emplace_back( worker );
        for( auto& r : vT )
            r.join( );
        printf( "{\"mode\": \"throw\", \"threads_failed\": %d, \"served_before\": %d}\n", iFailed.load( ), iServed.load( ) );
        return iFailed >= 1 ? 0 : 1; // nobody hangs (we got here); whoever asked after the failure saw it
    }
    fprintf( stderr, "unknown mode\n" );
    return 2;
}
