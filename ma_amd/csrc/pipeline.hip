// pipeline.hip -- the batch pipeline behind the C ABI: BinarySeeding -> ExtractSeeds -> SoC sweep +
// Harmonization -> NeedlemanWunsch (job enumeration, batched ksw, stitch) -> MappingQuality, wired as in
// libMA::setUpCompGraph (libs/ma/src/util/export.cpp:104-108) but over a whole batch of reads resident
// in HBM.  Every stage is a HIP kernel; there is no host fallback.
#include "chain.h"
#include "wave_sort.h"
#include "ksw_launch.h"
#include "nw.h"
#include "seeding.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstring>
#include <chrono>
#include <memory>
#include <mutex>

using namespace ma;

// ------------------------------------------------------------------------------------------------
// device-side counters of a batch
// ------------------------------------------------------------------------------------------------
enum : int
{
    CTR_SEG_USED = 0, // segment pool bump pointer
    CTR_NEXT_READ = 1, // seeding read queue
    CTR_STEPS = 2, // extend_backward steps
    CTR_BLOCKS = 3, // distinct occ blocks touched
    CTR_LF_STEPS = 4,
    CTR_SA_ROWS = 5,
    CTR_HSEED_USED = 6, // harmonized seed pool bump pointer
    CTR_ERR = 7,
    CTR_CIG_USED = 8,
    CTR_CELLS = 9,
    CTR_KSW_JOBS = 10,
    CTR_NEXT_SLOT = 11,
    CTR_MAX_STATE = 12, // ksw sizing (atomicMax)
    CTR_MAX_H = 13,
    CTR_MAX_P = 14,
    CTR_MAX_CIG = 15,
    CTR_N_JOBS = 16,
    CTR_N_ALIGNED = 17,
    CTR_SEQ_BYTES = 18, // sum of qlen+tlen over DP jobs
    CTR_PATH_BYTES = 19, // back-trace steps (direction bytes read back)
    CTR_CLS0 = 20, // DP jobs per kernel class (KSW_N_CLASSES = 15 consecutive words)
    CTR_MAX_QLEN = 35,
    CTR_NEXT_SLOTS = 36, // 28 x u32 job queues / list counters of the ksw launches (14 words; ksw_run_all: `next`)
    CTR_N_REDO = 50, // u32: jobs the extension kernels (and the band of 120) handed back
    CTR_CIG_WORDS = 51, // cigar words written (CTR_CIG_USED counts pool words reserved)
    CTR_NEXT_SEED = 52, // queue of k_lf_walk
    CTR_OPS_ALL = 53, // alignment ops of all alignments / of the MappingQuality selection (exact sizes of the downloads)
    CTR_OPS_MQ = 54,
    CTR_ALN_MQ = 55, // alignments MappingQuality keeps
    CTR_MAX_PC0 = 56, // per kernel class: largest direction-byte scratch of a job (15 words) ...
    CTR_MAX_CIGC0 = 71, // ... and largest cigar scratch in words (15 words)
    CTR_MAX_P_REDO = 86, // the same two for the extension kernels' jobs if they are handed back to the exact kernel
    CTR_MAX_CIG_REDO = 87,
    CTR_NEXT_BIG = 88, // 4 x u32 job queues of the second (few waves, large scratch) launch of a class (2 words)
    CTR_N_1X1 = 90, // 1 x 1 gap fills answered by k_dp_enum itself (counted as ksw calls of one cell each)
    CTR_MAX_BANDL = 91, // largest min(qlen, tlen) of the long jobs on the band of 120 (scratch rows of their launch)
    CTR_COUNT = 92
};

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
#include "stage_seed.h"

#include "stage_extract.h"

#include "stage_chain.h"

#include "stage_dp.h"

#include "stage_output.h"

// ------------------------------------------------------------------------------------------------
// batch object
// ------------------------------------------------------------------------------------------------
struct ma_batch
{
    const ma_index* idx = nullptr;
    int device = 0; // the index's device: every entry point binds the calling thread to it
    ma_params P;
    hipStream_t stream = nullptr;
    u64 max_reads = 0, max_bases = 0;
    u64 n_reads = 0, n_bases = 0;
    u32 max_qlen = 0;
    bool reads_external = false;
    const uint8_t* d_reads = nullptr;
    const u64* d_roff = nullptr;
    DevBuf reads, roff, ctr, seedStack, seedRow, seedSteps, seedSeg, hlocal, hdense, hseedCnt, hseedOff;
    // seeding
    DevBuf stage, smemA, smemB, segPool, segRead, segOff, segCnt, memsCnt, memsOff;
    DevBuf taskA, taskB, taskCnt, taskKey, taskKey2, taskPerm, taskPerm2, taskStage; // (taskStage: per-lane segments of an SMEM task) // area tasks of long reads (seed_tasks)
    u64 segPoolCap = 0, segPoolMin = 0;
    // extraction
    DevBuf segSeedCnt, segSeedOff, seedOff, seedCnt, seeds, cubTmp;
    u64 nSegs = 0, nSeeds = 0;
    // chaining
    DevBuf cWork, cMax, cMm, cA, cB, cOut, cSh1, cSh2, cVx, cVy, cMed, cInl, cBest, hpool, setTab, nsets, hsetOff,
        hsetFlat, hsetRead;
    u64 hpoolCap = 0, nHsets = 0, nHseeds = 0;
    DevBuf socIn, socInCnt; // SoC queues swept elsewhere (ma_batch_set_soc_heap), carved by seed offset
    DevBuf preNmx, preSorted; // long reads: strips per read after k_soc_windows, which sorts k_sort_seeds_wave did
    bool socGiven = false;
    // dp
    DevBuf jobs, info, ez, cigOff, cigPool, kswScratch, clsLists, opsCap, opsOff, ops, hdr, order, mqOrder, mqCnt;
    DevBuf outCnt, outOps, outAlnOff, outOpsOff, outAlns, outOpsPairs; // packed results (get_alns)
    DevBuf sortKey, sortKey2, sortVal2; // longest-job-first order of the DP job lists
    u64 cigPoolCap = 0, cigPoolMin = 0, nOpsCap = 0, nJobSlots = 0;
    KswSide kswSide; // created on first use
#if defined( MA_EXP_DP_PRIO ) // experiment build: the DP kernels of a batch on a stream of the lowest priority (launch_dp.h)
    hipStream_t dpLow = nullptr;
    hipEvent_t dpFork = nullptr, dpJoin = nullptr;
#endif
    // double-buffered I/O (ma_batch_stage_reads / ma_batch_start_mapq_download): the next reads are uploaded into reads2 / roff2
    // and the packed results of the last step downloaded on ioStream while the batch's own stream runs kernels
    DevBuf reads2, roff2;
    hipStream_t ioStream = nullptr;
    hipEvent_t evReadsFree = nullptr, evStaged = nullptr, evPacked = nullptr, evDown = nullptr;
    bool stagedPending = false, downPending = false;
    u64 stN = 0, stBases = 0;
    u32 stMaxQ = 0;
    int stage_done = 0; // 0 none, 1 seeded, 2 extracted, 3 chained, 4 dp
    bool timing = false;
    bool blocking = false; // batch_wait: sleep on an event instead of spinning
    hipEvent_t waitEv = nullptr;
    hipEvent_t ev[ 16 ];
    bool evInit = false;
    float kms[ 8 ] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    float hostMs[ 8 ] = { 0, 0, 0, 0, 0, 0, 0, 0 }; // wall time of the last stage calls on the host: seed, extract, chain, dp
    unsigned long long hctr[ CTR_COUNT ];
};

// Waits for the batch's stream.  Default: hipStreamSynchronize (the runtime spins: lowest latency, one host core busy).
// With blocking waits (ma_batch_set_blocking_sync) the host thread sleeps on an interrupt-driven event instead: the mode for
// hosts that run many more threads than cores (the per-read graph funnel), where a spinning runner burns the time slice
// it then lacks for the next launch.
static int batch_wait( ma_batch* b )
{
    if( !b->blocking )
    {
        MA_HIP( hipStreamSynchronize( b->stream ) );
        return 0;
    }
    if( !b->waitEv )
        MA_HIP( hipEventCreateWithFlags( &b->waitEv, hipEventBlockingSync | hipEventDisableTiming ) );
    MA_HIP( hipEventRecord( b->waitEv, b->stream ) );
    MA_HIP( hipEventSynchronize( b->waitEv ) );
    return 0;
}

static int read_ctr( ma_batch* b )
{
    MA_HIP( hipMemcpyAsync( b->hctr, b->ctr.p, sizeof( b->hctr ), hipMemcpyDeviceToHost, b->stream ) );
    if( batch_wait( b ) )
        return 1;
    return 0;
}

static int check_err( ma_batch* b, const char* where )
{
    const u32 e = (u32)b->hctr[ CTR_ERR ];
    if( !e )
        return 0;
    std::string s = std::string( where ) + ": device capacity overflow:";
    if( e & MA_ERR_SEG_OVERFLOW )
        s += " segments";
    if( e & MA_ERR_SEED_OVERFLOW )
        s += " seeds";
    if( e & MA_ERR_STACK_OVERFLOW )
        s += " interval-stack";
    if( e & MA_ERR_CIGAR_OVERFLOW )
        s += " cigar";
    if( e & MA_ERR_OPS_OVERFLOW )
        s += " alignment-ops";
    if( e & MA_ERR_SCRATCH_OVERFLOW )
        s += " scratch";
    if( e & MA_ERR_SMEM_OVERFLOW )
        s += " smem-lists";
    return fail( s );
}

template <typename T> static int scan_exclusive( ma_batch* b, const T* in, T* out, u64 n )
{
    size_t tb = 0;
    MA_HIP( hipcub::DeviceScan::ExclusiveSum( nullptr, tb, in, out, (int)n, b->stream ) );
    if( b->cubTmp.reserve( tb + 256 ) )
        return 1;
    MA_HIP( hipcub::DeviceScan::ExclusiveSum( b->cubTmp.p, tb, in, out, (int)n, b->stream ) );
    return 0;
}

struct EvTimer
{
    ma_batch* b;
    int slot;
    EvTimer( ma_batch* b_, int s ) : b( b_ ), slot( s )
    {
        if( b->timing )
            (void)hipEventRecord( b->ev[ 2 * slot ], b->stream );
    }
    ~EvTimer( )
    {
        if( b->timing )
            (void)hipEventRecord( b->ev[ 2 * slot + 1 ], b->stream );
    }
};

extern "C" {

int ma_batch_create( const ma_index* idx, const ma_params* P, uint64_t max_reads, uint64_t max_bases, ma_batch** out )
{
    if( !idx || !P || !out )
        return fail( "ma_batch_create: null argument" );
    if( P->seeding_technique < 0 || P->seeding_technique > 2 )
        return fail( "ma_batch_create: unknown seeding technique " + std::to_string( P->seeding_technique ) +
                     " (0 maxSpan, 1 SMEMs, 2 MEMs; binarySeeding.h:560-561)" );
    MA_BIND_DEVICE( idx->device );
    std::unique_ptr<ma_batch> b( new ma_batch( ) );
    b->idx = idx;
    b->device = idx->device;
    b->P = *P;
    b->max_reads = max_reads;
    b->max_bases = max_bases;
    if( b->ctr.reserve( CTR_COUNT * 8 ) || b->reads.reserve( max_bases + 64 ) || b->roff.reserve( ( max_reads + 1 ) * 8 ) )
        return 1;
    *out = b.release( );
    return 0;
}

int ma_batch_destroy( ma_batch* b )
{
    if( !b )
        return 0;
    MA_BIND_DEVICE( b->device );
    if( b->evInit )
        for( int i = 0; i < 16; i++ )
            (void)hipEventDestroy( b->ev[ i ] );
    if( b->waitEv )
        (void)hipEventDestroy( b->waitEv );
    if( b->ioStream )
    {
        (void)hipStreamSynchronize( b->ioStream );
        (void)hipStreamDestroy( b->ioStream );
        for( hipEvent_t e : { b->evReadsFree, b->evStaged, b->evPacked, b->evDown } )
            if( e )
                (void)hipEventDestroy( e );
    }
    if( b->kswSide.fork )
    {
        (void)hipEventDestroy( b->kswSide.fork );
        if( b->kswSide.band )
            (void)hipEventDestroy( b->kswSide.band );
        for( int l = 0; l < 3; l++ )
        {
            if( b->kswSide.stream[ l ] )
                (void)hipStreamDestroy( b->kswSide.stream[ l ] );
            if( b->kswSide.join[ l ] )
                (void)hipEventDestroy( b->kswSide.join[ l ] );
        }
    }
    delete b;
    return 0;
}

int ma_batch_set_stream( ma_batch* b, void* s )
{
    if( !b )
        return fail( "ma_batch_set_stream: null batch" );
    b->stream = (hipStream_t)s;
    return 0;
}

int ma_batch_set_blocking_sync( ma_batch* b, int on )
{
    if( !b )
        return fail( "ma_batch_set_blocking_sync: null batch" );
    b->blocking = on != 0;
    return 0;
}

int ma_batch_enable_timing( ma_batch* b, int on )
{
    if( !b )
        return fail( "null batch" );
    MA_BIND_DEVICE( b->device );
    if( on && !b->evInit )
    {
        for( int i = 0; i < 16; i++ )
            MA_HIP( hipEventCreate( &b->ev[ i ] ) );
        b->evInit = true;
    }
    b->timing = on != 0;
    return 0;
}

// ---- double-buffered I/O ------------------------------------------------------------------------------------------------
static int io_init( ma_batch* b )
{
    if( b->ioStream )
        return 0;
    MA_HIP( hipStreamCreateWithFlags( &b->ioStream, hipStreamNonBlocking ) );
    const unsigned fl = hipEventDisableTiming | ( b->blocking ? hipEventBlockingSync : 0u );
    MA_HIP( hipEventCreateWithFlags( &b->evReadsFree, hipEventDisableTiming ) );
    MA_HIP( hipEventCreateWithFlags( &b->evStaged, fl ) );
    MA_HIP( hipEventCreateWithFlags( &b->evPacked, hipEventDisableTiming ) );
    MA_HIP( hipEventCreateWithFlags( &b->evDown, fl ) );
    return 0;
}

int ma_batch_stage_reads( ma_batch* b, const uint8_t* codes, const uint64_t* offsets, uint64_t n )
{
    if( !b || !offsets || ( n && !codes ) )
        return fail( "ma_batch_stage_reads: null argument" );
    MA_BIND_DEVICE( b->device );
    if( n > b->max_reads || offsets[ n ] > b->max_bases )
        return fail( "ma_batch_stage_reads: batch capacity exceeded" );
    if( b->stagedPending )
        return fail( "ma_batch_stage_reads: the reads staged before were not taken (ma_batch_use_staged_reads)" );
    if( io_init( b ) || b->reads2.reserve( b->max_bases + 64 ) || b->roff2.reserve( ( b->max_reads + 1 ) * 8 ) )
        return 1;
    u32 mq = 0;
    for( u64 i = 0; i < n; i++ )
        mq = std::max<u32>( mq, (u32)( offsets[ i + 1 ] - offsets[ i ] ) );
    b->stN = n, b->stBases = offsets[ n ], b->stMaxQ = mq;
    // the second buffer held the reads of the step before the current one: everything that was enqueued on the batch's stream
    // up to the last swap may still read it
    if( b->evReadsFree )
        MA_HIP( hipStreamWaitEvent( b->ioStream, b->evReadsFree, 0 ) );
    if( b->stBases )
        MA_HIP( hipMemcpyAsync( b->reads2.p, codes, b->stBases, hipMemcpyHostToDevice, b->ioStream ) );
    MA_HIP( hipMemcpyAsync( b->roff2.p, offsets, ( n + 1 ) * 8, hipMemcpyHostToDevice, b->ioStream ) );
    MA_HIP( hipEventRecord( b->evStaged, b->ioStream ) );
    b->stagedPending = true;
    return 0;
}

int ma_batch_use_staged_reads( ma_batch* b )
{
    if( !b )
        return fail( "ma_batch_use_staged_reads: null batch" );
    if( !b->stagedPending )
        return fail( "ma_batch_use_staged_reads: no reads staged (ma_batch_stage_reads)" );
    MA_BIND_DEVICE( b->device );
    MA_HIP( hipEventSynchronize( b->evStaged ) );
    MA_HIP( hipEventRecord( b->evReadsFree, b->stream ) ); // what runs on the stream now is the last reader of the old buffer
    std::swap( b->reads, b->reads2 );
    std::swap( b->roff, b->roff2 );
    b->n_reads = b->stN;
    b->n_bases = b->stBases;
    b->max_qlen = b->stMaxQ;
    b->d_reads = b->reads.as<uint8_t>( );
    b->d_roff = b->roff.as<u64>( );
    b->reads_external = false;
    b->stage_done = 0;
    b->stagedPending = false;
    return 0;
}

int ma_batch_set_reads( ma_batch* b, const uint8_t* codes, const uint64_t* offsets, uint64_t n )
{
    if( !b || !offsets || ( n && !codes ) )
        return fail( "ma_batch_set_reads: null argument" );
    MA_BIND_DEVICE( b->device );
    if( n > b->max_reads || offsets[ n ] > b->max_bases )
        return fail( "ma_batch_set_reads: batch capacity exceeded" );
    b->n_reads = n;
    b->n_bases = offsets[ n ];
    u32 mq = 0;
    for( u64 i = 0; i < n; i++ )
        mq = std::max<u32>( mq, (u32)( offsets[ i + 1 ] - offsets[ i ] ) );
    b->max_qlen = mq;
    if( b->n_bases )
        MA_HIP( hipMemcpyAsync( b->reads.p, codes, b->n_bases, hipMemcpyHostToDevice, b->stream ) );
    MA_HIP( hipMemcpyAsync( b->roff.p, offsets, ( n + 1 ) * 8, hipMemcpyHostToDevice, b->stream ) );
    if( batch_wait( b ) )
        return 1;
    b->d_reads = b->reads.as<uint8_t>( );
    b->d_roff = b->roff.as<u64>( );
    b->reads_external = false;
    b->stage_done = 0;
    return 0;
}

// longest read of the batch.  A million same-address atomics serialise in L2 (182 us per 1 M reads in
// profiles/r05_step_timeline_150bp.txt; one per wavefront: still 16 k of them, 180 us): a fixed grid strides over the reads and
// every wavefront issues ONE atomic (1 024 in all).
__global__ void k_max_qlen( const u64* roff, u64 n, unsigned long long* out )
{
    u32 v = 0;
    for( u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x )
        v = max( v, (u32)std::min<u64>( roff[ i + 1 ] - roff[ i ], 0xffffffffull ) );
    for( int m = 32; m; m >>= 1 )
        v = max( v, (u32)__shfl_xor( (int)v, m, 64 ) );
    if( ( threadIdx.x & 63 ) == 0 && v )
        atomicMax( out, (unsigned long long)v );
}

int ma_batch_set_reads_device( ma_batch* b, const void* d_codes, const void* d_offsets, uint64_t n, uint64_t n_bases )
{
    if( !b || !d_offsets || ( n && !d_codes ) )
        return fail( "ma_batch_set_reads_device: null argument" );
    MA_BIND_DEVICE( b->device );
    b->n_reads = n;
    b->n_bases = n_bases;
    b->d_reads = (const uint8_t*)d_codes;
    b->d_roff = (const u64*)d_offsets;
    b->reads_external = true;
    MA_HIP( hipMemsetAsync( b->ctr.p, 0, CTR_COUNT * 8, b->stream ) );
    if( n )
        hipLaunchKernelGGL( k_max_qlen, dim3( (unsigned)std::min<u64>( 256, ( n + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream, b->d_roff, n,
                            b->ctr.as<unsigned long long>( ) );
    if( read_ctr( b ) )
        return 1;
    b->max_qlen = (u32)b->hctr[ 0 ];
    b->stage_done = 0;
    return 0;
}

#include "launch_seed.h"

#include "launch_extract.h"

#include "launch_chain.h"

#include "launch_dp.h"

// ---- streams -----------------------------------------------------------------------------------------------
int ma_stream_create( const ma_index* x, void** out )
{
    if( !x || !out )
        return fail( "ma_stream_create: null argument" );
    MA_BIND_DEVICE( x->device );
    hipStream_t s = nullptr;
    MA_HIP( hipStreamCreateWithFlags( &s, hipStreamNonBlocking ) );
    *out = (void*)s;
    return 0;
}
int ma_stream_destroy( const ma_index* x, void* s )
{
    if( !x )
        return fail( "ma_stream_destroy: null index" );
    if( !s )
        return 0;
    MA_BIND_DEVICE( x->device );
    MA_HIP( hipStreamDestroy( (hipStream_t)s ) );
    return 0;
}

// ---- stage inputs from the host -------------------------------------------------------------------------------
static int reset_ctr( ma_batch* b )
{
    MA_HIP( hipMemsetAsync( b->ctr.p, 0, CTR_COUNT * 8, b->stream ) );
    return 0;
}
// per-read counts (u32) and exclusive offsets (u64) from a host CSR
static void csr_parts( const uint64_t* off, u64 n, std::vector<u64>& o, std::vector<u32>& c )
{
    o.resize( n + 1 );
    c.resize( n + 1 );
    for( u64 r = 0; r < n; r++ )
    {
        o[ r ] = off[ r ];
        c[ r ] = (u32)( off[ r + 1 ] - off[ r ] );
    }
    o[ n ] = off[ n ];
    c[ n ] = 0;
}

int ma_batch_set_segments( ma_batch* b, const uint64_t* seg_off, const ma_segment* segs )
{
    if( !b || !b->d_roff || !seg_off || ( seg_off[ b->n_reads ] && !segs ) )
        return fail( "ma_batch_set_segments: no reads set or null argument" );
    MA_BIND_DEVICE( b->device );
    const u64 n = b->n_reads, ns = seg_off[ n ];
    std::vector<u64> o;
    std::vector<u32> c, rd( ns + 1 );
    csr_parts( seg_off, n, o, c );
    for( u64 r = 0; r < n; r++ )
        for( u64 k = seg_off[ r ]; k < seg_off[ r + 1 ]; k++ )
            rd[ k ] = (u32)r;
    b->segPoolCap = std::max<u64>( ns + 1024, b->segPoolCap );
    if( reset_ctr( b ) || b->segPool.reserve( b->segPoolCap * sizeof( ma_segment ) ) || b->segRead.reserve( b->segPoolCap * 4 ) ||
        b->segOff.reserve( ( n + 1 ) * 8 ) || b->segCnt.reserve( ( n + 1 ) * 4 ) )
        return 1;
    if( ns )
    {
        MA_HIP( hipMemcpyAsync( b->segPool.p, segs, ns * sizeof( ma_segment ), hipMemcpyHostToDevice, b->stream ) );
        MA_HIP( hipMemcpyAsync( b->segRead.p, rd.data( ), ns * 4, hipMemcpyHostToDevice, b->stream ) );
    }
    MA_HIP( hipMemcpyAsync( b->segOff.p, o.data( ), ( n + 1 ) * 8, hipMemcpyHostToDevice, b->stream ) );
    MA_HIP( hipMemcpyAsync( b->segCnt.p, c.data( ), ( n + 1 ) * 4, hipMemcpyHostToDevice, b->stream ) );
    const unsigned long long used = ns;
    MA_HIP( hipMemcpyAsync( b->ctr.as<unsigned long long>( ) + CTR_SEG_USED, &used, 8, hipMemcpyHostToDevice, b->stream ) );
    if( batch_wait( b ) )
        return 1; // the host vectors go out of scope
    b->nSegs = ns;
    b->nSeeds = b->nHsets = b->nHseeds = 0;
    b->stage_done = 1;
    return 0;
}

int ma_batch_set_seeds( ma_batch* b, const uint64_t* seed_off, const ma_seed* seeds )
{
    if( !b || !b->d_roff || !seed_off || ( seed_off[ b->n_reads ] && !seeds ) )
        return fail( "ma_batch_set_seeds: no reads set or null argument" );
    MA_BIND_DEVICE( b->device );
    const u64 n = b->n_reads, total = seed_off[ n ];
    std::vector<u64> o;
    std::vector<u32> c;
    csr_parts( seed_off, n, o, c );
    if( reset_ctr( b ) || b->seeds.reserve( ( total + 1 ) * sizeof( ma_seed ) ) || b->seedOff.reserve( ( n + 1 ) * 8 ) ||
        b->seedCnt.reserve( ( n + 1 ) * 4 ) )
        return 1;
    if( total )
        MA_HIP( hipMemcpyAsync( b->seeds.p, seeds, total * sizeof( ma_seed ), hipMemcpyHostToDevice, b->stream ) );
    MA_HIP( hipMemcpyAsync( b->seedOff.p, o.data( ), ( n + 1 ) * 8, hipMemcpyHostToDevice, b->stream ) );
    MA_HIP( hipMemcpyAsync( b->seedCnt.p, c.data( ), ( n + 1 ) * 4, hipMemcpyHostToDevice, b->stream ) );
    if( batch_wait( b ) )
        return 1;
    b->nSeeds = total;
    b->nHsets = b->nHseeds = 0;
    b->socGiven = false;
    b->stage_done = 2;
    return 0;
}

// A SoC queue per read that was swept elsewhere -- the reference's StripOfConsideration in a mixed graph -- as input of
// ma_chain_batch, which then only harmonizes: sorted_seeds = the read's seeds as rectangularSoC left them (pSeeds of the
// SoCPriorityQueue), socs = its array vMaxima (score triple + seed range), both CSR per read (the layout
// ma_batch_get_soc_heap returns).
int ma_batch_set_soc_heap( ma_batch* b, const uint64_t* soc_off, const ma_soc* socs, const uint64_t* seed_off, const ma_seed* sorted_seeds )
{
    if( !b || !b->d_roff || !soc_off || !seed_off )
        return fail( "ma_batch_set_soc_heap: no reads set or null argument" );
    const u64 n = b->n_reads;
    if( ( soc_off[ n ] && !socs ) || ( seed_off[ n ] && !sorted_seeds ) )
        return fail( "ma_batch_set_soc_heap: null argument" );
    for( u64 r = 0; r < n; r++ )
        if( soc_off[ r + 1 ] - soc_off[ r ] > seed_off[ r + 1 ] - seed_off[ r ] )
            return fail( "ma_batch_set_soc_heap: a read has more strips than seeds" );
    if( ma_batch_set_seeds( b, seed_off, sorted_seeds ) )
        return 1;
    const u64 total = seed_off[ n ];
    std::vector<ma_soc> q( total + 1 ); // the strips of read r at its SEED offset: the carving every chain scratch uses
    std::vector<u32> c( n + 1 );
    for( u64 r = 0; r < n; r++ )
    {
        c[ r ] = (u32)( soc_off[ r + 1 ] - soc_off[ r ] );
        for( u32 k = 0; k < c[ r ]; k++ )
            q[ seed_off[ r ] + k ] = socs[ soc_off[ r ] + k ];
    }
    if( b->socIn.reserve( ( total + 1 ) * sizeof( ma_soc ) ) || b->socInCnt.reserve( ( n + 1 ) * 4 ) )
        return 1;
    if( total )
        MA_HIP( hipMemcpyAsync( b->socIn.p, q.data( ), total * sizeof( ma_soc ), hipMemcpyHostToDevice, b->stream ) );
    MA_HIP( hipMemcpyAsync( b->socInCnt.p, c.data( ), ( n + 1 ) * 4, hipMemcpyHostToDevice, b->stream ) );
    if( batch_wait( b ) )
        return 1;
    b->socGiven = true;
    return 0;
}

int ma_batch_set_hsets( ma_batch* b, const uint64_t* hset_off, const uint64_t* hseed_off, const uint32_t* hset_soc,
                        const ma_seed* hseeds )
{
    if( !b || !b->d_roff || !hset_off )
        return fail( "ma_batch_set_hsets: no reads set or null argument" );
    const u64 n = b->n_reads, nh = hset_off[ n ];
    if( nh && ( !hseed_off || !hset_soc ) )
        return fail( "ma_batch_set_hsets: null argument" );
    const u64 nhs = nh ? hseed_off[ nh ] : 0;
    if( nhs && !hseeds )
        return fail( "ma_batch_set_hsets: null argument" );
    MA_BIND_DEVICE( b->device );
    std::vector<HSet> flat( nh + 1 );
    std::vector<u32> rd( nh + 1 );
    for( u64 r = 0; r < n; r++ )
        for( u64 s = hset_off[ r ]; s < hset_off[ r + 1 ]; s++ )
        {
            flat[ s ].off = hseed_off[ s ];
            flat[ s ].cnt = (u32)( hseed_off[ s + 1 ] - hseed_off[ s ] );
            flat[ s ].soc = hset_soc[ s ];
            rd[ s ] = (u32)r;
        }
    if( reset_ctr( b ) || b->hsetFlat.reserve( ( nh + 1 ) * sizeof( HSet ) ) || b->hsetRead.reserve( ( nh + 1 ) * 4 ) ||
        b->hdense.reserve( ( nhs + 1 ) * sizeof( ma_seed ) ) || b->hsetOff.reserve( ( n + 2 ) * 8 ) )
        return 1;
    if( nh )
    {
        MA_HIP( hipMemcpyAsync( b->hsetFlat.p, flat.data( ), nh * sizeof( HSet ), hipMemcpyHostToDevice, b->stream ) );
        MA_HIP( hipMemcpyAsync( b->hsetRead.p, rd.data( ), nh * 4, hipMemcpyHostToDevice, b->stream ) );
    }
    if( nhs )
        MA_HIP( hipMemcpyAsync( b->hdense.p, hseeds, nhs * sizeof( ma_seed ), hipMemcpyHostToDevice, b->stream ) );
    MA_HIP( hipMemcpyAsync( b->hsetOff.p, hset_off, ( n + 1 ) * 8, hipMemcpyHostToDevice, b->stream ) );
    if( batch_wait( b ) )
        return 1;
    b->nHsets = nh;
    b->nHseeds = nhs;
    b->stage_done = 3;
    return 0;
}

// ---- the SoC queue across the boundary ------------------------------------------------------------------------
static int get_socs( ma_batch* b, int heap_layout, uint64_t* n_socs, uint64_t* soc_off, ma_soc* socs, uint64_t* seed_off,
                     ma_seed* sorted_seeds );
int ma_batch_get_socs( ma_batch* b, uint64_t* n_socs, uint64_t* soc_off, ma_soc* socs, uint64_t* seed_off, ma_seed* sorted_seeds )
{
    return get_socs( b, 0, n_socs, soc_off, socs, seed_off, sorted_seeds );
}
int ma_batch_get_soc_heap( ma_batch* b, uint64_t* n_socs, uint64_t* soc_off, ma_soc* socs, uint64_t* seed_off, ma_seed* sorted_seeds )
{
    return get_socs( b, 1, n_socs, soc_off, socs, seed_off, sorted_seeds );
}
} // extern "C"
static int get_socs( ma_batch* b, int heap_layout, uint64_t* n_socs, uint64_t* soc_off, ma_soc* socs, uint64_t* seed_off,
                     ma_seed* sorted_seeds )
{
    if( !b || b->stage_done < 2 )
        return fail( "ma_batch_get_socs: run ma_extract_seeds_batch first" );
    MA_BIND_DEVICE( b->device );
    const u64 n = b->n_reads, ts = b->nSeeds + 1;
    if( n_socs )
        *n_socs = 0;
    if( soc_off )
        soc_off[ 0 ] = 0;
    if( seed_off )
        seed_off[ 0 ] = 0;
    if( n == 0 )
        return 0;
    DevBuf dSocs, dN;
    if( b->cWork.reserve( ts * sizeof( ma_seed ) ) || b->cMax.reserve( ts * sizeof( SoCEntry ) ) ||
        b->cMm.reserve( ts * sizeof( RefMinMax ) ) || dSocs.reserve( ts * sizeof( ma_soc ) ) || dN.reserve( ( n + 1 ) * 4 ) )
        return 1;
    hipLaunchKernelGGL( k_soc_dump, dim3( (unsigned)( ( n + 63 ) / 64 ) ), dim3( 64 ), 0, b->stream, b->idx->v, chain_params( b->P ),
                        (u32)n, b->d_roff, b->seedOff.as<u64>( ), b->seedCnt.as<u32>( ), b->seeds.as<ma_seed>( ),
                        b->cWork.as<ma_seed>( ), b->cMax.as<SoCEntry>( ), b->cMm.as<RefMinMax>( ), dSocs.as<ma_soc>( ),
                        dN.as<u32>( ), heap_layout );
    MA_HIP( hipGetLastError( ) );
    std::vector<u32> cnt( n ), scnt( n );
    std::vector<u64> off( n );
    MA_HIP( hipMemcpyAsync( cnt.data( ), dN.p, n * 4, hipMemcpyDeviceToHost, b->stream ) );
    MA_HIP( hipMemcpyAsync( scnt.data( ), b->seedCnt.p, n * 4, hipMemcpyDeviceToHost, b->stream ) );
    MA_HIP( hipMemcpyAsync( off.data( ), b->seedOff.p, n * 8, hipMemcpyDeviceToHost, b->stream ) );
    if( batch_wait( b ) )
        return 1;
    u64 total = 0;
    for( u64 r = 0; r < n; r++ )
        total += cnt[ r ];
    if( n_socs )
        *n_socs = total;
    if( !soc_off && !socs && !seed_off && !sorted_seeds )
        return 0;
    std::vector<ma_soc> hs( b->nSeeds + 1 );
    std::vector<ma_seed> hw( b->nSeeds + 1 );
    if( b->nSeeds )
    {
        MA_HIP( hipMemcpy( hs.data( ), dSocs.p, b->nSeeds * sizeof( ma_soc ), hipMemcpyDeviceToHost ) );
        MA_HIP( hipMemcpy( hw.data( ), b->cWork.p, b->nSeeds * sizeof( ma_seed ), hipMemcpyDeviceToHost ) );
    }
    u64 so = 0, sd = 0;
    for( u64 r = 0; r < n; r++ )
    {
        if( socs )
            for( u32 k = 0; k < cnt[ r ]; k++ )
                socs[ so + k ] = hs[ off[ r ] + k ];
        if( sorted_seeds )
            for( u32 k = 0; k < scnt[ r ]; k++ )
                sorted_seeds[ sd + k ] = hw[ off[ r ] + k ];
        so += cnt[ r ];
        sd += scnt[ r ];
        if( soc_off )
            soc_off[ r + 1 ] = so;
        if( seed_off )
            seed_off[ r + 1 ] = sd;
    }
    return 0;
}
extern "C" {

int ma_align_batch( ma_batch* b )
{
    if( !b )
        return fail( "ma_align_batch: null batch" );
    int ( *const stage[ 4 ] )( ma_batch* ) = { ma_seed_batch, ma_extract_seeds_batch, ma_chain_batch, ma_dp_batch };
    for( int k = 0; k < 4; k++ )
    {
        const auto t0 = std::chrono::steady_clock::now( );
        if( stage[ k ]( b ) )
            return 1;
        b->hostMs[ k ] = std::chrono::duration<float, std::milli>( std::chrono::steady_clock::now( ) - t0 ).count( );
    }
    if( getenv( "MA_MEM_REPORT" ) ) // diagnostics: device bytes this batch holds after a full pass
    {
        struct Row
        {
            const char* name;
            size_t cap;
        };
#define MA_ROW( x ) Row{ #x, b->x.cap }
        std::vector<Row> rows = { MA_ROW( reads ), MA_ROW( roff ), MA_ROW( seedStack ), MA_ROW( seedRow ), MA_ROW( seedSteps ), MA_ROW( seedSeg ),
            MA_ROW( hlocal ), MA_ROW( hdense ), MA_ROW( hseedCnt ), MA_ROW( hseedOff ), MA_ROW( stage ), MA_ROW( smemA ), MA_ROW( smemB ),
            MA_ROW( segPool ), MA_ROW( segRead ), MA_ROW( segOff ), MA_ROW( segCnt ), MA_ROW( memsCnt ), MA_ROW( memsOff ), MA_ROW( taskA ),
            MA_ROW( taskB ), MA_ROW( taskCnt ), MA_ROW( taskKey ), MA_ROW( taskKey2 ), MA_ROW( taskPerm ), MA_ROW( taskPerm2 ),
            MA_ROW( segSeedCnt ), MA_ROW( segSeedOff ), MA_ROW( seedOff ), MA_ROW( seedCnt ), MA_ROW( seeds ), MA_ROW( cubTmp ), MA_ROW( cWork ),
            MA_ROW( cMax ), MA_ROW( cMm ), MA_ROW( cA ), MA_ROW( cB ), MA_ROW( cOut ), MA_ROW( cSh1 ), MA_ROW( cSh2 ), MA_ROW( cVx ), MA_ROW( cVy ),
            MA_ROW( cMed ), MA_ROW( cInl ), MA_ROW( cBest ), MA_ROW( hpool ), MA_ROW( setTab ), MA_ROW( nsets ), MA_ROW( hsetOff ),
            MA_ROW( hsetFlat ), MA_ROW( hsetRead ), MA_ROW( jobs ), MA_ROW( info ), MA_ROW( ez ), MA_ROW( cigOff ), MA_ROW( cigPool ),
            MA_ROW( kswScratch ), MA_ROW( clsLists ), MA_ROW( opsCap ), MA_ROW( opsOff ), MA_ROW( ops ), MA_ROW( hdr ), MA_ROW( order ),
            MA_ROW( mqOrder ), MA_ROW( mqCnt ), MA_ROW( outCnt ), MA_ROW( outOps ), MA_ROW( outAlnOff ), MA_ROW( outOpsOff ), MA_ROW( outAlns ),
            MA_ROW( outOpsPairs ) };
#undef MA_ROW
        std::sort( rows.begin( ), rows.end( ), []( const Row& a, const Row& c ) { return a.cap > c.cap; } );
        size_t total = 0;
        for( const Row& r : rows )
            total += r.cap;
        fprintf( stderr, "[ma_amd] batch of %llu reads / %llu bases holds %.2f GB on the device:", (unsigned long long)b->n_reads,
                 (unsigned long long)b->n_bases, total / 1e9 );
        for( size_t i = 0; i < rows.size( ) && i < 14; i++ )
            fprintf( stderr, " %s %.2f", rows[ i ].name, rows[ i ].cap / 1e9 );
        fprintf( stderr, "\n" );
    }
    return 0;
}

int ma_batch_host_ms( ma_batch* b, float out[ 8 ] )
{
    if( !b )
        return fail( "null batch" );
    for( int i = 0; i < 8; i++ )
        out[ i ] = b->hostMs[ i ];
    return 0;
}

int ma_batch_sync( ma_batch* b )
{
    if( !b )
        return fail( "ma_batch_sync: null batch" );
    MA_BIND_DEVICE( b->device );
    if( read_ctr( b ) )
        return 1;
    if( b->timing && b->evInit )
        for( int i = 0; i < 6; i++ )
        {
            float ms = 0;
            if( hipEventElapsedTime( &ms, b->ev[ 2 * i ], b->ev[ 2 * i + 1 ] ) == hipSuccess )
                b->kms[ i ] = ms;
        }
    return check_err( b, "ma_batch_sync" );
}

int ma_batch_kernel_ms( ma_batch* b, float out[ 8 ] )
{
    if( !b )
        return fail( "null batch" );
    for( int i = 0; i < 8; i++ )
        out[ i ] = b->kms[ i ];
    return 0;
}

int ma_batch_counters( ma_batch* b, uint64_t out[ 8 ] )
{
    if( ma_batch_sync( b ) )
        return 1;
    out[ 0 ] = b->hctr[ CTR_STEPS ];
    out[ 1 ] = b->hctr[ CTR_BLOCKS ];
    out[ 2 ] = b->hctr[ CTR_LF_STEPS ];
    out[ 3 ] = b->nSeeds;
    out[ 4 ] = b->hctr[ CTR_CELLS ] + b->hctr[ CTR_N_1X1 ];
    out[ 5 ] = b->hctr[ CTR_KSW_JOBS ] + b->hctr[ CTR_N_1X1 ];
    out[ 6 ] = b->hctr[ CTR_SEQ_BYTES ];
    out[ 7 ] = b->hctr[ CTR_PATH_BYTES ] + 4 * b->hctr[ CTR_CIG_WORDS ] + 5 * b->hctr[ CTR_N_1X1 ]; // (a 1 x 1 job: one back-trace step, one cigar word)
    return 0;
}

int ma_batch_counts( ma_batch* b, uint64_t* n_segments, uint64_t* n_seeds, uint64_t* n_hsets, uint64_t* n_hseeds,
                     uint64_t* n_alignments, uint64_t* n_ops, uint64_t* n_aligned_reads )
{
    if( ma_batch_sync( b ) )
        return 1;
    if( b->stage_done >= 1 && b->nSegs == 0 )
        b->nSegs = b->hctr[ CTR_SEG_USED ];
    if( n_segments )
        *n_segments = b->nSegs;
    if( n_seeds )
        *n_seeds = b->nSeeds;
    if( n_hsets )
        *n_hsets = b->nHsets;
    if( n_hseeds )
        *n_hseeds = b->nHseeds;
    if( n_alignments )
        *n_alignments = b->stage_done >= 4 ? b->nHsets : 0;
    if( n_ops ) // exact: the ops of all alignments (the MappingQuality selection has at most as many)
        *n_ops = b->stage_done >= 4 ? b->hctr[ CTR_OPS_ALL ] : 0;
    if( n_aligned_reads )
        *n_aligned_reads = b->hctr[ CTR_N_ALIGNED ];
    return 0;
}

// Downloads gather the per-read ranges (pools are in completion order) into read-order CSR arrays.
int ma_batch_get_segments( ma_batch* b, uint64_t* seg_off, ma_segment* segs )
{
    if( !b || b->stage_done < 1 )
        return fail( "ma_batch_get_segments: stage not run" );
    MA_BIND_DEVICE( b->device );
    if( ma_batch_sync( b ) )
        return 1;
    const u64 n = b->n_reads, ns = b->hctr[ CTR_SEG_USED ];
    std::vector<u64> off( n );
    std::vector<u32> cnt( n );
    std::vector<ma_segment> pool( ns );
    if( n )
    {
        MA_HIP( hipMemcpy( off.data( ), b->segOff.p, n * 8, hipMemcpyDeviceToHost ) );
        MA_HIP( hipMemcpy( cnt.data( ), b->segCnt.p, n * 4, hipMemcpyDeviceToHost ) );
    }
    if( ns )
        MA_HIP( hipMemcpy( pool.data( ), b->segPool.p, ns * sizeof( ma_segment ), hipMemcpyDeviceToHost ) );
    u64 o = 0;
    for( u64 r = 0; r < n; r++ )
    {
        if( seg_off )
            seg_off[ r ] = o;
        if( segs )
            for( u32 k = 0; k < cnt[ r ]; k++ )
                segs[ o + k ] = pool[ off[ r ] + k ];
        o += cnt[ r ];
    }
    if( seg_off )
        seg_off[ n ] = o;
    return 0;
}

int ma_batch_get_seeds( ma_batch* b, uint64_t* seed_off, ma_seed* seeds )
{
    if( !b || b->stage_done < 2 )
        return fail( "ma_batch_get_seeds: stage not run" );
    MA_BIND_DEVICE( b->device );
    if( ma_batch_sync( b ) )
        return 1;
    const u64 n = b->n_reads;
    std::vector<u64> off( n );
    std::vector<u32> cnt( n );
    std::vector<ma_seed> pool( b->nSeeds );
    if( n )
    {
        MA_HIP( hipMemcpy( off.data( ), b->seedOff.p, n * 8, hipMemcpyDeviceToHost ) );
        MA_HIP( hipMemcpy( cnt.data( ), b->seedCnt.p, n * 4, hipMemcpyDeviceToHost ) );
    }
    if( b->nSeeds )
        MA_HIP( hipMemcpy( pool.data( ), b->seeds.p, b->nSeeds * sizeof( ma_seed ), hipMemcpyDeviceToHost ) );
    u64 o = 0;
    for( u64 r = 0; r < n; r++ )
    {
        if( seed_off )
            seed_off[ r ] = o;
        if( seeds )
            for( u32 k = 0; k < cnt[ r ]; k++ )
                seeds[ o + k ] = pool[ off[ r ] + k ];
        o += cnt[ r ];
    }
    if( seed_off )
        seed_off[ n ] = o;
    return 0;
}

int ma_batch_get_hsets( ma_batch* b, uint64_t* hset_off, uint64_t* hseed_off, uint32_t* hset_soc, ma_seed* hseeds )
{
    if( !b || b->stage_done < 3 )
        return fail( "ma_batch_get_hsets: stage not run" );
    MA_BIND_DEVICE( b->device );
    if( ma_batch_sync( b ) )
        return 1;
    const u64 n = b->n_reads, nh = b->nHsets;
    if( hset_off && n )
        MA_HIP( hipMemcpy( hset_off, b->hsetOff.p, ( n + 1 ) * 8, hipMemcpyDeviceToHost ) );
    if( hset_off && !n )
        hset_off[ 0 ] = 0;
    std::vector<HSet> flat( nh );
    std::vector<ma_seed> pool( b->nHseeds );
    if( nh )
        MA_HIP( hipMemcpy( flat.data( ), b->hsetFlat.p, nh * sizeof( HSet ), hipMemcpyDeviceToHost ) );
    if( b->nHseeds )
        MA_HIP( hipMemcpy( pool.data( ), b->hdense.p, b->nHseeds * sizeof( ma_seed ), hipMemcpyDeviceToHost ) );
    u64 o = 0;
    for( u64 s = 0; s < nh; s++ )
    {
        if( hseed_off )
            hseed_off[ s ] = o;
        if( hset_soc )
            hset_soc[ s ] = flat[ s ].soc;
        if( hseeds )
            for( u32 k = 0; k < flat[ s ].cnt; k++ )
                hseeds[ o + k ] = pool[ flat[ s ].off + k ];
        o += flat[ s ].cnt;
    }
    if( hseed_off )
        hseed_off[ nh ] = o;
    return 0;
}

// async = the double-buffered form: the copies go to the I/O stream behind the pack kernels and nobody waits here
static int get_alns( ma_batch* b, bool mq, uint64_t* aln_off, ma_alignment* alns, uint64_t* ops, bool async = false )
{
    if( !b || b->stage_done < 4 )
        return fail( "ma_batch_get_alignments: stage not run" );
    MA_BIND_DEVICE( b->device );
    if( ma_batch_sync( b ) )
        return 1;
    if( b->downPending )
    {
        // the packed arrays are still being downloaded: nothing may overwrite them before that
        if( async )
            return fail( "ma_batch_start_mapq_download: the download started before was not finished (ma_batch_finish_download)" );
        MA_HIP( hipEventSynchronize( b->evDown ) );
        b->downPending = false;
    }
    if( async && io_init( b ) )
        return 1;
    const u64 n = b->n_reads, nh = b->nHsets;
    if( aln_off )
        aln_off[ 0 ] = 0;
    if( n == 0 )
        return 0;
    if( nh == 0 )
    {
        if( aln_off )
            memset( aln_off, 0, ( n + 1 ) * 8 );
        return 0;
    }
    const u64 totalA = mq ? b->hctr[ CTR_ALN_MQ ] : nh, totalO = mq ? b->hctr[ CTR_OPS_MQ ] : b->hctr[ CTR_OPS_ALL ];
    if( b->outCnt.reserve( ( n + 2 ) * 8 ) || b->outOps.reserve( ( n + 2 ) * 8 ) || b->outAlnOff.reserve( ( n + 2 ) * 8 ) ||
        b->outOpsOff.reserve( ( n + 2 ) * 8 ) || b->outAlns.reserve( ( totalA + 1 ) * sizeof( ma_alignment ) ) ||
        b->outOpsPairs.reserve( ( totalO + 1 ) * 16 ) )
        return 1;
    const u32* ord = mq ? b->mqOrder.as<u32>( ) : b->order.as<u32>( );
    const dim3 grid( (unsigned)( ( n + 255 ) / 256 ) ), block( 256 );
    hipLaunchKernelGGL( k_aln_sizes, grid, block, 0, b->stream, (u32)n, b->hsetOff.as<u64>( ), b->hdr.as<AlnHeader>( ), ord,
                        b->mqCnt.as<u32>( ), mq ? 1 : 0, b->outCnt.as<u64>( ), b->outOps.as<u64>( ) );
    MA_HIP( hipMemsetAsync( (char*)b->outCnt.p + n * 8, 0, 8, b->stream ) );
    MA_HIP( hipMemsetAsync( (char*)b->outOps.p + n * 8, 0, 8, b->stream ) );
    if( scan_exclusive<u64>( b, b->outCnt.as<u64>( ), b->outAlnOff.as<u64>( ), n + 1 ) ||
        scan_exclusive<u64>( b, b->outOps.as<u64>( ), b->outOpsOff.as<u64>( ), n + 1 ) )
        return 1;
    hipLaunchKernelGGL( k_aln_pack, grid, block, 0, b->stream, (u32)n, b->hsetOff.as<u64>( ), b->hdr.as<AlnHeader>( ), ord,
                        b->ops.as<u64>( ), mq ? 1 : 0, b->outAlnOff.as<u64>( ), b->outOpsOff.as<u64>( ), b->outAlns.as<ma_alignment>( ),
                        b->outOpsPairs.as<u64>( ) );
    MA_HIP( hipGetLastError( ) );
    hipStream_t cs = b->stream;
    if( async )
    {
        cs = b->ioStream;
        MA_HIP( hipEventRecord( b->evPacked, b->stream ) );
        MA_HIP( hipStreamWaitEvent( cs, b->evPacked, 0 ) );
    }
    if( aln_off )
        MA_HIP( hipMemcpyAsync( aln_off, b->outAlnOff.p, ( n + 1 ) * 8, hipMemcpyDeviceToHost, cs ) );
    if( alns && totalA )
        MA_HIP( hipMemcpyAsync( alns, b->outAlns.p, totalA * sizeof( ma_alignment ), hipMemcpyDeviceToHost, cs ) );
    if( ops && totalO )
        MA_HIP( hipMemcpyAsync( ops, b->outOpsPairs.p, totalO * 16, hipMemcpyDeviceToHost, cs ) );
    if( async )
    {
        MA_HIP( hipEventRecord( b->evDown, cs ) );
        b->downPending = true;
        return 0;
    }
    if( batch_wait( b ) )
        return 1;
    return 0;
}

int ma_batch_start_mapq_download( ma_batch* b, uint64_t* aln_off, ma_alignment* alns, uint64_t* ops )
{
    return get_alns( b, true, aln_off, alns, ops, true );
}

int ma_batch_finish_download( ma_batch* b )
{
    if( !b )
        return fail( "ma_batch_finish_download: null batch" );
    if( !b->downPending )
        return 0;
    MA_BIND_DEVICE( b->device );
    MA_HIP( hipEventSynchronize( b->evDown ) );
    b->downPending = false;
    return 0;
}

int ma_batch_get_alignments( ma_batch* b, uint64_t* aln_off, ma_alignment* alns, uint64_t* ops )
{
    return get_alns( b, false, aln_off, alns, ops );
}
int ma_batch_get_mapq_alignments( ma_batch* b, uint64_t* aln_off, ma_alignment* alns, uint64_t* ops )
{
    return get_alns( b, true, aln_off, alns, ops );
}

int ma_batch_get_dp_jobs( ma_batch* b, uint64_t* n_jobs, int32_t* shapes /* 8 x i32 per job */, uint64_t cap )
{
    if( !b || b->stage_done < 4 )
        return fail( "ma_batch_get_dp_jobs: stage not run" );
    MA_BIND_DEVICE( b->device );
    if( ma_batch_sync( b ) )
        return 1;
    const u64 nSlots = b->nJobSlots;
    std::vector<DpJob> jobs( nSlots + 1 );
    std::vector<ma_ez> ez( nSlots + 1 );
    if( nSlots )
    {
        MA_HIP( hipMemcpy( jobs.data( ), b->jobs.p, nSlots * sizeof( DpJob ), hipMemcpyDeviceToHost ) );
        MA_HIP( hipMemcpy( ez.data( ), b->ez.p, nSlots * sizeof( ma_ez ), hipMemcpyDeviceToHost ) );
    }
    u64 n = 0;
    for( u64 i = 0; i < nSlots; i++ )
    {
        const DpJob& j = jobs[ i ];
        if( j.q_to <= j.q_from )
            continue;
        if( shapes && n < cap )
        {
            int32_t* o = shapes + 8 * n;
            o[ 0 ] = (int32_t)( j.q_to - j.q_from );
            o[ 1 ] = (int32_t)( j.r_to - j.r_from );
            o[ 2 ] = j.w;
            o[ 3 ] = j.zdrop;
            o[ 4 ] = j.flag;
            o[ 5 ] = ez[ i ].zdropped;
            o[ 6 ] = ez[ i ].max_q;
            o[ 7 ] = ez[ i ].max_t;
        }
        n++;
    }
    if( n_jobs )
        *n_jobs = n;
    return 0;
}

// diagnostics: what became of the extension jobs tried on the proven narrow band (ksw_band.h) since the library was loaded:
// tried, proved, failed check 1 / 2 / 3 / 4, handed back for another reason, diagonals run
int ma_debug_band_stats( unsigned long long out[ 8 ] )
{
    if( !out )
        return fail( "ma_debug_band_stats: null argument" );
    // (a static __device__ array per translation unit: the pipeline's kernels count here, those of ma_ksw_ext_batch in prims.hip)
    unsigned long long other[ 8 ];
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( ma::g_band_stats ), 8 * 8 ) );
    if( ma::band_stats_of_prims( other ) )
        return 1;
    for( int i = 0; i < 8; i++ )
        out[ i ] += other[ i ];
    return 0;
}

// the same for the LONG extension jobs, one per wavefront on the band of 120 (ksw_band.h, G = 1)
int ma_debug_band_long_stats( unsigned long long out[ 8 ] )
{
    if( !out )
        return fail( "ma_debug_band_long_stats: null argument" );
    unsigned long long other[ 8 ];
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( ma::g_band_stats ), 8 * 8, 8 * 8 ) );
    if( ma::band_long_stats_of_prims( other ) )
        return 1;
    for( int i = 0; i < 8; i++ )
        out[ i ] += other[ i ];
    return 0;
}

// diagnostics: DP cells computed and jobs finished per kernel family since the library was loaded (ksw_launch.h: g_dp_family):
// out[ 2 f ], out[ 2 f + 1 ] for f = 0 k_ksw_ext<1>, 1 k_ksw_ext<2>, 2 k_ksw_grp<2>, 3 k_ksw_grp<4>, 4 k_ksw_band (four short jobs per wave),
// 5 k_ksw_pk, 6 k_ksw, 7 k_ksw_band (one long job per wave)
int ma_debug_dp_family_stats( unsigned long long out[ 16 ] )
{
    if( !out )
        return fail( "ma_debug_dp_family_stats: null argument" );
    unsigned long long other[ 2 * KSW_N_FAMILIES ];
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( ma::g_dp_family ), 2 * KSW_N_FAMILIES * 8 ) );
    if( ma::dp_family_stats_of_prims( other ) )
        return 1;
    for( int i = 0; i < 2 * KSW_N_FAMILIES; i++ )
        out[ i ] += other[ i ];
    return 0;
}

#if defined( MA_CHAIN_PROF )
int ma_debug_chain_prof( unsigned long long* out )
{
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( ma::g_chain_prof ), 16 * 8 ) );
    return 0;
}
#endif

#if defined( MA_KSW_PROF )
int ma_debug_ksw_prof( unsigned long long* out )
{
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( ma::g_ksw_prof ), 16 * 8 ) );
    return 0;
}
int ma_debug_pk_prof( unsigned long long* out )
{
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( ma::g_pk_prof ), 16 * 8 ) );
    return 0;
}
int ma_debug_grp_prof( unsigned long long* out ) // 24 words (ksw_grp.h)
{
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( ma::g_grp_prof ), 24 * 8 ) );
    return 0;
}
int ma_debug_seed_prof( unsigned long long* out )
{
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( g_seed_prof ), 8 * 8 ) );
    return 0;
}
#endif

} // extern "C"
