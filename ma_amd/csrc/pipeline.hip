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
    CTR_CLS0 = 20, // DP jobs per kernel class (KSW_N_CLASSES = 7 consecutive words)
    CTR_MAX_QLEN = 27,
    CTR_NEXT_SLOTS = 28, // 12 x u32 job queues of the ksw launches (6 words)
    CTR_N_REDO = 34, // u32: jobs the extension kernel handed back
    CTR_CIG_WORDS = 35, // cigar words written (CTR_CIG_USED counts pool words reserved)
    CTR_NEXT_SEED = 36, // queue of k_lf_walk
    CTR_OPS_ALL = 37, // alignment ops of all alignments / of the MappingQuality selection (exact sizes of the downloads)
    CTR_OPS_MQ = 38,
    CTR_ALN_MQ = 39, // alignments MappingQuality keeps
    CTR_MAX_PC0 = 40, // per kernel class: largest direction-byte scratch of a job (7 words) ...
    CTR_MAX_CIGC0 = 47, // ... and largest cigar scratch in words (7 words)
    CTR_MAX_P_REDO = 54, // the same two for the extension kernel's jobs if they are handed back to the exact kernel
    CTR_MAX_CIG_REDO = 55,
    CTR_NEXT_BIG = 56, // 4 x u32 job queues of the second (few waves, large scratch) launch of a class (2 words)
    CTR_COUNT = 58
};

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
struct SeedKernelArgs
{
    IndexView X;
    SeedParams P;
    const uint8_t* reads;
    const u64* roff;
    u32 n_reads;
    ma_segment* stage; // lanes * seg_cap
    u32 seg_cap;
    ma_segment* smem_a; // lanes * smem_cap
    ma_segment* smem_b;
    u32 smem_cap;
    u32* stack; // lanes * 2 * MA_SEED_STACK
    u32 q_lds; // LDS bytes per lane for the read in flight (0: reads stay in HBM)
    u32 slow_batch; // lanes of a wave that must wait for a phase transition before the transitions are run
    ma_segment* pool;
    u32* pool_read; // read id per pooled segment
    u64 pool_cap;
    u64* seg_off; // per read
    u32* seg_cnt; // per read
    unsigned long long* ctr;
};

#if defined( MA_KSW_PROF )
static __device__ unsigned long long g_seed_prof[ 8 ];
#endif
// One read per lane at a time; lanes refill from a global queue, so a wavefront keeps stepping 64
// reads in lockstep through extend_backward until the batch is exhausted.
// LONG: the reads stay in HBM (longer than 240 bases) and are read through the register window of seed_qbyte.
template <bool LONG, bool SM, bool MS = !SM> __device__ __forceinline__ void seed_kernel_body( const SeedKernelArgs& A )
{
    const u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    SeedScratch S;
    S.stage = A.stage + (u64)lane * A.seg_cap;
    S.seg_cap = A.seg_cap;
    S.smem_a = A.smem_a ? A.smem_a + (u64)lane * A.smem_cap : nullptr;
    S.smem_b = A.smem_b ? A.smem_b + (u64)lane * A.smem_cap : nullptr;
    S.smem_cap = A.smem_cap;
    S.stack = A.stack + (u64)lane * ( 2 * MA_SEED_STACK );
    S.drop_div = A.P.min_seed_size_drop;
    SeedLane L;
    L.phase = PH_DONE;
    u32 read = 0xffffffffu;
    u64 steps = 0, blocks = 0;
    const u32 wl = threadIdx.x & 63;
#if defined( MA_KSW_PROF )
    unsigned long long pf[ 7 ] = { 0, 0, 0, 0, 0, 0, 0 };
#endif
    // the read in flight is staged in LDS (stride = odd number of words: conflict-free): every step of the state
    // machine starts with a query base, and an LDS read is ~20x closer than an HBM one
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t q_lds[];
    uint8_t* myq = q_lds + (size_t)threadIdx.x * A.q_lds;
    bool alive = true;
    while( true )
    {
        // Flush finished reads and fetch new ones wave-wide: one atomic per wave on the segment pool pointer and on
        // the read queue instead of one per read (same-address atomics serialise in L2).
        const bool done = alive && L.phase == PH_DONE;
        const unsigned long long dm = __ballot( done );
#if defined( MA_KSW_PROF )
        const unsigned long long tA = clock64( );
#endif
        // refill when at least 8 lanes wait (a refill stalls the whole wave for several memory round trips) or when
        // nothing else is left to do
        if( dm && ( __popcll( dm ) >= 8 || dm == __ballot( alive ) ) )
        {
            const bool flush = done && read != 0xffffffffu;
            const u32 n = flush ? seed_finish( L, A.P, S, A.X ) : 0u;
            u32 inc = n;
            for( int d = 1; d < 64; d <<= 1 )
            {
                const u32 o = (u32)__shfl_up( (int)inc, d, 64 );
                if( wl >= (u32)d )
                    inc += o;
            }
            const u32 total = (u32)__shfl( (int)inc, 63, 64 );
            unsigned long long base = 0, rb = 0;
            if( wl == 0 )
            {
                // both in flight before either result is needed
                rb = atomicAdd( &A.ctr[ CTR_NEXT_READ ], (unsigned long long)__popcll( dm ) );
                base = atomicAdd( &A.ctr[ CTR_SEG_USED ], (unsigned long long)total );
            }
            base = ( (u64)(u32)__shfl( (int)( base >> 32 ), 0, 64 ) << 32 ) | (u32)__shfl( (int)(u32)base, 0, 64 );
            rb = ( (u64)(u32)__shfl( (int)( rb >> 32 ), 0, 64 ) << 32 ) | (u32)__shfl( (int)(u32)rb, 0, 64 );
            if( flush )
            {
                const u64 off = base + inc - n;
                if( off + n <= A.pool_cap )
                {
                    const ma_segment* __restrict__ src = S.stage;
                    ma_segment* __restrict__ dst = A.pool + off;
                    u32* __restrict__ dr = A.pool_read + off;
#pragma unroll 4
                    for( u32 k = 0; k < n; k++ )
                    {
                        dst[ k ] = src[ k ];
                        dr[ k ] = read;
                    }
                }
                else
                    L.err |= MA_ERR_SEG_OVERFLOW;
                A.seg_off[ read ] = off;
                A.seg_cnt[ read ] = off + n <= A.pool_cap ? n : 0;
                if( L.err )
                    atomicOr( (unsigned long long*)&A.ctr[ CTR_ERR ], (unsigned long long)L.err );
                steps += L.steps;
                blocks += L.blocks;
            }
            if( done )
            {
                const u64 mine = rb + (u64)__popcll( dm & ( ( 1ull << wl ) - 1 ) );
                if( mine >= A.n_reads )
                    alive = false;
                else
                {
                    read = (u32)mine;
                    const uint8_t* src = A.reads + A.roff[ read ];
                    const u32 ql = (u32)( A.roff[ read + 1 ] - A.roff[ read ] );
                    if( !LONG )
                    {
                        u32 k = 0;
                        for( ; k + 16 <= ql; k += 16 )
                        {
                            uint4 v;
                            __builtin_memcpy( &v, src + k, 16 );
                            u32* d = (u32*)( myq + k ); // 4-byte aligned (stride and k are multiples of 4)
                            d[ 0 ] = v.x, d[ 1 ] = v.y, d[ 2 ] = v.z, d[ 3 ] = v.w;
                        }
                        for( ; k < ql; k++ )
                            myq[ k ] = src[ k ];
                        src = myq;
                    }
                    seed_begin_read( L, src, ql );
                }
            }
        }
        if( __ballot( alive ) == 0 )
            break;
#if defined( MA_KSW_PROF )
        const unsigned long long tB = clock64( );
#endif
        u32 c = 0;
        const bool act = alive && L.phase != PH_DONE;
        bool ext = act && seed_try<LONG, SM, MS>( L, A.P, c, &S );
        {
            // phase transitions are batched like the refills: run them when enough lanes wait for one (or nobody can step)
            const unsigned long long sm = __ballot( act && !ext );
#if defined( MA_KSW_PROF )
            const unsigned long long tB1 = clock64( );
            if( sm && ( (u32)__popcll( sm ) >= A.slow_batch || __ballot( ext ) == 0 ) )
                pf[ 5 ] += 1ull << 32, pf[ 0 ] -= tB1; // slow trips in the high word; slow cycles: + tC below
#endif
            if( sm && ( (u32)__popcll( sm ) >= A.slow_batch || __ballot( ext ) == 0 ) )
                if( act && !ext )
                    ext = seed_prepare<LONG, true, SM, MS>( L, A.P, S, A.X, c );
#if defined( MA_KSW_PROF )
            if( sm && ( (u32)__popcll( sm ) >= A.slow_batch || __ballot( ext ) == 0 ) )
                pf[ 0 ] += clock64( );
#endif // K-mer table: 150 bp 8.1 -> 7.5 ms; reads in HBM with K byte loads per key 145 -> 188 ms (10 kb), hence seed_jump's block loads
        }
#if defined( MA_KSW_PROF )
        const unsigned long long tC = clock64( );
        pf[ 4 ] += __popcll( __ballot( ext ) );
#endif
        if( ext )
        {
            i64 ok[ 3 ];
            u32 nb;
            if( LONG )
                seed_prefetch<LONG>( L, A.P );
            extend_backward( A.X, L.ik, c, ok, nb );
            L.steps++;
            L.blocks += nb;
            seed_apply<SM, MS>( L, A.P, S, ok );
        }
#if defined( MA_KSW_PROF )
        const unsigned long long tD = clock64( );
        pf[ 6 ] += tB - tA;
        pf[ 1 ] += tC - tB;
        pf[ 2 ] += tD - tC;
        pf[ 3 ] += 1;
        pf[ 5 ] += dm ? 1 : 0;
#endif
    }
#if defined( MA_KSW_PROF )
    if( wl == 0 )
        for( int i = 0; i < 6; i++ )
            atomicAdd( &g_seed_prof[ i ], pf[ i ] );
#endif
    atomicAdd( &A.ctr[ CTR_STEPS ], (unsigned long long)steps );
    atomicAdd( &A.ctr[ CTR_BLOCKS ], (unsigned long long)blocks );
}
// Two register budgets: short reads (staged in LDS) run best without spills at 3 waves per SIMD (137 VGPRs: 8.4 vs 9.2 ms
// per 1 M x 150 bp reads), long reads want the fourth wave more than the 16 spilled dwords hurt (200 k x 10 kb: 155 vs 180 ms).
template <bool SM> __global__ void __launch_bounds__( 256 ) k_seed( SeedKernelArgs A )
{
    seed_kernel_body<false, SM>( A );
}
template <bool SM> __global__ void __launch_bounds__( 256 ) __attribute__( ( amdgpu_waves_per_eu( 4 ) ) ) k_seed_long( SeedKernelArgs A )
{
    seed_kernel_body<true, SM>( A );
}

__device__ __forceinline__ u64 wave_sum_u64( u64 v );
// ---- MEMs seeding (binarySeeding.h:460-537): every start position of every read is independent, so one lane per base.
// Pass 1 counts the segments of each position, a scan lays them out in (read, position) order -- the order the reference
// pushes them in -- pass 2 writes them, k_mems_finish derives the per-read ranges and applies execute()'s drop rule.
struct MemsArgs
{
    IndexView X;
    SeedParams P;
    const uint8_t* reads;
    const u64* roff;
    u32 n_reads;
    u64 n_bases;
    u64* cnt; // pass 1: out, n_bases + 1
    const u64* off; // pass 2: in
    ma_segment* pool;
    u32* pool_read;
    unsigned long long* ctr;
};
struct MemsCount
{
    u64 n = 0;
    MA_HD void emit( u32, u32, i64, i64 )
    {
        n++;
    }
};
struct MemsFill
{
    ma_segment* out;
    u32* out_read;
    u32 read;
    u64 n = 0;
    MA_HD void emit( u32 qs, u32 qsz, i64 sa, i64 san )
    {
        ma_segment s;
        s.q_start = qs, s.q_size = qsz, s.sa_start = sa, s.sa_start_rc = -1, s.sa_size = san;
        out[ n ] = s;
        out_read[ n ] = read;
        n++;
    }
};
template <bool FILL> __global__ void __launch_bounds__( 256 ) k_mems( MemsArgs A )
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 steps = 0, blocks = 0;
    if( t < A.n_bases )
    {
        const u64 b = A.roff[ 0 ] + t;
        // the read of base b: the last r with roff[r] <= b
        u32 lo = 0, hi = A.n_reads;
        while( hi - lo > 1 )
        {
            const u32 mid = ( lo + hi ) / 2;
            if( A.roff[ mid ] <= b )
                lo = mid;
            else
                hi = mid;
        }
        const u64 r0 = A.roff[ lo ];
        const u32 qlen = (u32)( A.roff[ lo + 1 ] - r0 ), i = (u32)( b - r0 );
        if( FILL )
        {
            MemsFill sink{ A.pool + A.off[ t ], A.pool_read + A.off[ t ], lo };
            if( A.off[ t + 1 ] > A.off[ t ] )
                mems_from( A.X, A.P, A.reads + r0, qlen, i, sink, steps, blocks );
        }
        else
        {
            MemsCount sink;
            mems_from( A.X, A.P, A.reads + r0, qlen, i, sink, steps, blocks );
            A.cnt[ t ] = sink.n;
        }
    }
    if( !FILL )
    {
        steps = wave_sum_u64( steps );
        blocks = wave_sum_u64( blocks );
        if( ( threadIdx.x & 63 ) == 0 && steps )
        {
            atomicAdd( &A.ctr[ CTR_STEPS ], (unsigned long long)steps );
            atomicAdd( &A.ctr[ CTR_BLOCKS ], (unsigned long long)blocks );
        }
    }
}
// per read: its segment range; BinarySeeding::execute's drop rule (binarySeeding.cpp:172-175, numSeedsLarger segment.h:278-289):
// a dropped read keeps no segment (its pool entries are blanked so that they yield no seeds)
__global__ void k_mems_finish( IndexView X, SeedParams P, const u64* roff, u32 n_reads, const u64* off, ma_segment* pool, u64* seg_off,
                               u32* seg_cnt )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if( r >= n_reads )
        return;
    const u64 b = off[ roff[ r ] - roff[ 0 ] ], e = off[ roff[ r + 1 ] - roff[ 0 ] ];
    u32 n = (u32)( e - b );
    if( !P.disable_heuristics && P.min_seed_size_drop != 0 )
    {
        u64 sum = 0;
        for( u64 k = b; k < e; k++ )
            sum += (u64)pool[ k ].q_size / (u64)P.min_seed_size_drop;
        if( (double)sum < P.rel_min_seed_size_amount * (double)( roff[ r + 1 ] - roff[ r ] ) && P.genome_size_disable < X.n )
        {
            for( u64 k = b; k < e; k++ )
                pool[ k ].q_size = 0, pool[ k ].sa_size = 0;
            n = 0;
        }
    }
    seg_off[ r ] = b;
    seg_cnt[ r ] = n;
}

// ---- task-parallel maxSpan seeding for long reads --------------------------------------------------------------
// procesInterval (binarySeeding.cpp:32-84) is a binary recursion: the centre of an area is extended, then the part left
// of the covered interval and the part right of it are processed independently.  A read-per-lane walk leaves a 50 kb
// read on ONE lane (20 k reads = 1.2 wavefronts per CU); here every AREA is a task.  The tree is walked level by level
// (the centre is the middle of its area, so an area halves from level to level: depth <= log2(read length) + 1); the
// lanes of a level's launch pull tasks from the level's array, extend, append their 0..2 segments to the pool with the
// task's PRE-ORDER key -- node before its left subtree before its right subtree, two bits per level: exactly the order in
// which the recursion pushes segments -- and append the child areas to the next level's array.  A stable sort by
// (read, key) then restores the reference's segment order.
struct SeedTask
{
    u32 read, aS, aN, depth;
    u64 key; // pre-order path: digit 1 = left, 2 = right, 2 bits per level from bit 38 downwards
};
#define MA_TASK_KEY_BITS 40
struct TaskKernelArgs
{
    IndexView X;
    SeedParams P;
    const uint8_t* reads;
    const u64* roff;
    const SeedTask* in;
    const unsigned long long* nIn; // device: tasks of this level
    SeedTask* out;
    unsigned long long* nOut; // device: tasks of the next level (bump pointer)
    u64 task_cap;
    ma_segment* pool;
    u64* pool_key; // read << MA_TASK_KEY_BITS | path
    u64 pool_cap;
    unsigned long long* ctr;
    u32 slow_batch;
};
__global__ void k_task_roots( const u64* roff, u32 n_reads, SeedTask* out, unsigned long long* nOut )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if( r == 0 )
        *nOut = n_reads;
    if( r >= n_reads )
        return;
    SeedTask t;
    t.read = r, t.aS = 0, t.aN = (u32)( roff[ r + 1 ] - roff[ r ] ), t.depth = 0, t.key = 0;
    out[ r ] = t;
}
__global__ void __launch_bounds__( 256 ) k_seed_tasks( TaskKernelArgs A )
{
    const u32 wl = threadIdx.x & 63;
    // a level that overflowed the task array bumped *nOut past task_cap without writing those tasks: the levels queued
    // behind it must neither run on the unwritten slots nor read past the array (the host falls back to k_seed)
    if( A.ctr[ CTR_ERR ] & MA_ERR_STACK_OVERFLOW )
        return;
    const u64 nIn = *A.nIn < A.task_cap ? *A.nIn : A.task_cap;
    SeedLane L;
    L.phase = PH_DONE;
    L.err = 0;
    ma_segment mine[ 2 ]; // a centre yields at most two segments (maxSpan)
    SeedScratch S;
    S.stage = mine;
    S.seg_cap = 2;
    S.smem_a = S.smem_b = nullptr;
    S.smem_cap = 0;
    S.stack = nullptr;
    S.drop_div = 0;
    SeedTask T;
    T.read = 0xffffffffu;
    bool alive = true;
    u64 steps = 0, blocks = 0;
    u64 qCur = 0, qEnd = 0; // this wave's slice of the level's task array
    while( true )
    {
        const bool done = alive && L.phase == PH_DONE;
        const unsigned long long dm = __ballot( done ), am = __ballot( alive );
        if( dm && ( __popcll( dm ) >= 8 || dm == am ) )
        {
            // ---- finished tasks: segments to the pool, child areas to the next level (one atomic per wave and array)
            const bool flush = done && T.read != 0xffffffffu;
            const u32 ns = flush ? ( L.nseg < 2 ? L.nseg : 2 ) : 0;
            const u32 nc = flush ? ( L.childN[ 0 ] ? 1 : 0 ) + ( L.childN[ 1 ] ? 1 : 0 ) : 0;
            u32 incS = ns, incC = nc;
            for( int d = 1; d < 64; d <<= 1 )
            {
                const u32 o = (u32)__shfl_up( (int)incS, d, 64 ), o2 = (u32)__shfl_up( (int)incC, d, 64 );
                if( wl >= (u32)d )
                    incS += o, incC += o2;
            }
            const u32 totS = (u32)__shfl( (int)incS, 63, 64 ), totC = (u32)__shfl( (int)incC, 63, 64 );
            unsigned long long baseS = 0, baseC = 0;
            if( wl == 0 )
            {
                if( totS )
                    baseS = atomicAdd( &A.ctr[ CTR_SEG_USED ], (unsigned long long)totS );
                if( totC )
                    baseC = atomicAdd( A.nOut, (unsigned long long)totC );
            }
            baseS = ( (u64)(u32)__shfl( (int)( baseS >> 32 ), 0, 64 ) << 32 ) | (u32)__shfl( (int)(u32)baseS, 0, 64 );
            baseC = ( (u64)(u32)__shfl( (int)( baseC >> 32 ), 0, 64 ) << 32 ) | (u32)__shfl( (int)(u32)baseC, 0, 64 );
            if( flush )
            {
                const u64 so = baseS + incS - ns, co = baseC + incC - nc;
                if( so + ns <= A.pool_cap )
                    for( u32 k = 0; k < ns; k++ )
                    {
                        A.pool[ so + k ] = mine[ k ];
                        A.pool_key[ so + k ] = ( (u64)T.read << MA_TASK_KEY_BITS ) | T.key;
                    }
                else
                    L.err |= MA_ERR_SEG_OVERFLOW;
                if( co + nc <= A.task_cap && T.depth + 1 < MA_TASK_KEY_BITS / 2 )
                {
                    u32 w = 0;
                    for( int side = 0; side < 2; side++ )
                        if( L.childN[ side ] )
                        {
                            SeedTask c;
                            c.read = T.read, c.aS = L.childS[ side ], c.aN = L.childN[ side ], c.depth = T.depth + 1;
                            c.key = T.key | ( (u64)( side + 1 ) << ( MA_TASK_KEY_BITS - 2 * ( T.depth + 1 ) ) );
                            A.out[ co + w++ ] = c;
                        }
                }
                else if( nc )
                    L.err |= MA_ERR_STACK_OVERFLOW;
                if( L.err )
                    atomicOr( (unsigned long long*)&A.ctr[ CTR_ERR ], (unsigned long long)L.err );
                steps += L.steps;
                blocks += L.blocks;
                T.read = 0xffffffffu;
            }
            // ---- next tasks
            if( qCur == qEnd )
            {
                unsigned long long base = 0;
                if( wl == 0 )
                    base = atomicAdd( &A.ctr[ CTR_NEXT_READ ], 256ull );
                base = ( (u64)(u32)__shfl( (int)( base >> 32 ), 0, 64 ) << 32 ) | (u32)__shfl( (int)(u32)base, 0, 64 );
                qCur = base < nIn ? base : nIn;
                qEnd = base + 256 < nIn ? base + 256 : nIn;
            }
            const u64 avail = qEnd - qCur;
            const u64 rank = (u64)__popcll( dm & ( ( 1ull << wl ) - 1 ) );
            if( done )
            {
                if( rank < avail )
                {
                    T = A.in[ qCur + rank ];
                    const u64 r0 = A.roff[ T.read ];
                    seed_begin_area( L, A.reads + r0, (u32)( A.roff[ T.read + 1 ] - r0 ), T.aS, T.aN );
                }
                else if( qEnd == nIn )
                    alive = false;
            }
            const u64 want = (u64)__popcll( dm );
            qCur += want < avail ? want : avail;
        }
        if( __ballot( alive ) == 0 )
            break;
        u32 c = 0;
        const bool act = alive && L.phase != PH_DONE;
        bool ext = act && seed_try<true, false>( L, A.P, c );
        {
            const unsigned long long sm = __ballot( act && !ext );
            if( sm && ( (u32)__popcll( sm ) >= A.slow_batch || __ballot( ext ) == 0 ) )
                if( act && !ext )
                    ext = seed_prepare<true, true, false>( L, A.P, S, A.X, c );
        }
        if( ext )
        {
            i64 ok[ 3 ];
            u32 nb;
            seed_prefetch<true>( L, A.P );
            extend_backward( A.X, L.ik, c, ok, nb );
            L.steps++;
            L.blocks += nb;
            seed_apply<false>( L, A.P, S, ok );
        }
    }
    steps = wave_sum_u64( steps );
    blocks = wave_sum_u64( blocks );
    if( wl == 0 && steps )
    {
        atomicAdd( &A.ctr[ CTR_STEPS ], (unsigned long long)steps );
        atomicAdd( &A.ctr[ CTR_BLOCKS ], (unsigned long long)blocks );
    }
}
// segments into (read, pre-order) order; read id per segment
__global__ void k_task_permute( const ma_segment* in, const u64* sorted_key, const u32* perm, u64 n, ma_segment* out, u32* out_read )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n )
        return;
    out[ i ] = in[ perm[ i ] ];
    out_read[ i ] = (u32)( sorted_key[ i ] >> MA_TASK_KEY_BITS );
}
__global__ void k_iota32( u32* p, u64 n )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i < n )
        p[ i ] = (u32)i;
}
// per read: first segment and count (the read ids are sorted), then BinarySeeding::execute's drop rule
__global__ void k_task_ranges( const u32* seg_read, u64 n, u64* seg_off, u32* seg_cnt )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n )
        return;
    const u32 r = seg_read[ i ];
    if( i == 0 || seg_read[ i - 1 ] != r )
        seg_off[ r ] = i;
    if( i + 1 == n || seg_read[ i + 1 ] != r )
        seg_cnt[ r ] = (u32)( i + 1 ); // end; turned into a count by k_task_finish
}
__global__ void k_task_finish( IndexView X, SeedParams P, const u64* roff, u32 n_reads, ma_segment* pool, u64* seg_off, u32* seg_cnt )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if( r >= n_reads )
        return;
    if( seg_cnt[ r ] == 0 )
    {
        seg_off[ r ] = 0;
        return;
    }
    const u64 b = seg_off[ r ], e = seg_cnt[ r ];
    u32 n = (u32)( e - b );
    if( !P.disable_heuristics && P.min_seed_size_drop != 0 )
    {
        u64 sum = 0;
        for( u64 k = b; k < e; k++ )
            sum += (u64)pool[ k ].q_size / (u64)P.min_seed_size_drop;
        if( (double)sum < P.rel_min_seed_size_amount * (double)( roff[ r + 1 ] - roff[ r ] ) && P.genome_size_disable < X.n )
        {
            for( u64 k = b; k < e; k++ )
                pool[ k ].q_size = 0, pool[ k ].sa_size = 0;
            n = 0;
        }
    }
    seg_cnt[ r ] = n;
}

// per pooled segment: number of seeds it yields (segment.h:316-349 filters)
__global__ void k_seg_seed_counts( const ma_segment* pool, u64 n, u32 min_len, u32 max_amb, u64* cnt )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n )
        return;
    const ma_segment s = pool[ i ];
    u64 c = (u64)s.sa_size;
    if( (u64)s.q_size < (u64)min_len )
        c = 0;
    if( s.sa_size > (i64)max_amb && max_amb != 0 )
        c = 0; // bSkip == true (segment.h:365)
    cnt[ i ] = c;
}

// per read: seed range = ranges of its segments (contiguous in the pool)
__global__ void k_read_seed_ranges( const u64* seg_off, const u32* seg_cnt, const u64* seg_seed_off, u64 n_pool,
                                    u64 total_seeds, u32 n_reads, u64* seed_off, u32* seed_cnt )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if( r >= n_reads )
        return;
    const u64 b = seg_off[ r ], e = b + seg_cnt[ r ];
    const u64 sb = seg_cnt[ r ] ? seg_seed_off[ b ] : 0;
    const u64 se = seg_cnt[ r ] ? ( e < n_pool ? seg_seed_off[ e ] : total_seeds ) : 0;
    seed_off[ r ] = sb;
    seed_cnt[ r ] = (u32)( se - sb );
}

// ---- seed extraction (Segment::forEachSeed segment.h:89-113, setDeltaOfSeed stripOfConsideration.h:97-112 in
// rectangular mode) in three passes:
//  k_seed_rows   one lane per pooled segment: SA row and segment index of each of its seeds
//  k_lf_walk     persistent lanes, ONE LF step (one random 64-B block) per lane and trip, a finished lane takes
//                the next seed from a wave-aggregated queue: no divergence over the (unbounded, mean 16) steps a
//                row needs until it hits a sampled row
//  k_seed_final  one lane per seed: sampled SA value, strand, contig, delta
__global__ void k_seed_rows( const ma_segment* pool, const u64* seg_seed_off, u64 n_pool, i64* row, u32* seg_of )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n_pool )
        return;
    const u64 o = seg_seed_off[ i ], cnt = seg_seed_off[ i + 1 ] - o;
    if( cnt == 0 )
        return;
    const i64 r0 = pool[ i ].sa_start;
    for( u64 t = 0; t < cnt; t++ )
    {
        row[ o + t ] = r0 + (i64)t;
        seg_of[ o + t ] = (u32)i;
    }
}

__global__ void __launch_bounds__( 256 ) k_lf_walk( IndexView X, i64* row /* in: SA row, out: sampled row reached */,
                                                   u32* nsteps, u64 total, unsigned long long* next )
{
    const u32 wl = threadIdx.x & 63;
    const i64 saMask = ( (i64)1 << X.sa_shift ) - 1;
    bool alive = true, have = false;
    i64 k = 0;
    u64 j = 0;
    u32 st = 0;
    u64 qCur = 0, qEnd = 0; // this wave's slice of the seed queue: one device atomic per 256 seeds
    while( true )
    {
        if( have && ( k & saMask ) == 0 )
        {
            row[ j ] = k;
            nsteps[ j ] = st; // not bounded by the sampling interval: the walk ends when it HITS a sampled row
            have = false;
        }
        const bool need = alive && !have;
        const unsigned long long dm = __ballot( need ), am = __ballot( alive );
        if( dm && ( __popcll( dm ) >= 8 || dm == am ) )
        {
            if( qCur == qEnd )
            {
                unsigned long long base = 0;
                if( wl == 0 )
                    base = atomicAdd( next, 256ull );
                base = ( (u64)(u32)__shfl( (int)( base >> 32 ), 0, 64 ) << 32 ) | (u32)__shfl( (int)(u32)base, 0, 64 );
                qCur = base < total ? base : total;
                qEnd = base + 256 < total ? base + 256 : total;
            }
            const u64 avail = qEnd - qCur;
            const u64 rank = (u64)__popcll( dm & ( ( 1ull << wl ) - 1 ) );
            if( need )
            {
                if( rank < avail )
                {
                    j = qCur + rank;
                    k = row[ j ];
                    st = 0;
                    have = true;
                }
                else if( qEnd == total )
                    alive = false; // the queue is exhausted
            }
            const u64 want = (u64)__popcll( dm );
            qCur += want < avail ? want : avail;
        }
        if( __ballot( alive ) == 0 )
            break;
        if( have && ( k & saMask ) )
        {
            k = inv_psi( X, k );
            st++;
        }
    }
}

__global__ void k_seed_final( IndexView X, const ma_segment* pool, const u32* pool_read, const u64* seg_seed_off,
                              const i64* row, const u32* nsteps, const u32* seg_of, u64 total_seeds, const u64* roff,
                              ma_seed* seeds, unsigned long long* ctr )
{
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 steps = 0;
    if( j < total_seeds )
    {
        const u64 i = seg_of[ j ];
        const ma_segment s = pool[ i ];
        steps = nsteps[ j ];
        u64 r = (u64)( (i64)steps + sa_sample( X, row[ j ] ) ); // bwt_sa (fMIndex.h:788-814)
        const bool fwd = r < X.n / 2;
        if( !fwd )
            r = X.n - r - 1;
        const u32 rd = pool_read[ i ];
        const u64 qlen = roff[ rd + 1 ] - roff[ rd ];
        ma_seed sd;
        sd.q_start = s.q_start;
        sd.len = s.q_size + 1;
        sd.r_start = (i64)r;
        sd.ambiguity = (u32)s.sa_size;
        sd.on_forward = fwd ? 1 : 0;
        u64 delta = r + ( qlen - (u64)s.q_start );
        delta += ( qlen + 1 ) * (u64)seq_id_for_position( X, r );
        sd.delta = (i64)delta;
        seeds[ j ] = sd;
    }
    // wave-aggregated counters
    u64 st = steps;
    for( int m = 32; m >= 1; m >>= 1 )
        st += __shfl_xor( st, m, 64 );
    if( ( threadIdx.x & 63 ) == 0 && st )
        atomicAdd( &ctr[ CTR_LF_STEPS ], (unsigned long long)st );
}

struct ChainKernelArgs
{
    IndexView X;
    ChainParams P;
    u32 n_reads;
    const u64* roff;
    const u64* seed_off;
    const u32* seed_cnt;
    const ma_seed* seeds;
    // scratch carved by seed offset
    ma_seed* work;
    SoCEntry* maxima;
    RefMinMax* mm;
    ma_seed* setA;
    ma_seed* setB;
    ma_seed* outA;
    Shadow* sh1;
    Shadow* sh2;
    double* vX;
    double* vY;
    double* med;
    i32* inl;
    i32* best;
    // output
    ma_seed* hpool; // shared overflow pool (atomic bump pointer CTR_HSEED_USED)
    u64 hpool_cap;
    ma_seed* hlocal; // private regions: read r owns [3 * seed_off[r], + 3 * seed_cnt[r])
    HSet* sets; // n_reads * set_cap
    u32 set_cap;
    u32* nsets; // per read
    unsigned long long* ctr;
    u32 lanes; // reads per wavefront (lanes_per_wave)
    // optional: the SoC queues were swept elsewhere (ma_batch_set_soc_heap); carved by seed offset like the scratch
    const ma_soc* queue;
    const u32* queue_cnt;
    // optional (long reads): the sweep ran as separate kernels around the wave-cooperative sorts (k_sort_seeds_wave,
    // k_soc_windows): strips per read, and which of a read's two sorts the wave kernel did (bit 0 delta, bit 1 reference)
    const u32* pre_nmx;
    const u32* pre_sorted;
};

__global__ void __launch_bounds__( 64 ) __attribute__( ( amdgpu_waves_per_eu( 4, 4 ) ) ) k_chain( ChainKernelArgs A )
{
    const u32 r = blockIdx.x * A.lanes + threadIdx.x;
    if( threadIdx.x >= A.lanes || r >= A.n_reads )
        return;
    const u64 off = A.seed_off[ r ];
    const u32 n = A.seed_cnt[ r ];
    ChainScratch C;
    C.work = A.work + off;
    C.maxima = A.maxima + off;
    C.mm = A.mm + off;
    C.setA = A.setA + off;
    C.setB = A.setB + off;
    C.outA = A.outA + off;
    C.sh1 = A.sh1 + off;
    C.sh2 = A.sh2 + off;
    C.vX = A.vX + 3 * off;
    C.vY = A.vY + 3 * off;
    C.med = A.med + 6 * off;
    C.inl = A.inl + 3 * off;
    C.best = A.best + 3 * off;
    if( A.pre_nmx == nullptr )
        for( u32 i = 0; i < n; i++ )
            C.work[ i ] = A.seeds[ off + i ];
    ChainOut O;
    O.pool = A.hpool;
    O.pool_cap = A.hpool_cap;
    O.pool_used = &A.ctr[ CTR_HSEED_USED ];
    O.sets = A.sets + (u64)r * A.set_cap;
    O.set_cap = A.set_cap;
    O.local = A.hlocal + 3 * off; // seed ranges of different reads are disjoint (but not ordered by read)
    O.local_cap = 3 * n;
    u32 err = 0;
    const u32 qlen = (u32)( A.roff[ r + 1 ] - A.roff[ r ] );
    const u32 ns = chain_read( A.X, A.P, C, n, qlen, O, err, A.queue ? A.queue + off : nullptr, A.queue ? A.queue_cnt[ r ] : 0,
                               A.pre_nmx != nullptr, A.pre_nmx ? A.pre_nmx[ r ] : 0, A.pre_sorted ? ( A.pre_sorted[ r ] & 2u ) != 0 : false );
    A.nsets[ r ] = ns < A.set_cap ? ns : A.set_cap;
    if( err )
        atomicOr( (unsigned long long*)&A.ctr[ CTR_ERR ], (unsigned long long)err );
}

// ---- long reads: the two big sorts of the sweep as wave-cooperative kernels (wave_sort.h), the window sweep between them
struct PackedKeyLess
{
    __device__ bool operator( )( u64 a, u64 b ) const
    {
        return ( a >> 20 ) < ( b >> 20 );
    }
};
// reads with fewer seeds are sorted by their lane in k_soc_windows / k_chain as before: a wavefront per read pays off when the
// sort has many large ranges to partition (10 kb reads, ~250 seeds: 28 ms of wave sorts vs 17 ms inside the lane kernels)
#define MA_WSORT_MIN 768u
#define MA_WSORT_SMALL 2688u // reads with up to this many seeds: 37 KB of LDS per wavefront
#define MA_WSORT_LARGE 8192u // up to this many: 100 KB; more -> the lane-serial sort of chain.h
// One wavefront per read.  mode 0: work = seeds sorted by delta (reads outside [nMin, nMax] of this launch are left alone,
// reads it owns but cannot sort are copied unsorted); mode 1: work re-sorted by reference position in place (via tmp).
__global__ void __launch_bounds__( 64 ) k_sort_seeds_wave( u32 n_reads, const u64* seed_off, const u32* seed_cnt, const ma_seed* seeds,
                                                          ma_seed* work, ma_seed* tmp, u32* sorted, int mode, u32 nMin, u32 nMax,
                                                          u32 nSortMin, u32 nSortMax )
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t lds[];
    const u32 r = blockIdx.x;
    const int lane = threadIdx.x & 63;
    if( r >= n_reads )
        return;
    const u32 n = seed_cnt[ r ];
    if( n < nMin || n > nMax )
        return;
    const u64 off = seed_off[ r ];
    bool doSort = n >= nSortMin && n <= nSortMax;
    const ma_seed* src = mode == 0 ? seeds + off : work + off;
    ws::Scratch S = ws::carve( lds, doSort ? n : 1 );
    if( doSort )
    {
        bool wide = false;
        for( u32 i = lane; i < n; i += 64 )
        {
            const u64 key = mode == 0 ? (u64)src[ i ].delta : (u64)src[ i ].r_start;
            wide = wide || ( key >> 44 ) != 0;
            S.a[ i ] = ( key << 20 ) | (u64)i;
        }
        if( __ballot( wide ) != 0 )
            doSort = false; // does not pack (never for genomes below 2^44 positions)
        __syncthreads( );
    }
    if( !doSort )
    {
        if( mode == 0 )
            for( u32 i = lane; i < n; i += 64 )
                work[ off + i ] = src[ i ];
        return;
    }
    ws::wave_std_sort( S, (i32)n, PackedKeyLess( ) );
    if( mode == 0 )
        for( u32 i = lane; i < n; i += 64 )
            work[ off + i ] = src[ (u32)( S.a[ i ] & 0xfffffu ) ];
    else
    {
        for( u32 i = lane; i < n; i += 64 )
            tmp[ off + i ] = src[ (u32)( S.a[ i ] & 0xfffffu ) ];
        for( u32 i = lane; i < n; i += 64 ) // every lane copies back what it wrote itself
            work[ off + i ] = tmp[ off + i ];
    }
    if( lane == 0 )
        sorted[ r ] |= 1u << mode;
}
// the sweep between the two sorts, one read per lane (thin waves like k_chain)
__global__ void __launch_bounds__( 64 ) k_soc_windows( IndexView X, ChainParams P, u32 n_reads, u32 lanes, const u64* roff, const u64* seed_off,
                                                      const u32* seed_cnt, ma_seed* work, SoCEntry* maxima, RefMinMax* mm, ma_seed* tmp,
                                                      const u32* sorted, u32* pre_nmx )
{
    const u32 r = blockIdx.x * lanes + threadIdx.x;
    if( threadIdx.x >= lanes || r >= n_reads )
        return;
    const u64 off = seed_off[ r ];
    const bool byDelta = ( sorted[ r ] & 1u ) != 0;
    // (tmp is the keyed sort's scratch, free when the wave-cooperative kernel did the sort: 40 n bytes for the 12 (n + 1) of the prefix
    // sums.  Reads sorted in here -- 10 kb: ~240 seeds -- gain nothing: building the sums costs what they save, 55.3 vs 57.7 ms)
    pre_nmx[ r ] = soc_windows( X, P, work + off, seed_cnt[ r ], (u32)( roff[ r + 1 ] - roff[ r ] ), maxima + off, mm + off, tmp + off,
                                byDelta, byDelta && seed_cnt[ r ] >= 2 ? (u64*)( tmp + off ) : nullptr );
}

// the SoC queue of every read in pop order (ma_batch_get_socs); scratch and output carved by the read's seed offset
__global__ void __launch_bounds__( 64 ) k_soc_dump( IndexView X, ChainParams P, u32 n_reads, const u64* roff, const u64* seed_off,
                                                   const u32* seed_cnt, const ma_seed* seeds, ma_seed* work, SoCEntry* maxima,
                                                   RefMinMax* mm, ma_soc* socs, u32* nsocs, int heap_layout )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if( r >= n_reads )
        return;
    const u64 off = seed_off[ r ];
    const u32 n = seed_cnt[ r ];
    for( u32 i = 0; i < n; i++ )
        work[ off + i ] = seeds[ off + i ];
    nsocs[ r ] = soc_dump_read( X, P, work + off, n, (u32)( roff[ r + 1 ] - roff[ r ] ), maxima + off, mm + off, socs + off, heap_layout != 0 );
}

// harmonized seeds of a read (sum of its sets' sizes), input of the scan that lays out the dense pool
__global__ void k_hseed_counts( const HSet* sets, u32 set_cap, const u32* nsets, u32 n_reads, u64* cnt )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if( r >= n_reads )
        return;
    u64 c = 0;
    for( u32 k = 0; k < nsets[ r ]; k++ )
        c += sets[ (u64)r * set_cap + k ].cnt;
    cnt[ r ] = c;
}

// flatten the per-read set tables into CSR order (hset_off from an exclusive scan of nsets) and compact the seeds of
// the sets (private regions / overflow pool) into one dense pool in read order (hseed_off from a scan of the counts)
__global__ void k_hset_flatten( const HSet* sets, u32 set_cap, const u32* nsets, const u64* hset_off, u32 n_reads,
                                const u64* hseed_off, const u64* seed_off, const ma_seed* hlocal, const ma_seed* hovf,
                                ma_seed* dense, HSet* flat, u32* flat_read )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if( r >= n_reads )
        return;
    const u64 o = hset_off[ r ];
    u64 d = hseed_off[ r ];
    const ma_seed* mine = hlocal + 3 * seed_off[ r ];
    for( u32 k = 0; k < nsets[ r ]; k++ )
    {
        HSet h = sets[ (u64)r * set_cap + k ];
        const ma_seed* src = ( h.off & MA_HSET_LOCAL ) ? mine + ( h.off & ~MA_HSET_LOCAL ) : hovf + h.off;
        for( u32 i = 0; i < h.cnt; i++ )
            dense[ d + i ] = src[ i ];
        h.off = d;
        d += h.cnt;
        flat[ o + k ] = h;
        flat_read[ o + k ] = r;
    }
}

struct SetInfo // per harmonized set, filled by the enumeration pass
{
    u64 win_begin, win_end;
    u32 valid;
    u32 n_jobs;
};

struct EnumSink
{
    static const bool STITCH = false;
    DpJob* jobs; // slots of this set
    u32 n;
    u32 cap;
    u32 slot0; // global index of jobs[0]
    u64 win_begin, read_off;
    // sizing of the ksw launches, accumulated per lane and reduced once per wave by the kernel
    u64 mx_state = 0, mx_h = 0, mx_p = 0, mx_cig = 0, mx_qlen = 0, n_jobs = 0, seq_bytes = 0;
    MA_HD void job( u32 qf, u32 qt, u32 rf, u32 rt, i32 w, i32 zdrop, i32 flag, u32 rev )
    {
        if( n < cap )
        {
            DpJob j;
            j.win_begin = win_begin;
            j.read_off = read_off;
            j.q_from = qf, j.q_to = qt, j.r_from = rf, j.r_to = rt;
            j.w = w, j.zdrop = zdrop, j.flag = flag, j.rev = rev;
            jobs[ n ] = j;
#if defined( __HIP_DEVICE_COMPILE__ )
            const i32 ql = (i32)( qt - qf ), tl = (i32)( rt - rf );
            const u64 L = (u64)( ( tl + 15 ) / 16 ) * 16;
            const u64 p = (u64)( (i64)ql + tl - 1 ) * (u64)( ksw_ncol( ql, tl, w ) * 16 ) + 16;
            mx_state = mmax( mx_state, ksw_state_bytes( ql, tl ) );
            mx_h = mmax( mx_h, L * 4 );
            mx_p = mmax( mx_p, p );
            mx_cig = mmax( mx_cig, (u64)ql + tl + 2 );
            mx_qlen = mmax( mx_qlen, (u64)ql );
            n_jobs++;
            seq_bytes += (u64)( ql + tl );
#endif
        }
        n++;
    }
    MA_HD KswResult next( )
    {
        return KswResult{ -1, -1, nullptr, 0 };
    }
};

struct DpKernelArgs
{
    IndexView X;
    NwParams P;
    u32 n_sets;
    const HSet* sets;
    const u32* set_read;
    const ma_seed* hpool;
    const uint8_t* reads;
    const u64* roff;
    DpJob* jobs; // 2 slots per pooled harmonized seed: slots of set s start at 2*sets[s].off
    SetInfo* info;
    unsigned long long* ctr;
    u32* lists;
    u64 list_stride;
    KswScoring SC;
    u32 lanes; // sets per wavefront (lanes_per_wave)
    u32 wave_split; // long reads: sets that span >= 1024 query bases go to k_stitch_wave
};

#if defined( __HIPCC__ )
__device__ __forceinline__ u64 wave_max_u64( u64 v )
{
    for( int m = 32; m; m >>= 1 )
    {
        const u64 o = ( (u64)(u32)__shfl_xor( (int)( v >> 32 ), m, 64 ) << 32 ) | (u32)__shfl_xor( (int)(u32)v, m, 64 );
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ u64 wave_sum_u64( u64 v )
{
    for( int m = 32; m; m >>= 1 )
        v += ( (u64)(u32)__shfl_xor( (int)( v >> 32 ), m, 64 ) << 32 ) | (u32)__shfl_xor( (int)(u32)v, m, 64 );
    return v;
}
#endif

__device__ void dp_enum_one( const DpKernelArgs& A, u32 s, EnumSink& sink )
{
    const HSet hs = A.sets[ s ];
    const ma_seed* S = A.hpool + hs.off;
    const u32 rd = A.set_read[ s ];
    const u64 qlen = A.roff[ rd + 1 ] - A.roff[ rd ];
    SetInfo I;
    const NwWindow W = nw_window( A.X, A.P, S, hs.cnt );
    I.win_begin = W.begin_ref;
    I.win_end = W.end_ref;
    I.valid = W.valid ? 1 : 0;
    I.n_jobs = 0;
    if( W.valid )
    {
        sink.jobs = A.jobs + 2 * hs.off;
        sink.slot0 = (u32)( 2 * hs.off );
        sink.n = 0;
        sink.cap = 2 * hs.cnt;
        sink.win_begin = W.begin_ref;
        sink.read_off = A.roff[ rd ];
        NwWalk<EnumSink> walk{ A.X, A.P, sink, A.reads + A.roff[ rd ], W.begin_ref, AlnBuilder{ nullptr, nullptr, nullptr } };
        walk.run( S, hs.cnt, qlen, W );
        I.n_jobs = sink.n < sink.cap ? sink.n : sink.cap;
        if( sink.n > sink.cap )
            atomicOr( (unsigned long long*)&A.ctr[ CTR_ERR ], (unsigned long long)MA_ERR_SCRATCH_OVERFLOW );
    }
    A.info[ s ] = I;
}

__global__ void __launch_bounds__( 64 ) k_dp_enum( DpKernelArgs A )
{
    const u32 s = blockIdx.x * A.lanes + threadIdx.x;
    EnumSink sink;
    sink.n = 0;
    sink.cap = 0;
    sink.slot0 = 0;
    if( threadIdx.x < A.lanes && s < A.n_sets )
        dp_enum_one( A, s, sink );
    u32 pcl[ KSW_N_CLASSES ], cgl[ KSW_N_CLASSES ], pRedo = 0, cgRedo = 0;
    // append the jobs to the per-class lists: one atomic per wave, class and round instead of one per job
    {
        const u32 mine = sink.n < sink.cap ? sink.n : sink.cap;
        const u32 rounds = (u32)wave_max_u64( mine );
        const int lane = threadIdx.x & 63;
        // scratch per wave of each class's launch: the classes differ by orders of magnitude (a 50 kb end extension
        // needs 27 MB of direction bytes, a gap between two seeds a few KB), and a launch sized for the largest job of
        // the whole batch would leave most of the machine without waves
#pragma unroll
        for( int c = 0; c < KSW_N_CLASSES; c++ )
            pcl[ c ] = cgl[ c ] = 0;
        for( u32 k = 0; k < rounds; k++ )
        {
            int cls = -1;
            u32 pj = 0, cj = 0;
            if( k < mine )
            {
                const DpJob& j = A.jobs[ sink.slot0 + k ];
                const i32 ql = (i32)( j.q_to - j.q_from ), tl = (i32)( j.r_to - j.r_from );
                cls = ksw_job_class_pipe( A.SC, ql, tl, j.w, j.zdrop, j.flag );
                const u64 pk = ksw_p_bytes( ql, tl, j.w );
                pj = (u32)( ( ( cls >= 5 ? ksw_ext_p_bytes( ql, tl, cls - 4 ) : pk ) + 255 ) >> 8 ); // 256-byte units
                cj = (u32)( ql + tl + 2 );
                if( cls >= 5 )
                {
                    pRedo = max( pRedo, (u32)( ( pk + 255 ) >> 8 ) );
                    cgRedo = max( cgRedo, cj );
                }
            }
#pragma unroll
            for( int c = 0; c < KSW_N_CLASSES; c++ )
            {
                if( cls == c )
                {
                    pcl[ c ] = max( pcl[ c ], pj );
                    cgl[ c ] = max( cgl[ c ], cj );
                }
                const unsigned long long m = __ballot( cls == c );
                if( m == 0 )
                    continue;
                const int leader = __ffsll( (long long)m ) - 1;
                unsigned long long base = 0;
                if( lane == leader )
                    base = atomicAdd( &A.ctr[ CTR_CLS0 + c ], (unsigned long long)__popcll( m ) );
                base = ( (u64)(u32)__shfl( (int)( base >> 32 ), leader, 64 ) << 32 ) | (u32)__shfl( (int)(u32)base, leader, 64 );
                if( cls == c )
                    A.lists[ (u64)c * A.list_stride + base + __popcll( m & ( ( 1ull << lane ) - 1 ) ) ] = sink.slot0 + k;
            }
        }
    }
    // one atomic per wave and quantity instead of eight per job
    const u64 st = wave_max_u64( sink.mx_state ), h = wave_max_u64( sink.mx_h ), p = wave_max_u64( sink.mx_p );
    const u64 cg = wave_max_u64( sink.mx_cig ), ql = wave_max_u64( sink.mx_qlen );
    const u64 nj = wave_sum_u64( sink.n_jobs ), sb = wave_sum_u64( sink.seq_bytes );
    u64 pcW[ KSW_N_CLASSES ], cgW[ KSW_N_CLASSES ];
#pragma unroll
    for( int c = 0; c < KSW_N_CLASSES; c++ )
    {
        pcW[ c ] = wave_max_u64( pcl[ c ] );
        cgW[ c ] = wave_max_u64( cgl[ c ] );
    }
    const u64 pRedoW = wave_max_u64( pRedo ), cgRedoW = wave_max_u64( cgRedo );
    if( ( threadIdx.x & 63 ) == 0 && nj )
    {
        atomicMax( &A.ctr[ CTR_MAX_STATE ], (unsigned long long)st );
        atomicMax( &A.ctr[ CTR_MAX_H ], (unsigned long long)h );
        atomicMax( &A.ctr[ CTR_MAX_P ], (unsigned long long)p );
        atomicMax( &A.ctr[ CTR_MAX_CIG ], (unsigned long long)cg );
        atomicMax( &A.ctr[ CTR_MAX_QLEN ], (unsigned long long)ql );
#pragma unroll
        for( int c = 0; c < KSW_N_CLASSES; c++ )
        {
            if( pcW[ c ] )
                atomicMax( &A.ctr[ CTR_MAX_PC0 + c ], (unsigned long long)pcW[ c ] << 8 );
            if( cgW[ c ] )
                atomicMax( &A.ctr[ CTR_MAX_CIGC0 + c ], (unsigned long long)cgW[ c ] );
        }
        if( pRedoW )
        {
            atomicMax( &A.ctr[ CTR_MAX_P_REDO ], (unsigned long long)pRedoW << 8 );
            atomicMax( &A.ctr[ CTR_MAX_CIG_REDO ], (unsigned long long)cgRedoW );
        }
        atomicAdd( &A.ctr[ CTR_N_JOBS ], (unsigned long long)nj );
        atomicAdd( &A.ctr[ CTR_SEQ_BYTES ], (unsigned long long)sb );
    }
}

namespace
{
struct PipeFetch
{
    static const bool EARLY = true; // the stitch pass reads only max_q, max_t and the cigar (ksw_reg.h)
    IndexView X;
    const DpJob* jobs;
    const uint8_t* reads;
    __device__ bool valid( u32 s ) const
    {
        return jobs[ s ].q_to > jobs[ s ].q_from; // slots are zero-filled before enumeration
    }
    __device__ KswJobView view( u32 s ) const
    {
        const DpJob& j = jobs[ s ];
        KswJobView v;
        v.qlen = (i32)( j.q_to - j.q_from );
        v.tlen = (i32)( j.r_to - j.r_from );
        v.w = j.w;
        v.zdrop = j.zdrop;
        v.flag = j.flag;
        return v;
    }
    struct Q
    {
        const uint8_t* q;
        u32 from, to, rev;
        __device__ u32 operator( )( i32 i ) const
        {
            return rev ? q[ to - 1 - (u32)i ] : q[ from + (u32)i ];
        }
    };
    struct T
    {
        IndexView X;
        u64 base;
        u32 from, to, rev;
        __device__ u32 operator( )( i32 i ) const
        {
            return text_base( X, base + ( rev ? to - 1 - (u32)i : from + (u32)i ) );
        }
        // bases of cells i and i + 1 (low / high half) with one address computation; i < to - from; the high half repeats
        // cell i when i + 1 is past the window (the caller masks it).  A DP window never bridges the two strands.
        static const bool CLEAN = true; // codes 0..3 only (2-bit pack): no N to recode
        __device__ u32 pair( i32 i ) const
        {
            // the window lies on one strand, so strand and step direction are wave-uniform: forward position of cell i =
            // fFirst + sgn * i
            const u64 pFirst = base + ( rev ? to - 1 : from );
            const bool comp = pFirst >= X.F;
            const i32 sgn = ( rev != 0 ) != comp ? -1 : 1;
            const u64 fFirst = comp ? X.n - 1 - pFirst : pFirst;
            const u64 f0 = fFirst + (u64)(i64)( sgn * i );
            const u64 f1 = f0 + (u64)(i64)( (u32)i + 1 < to - from ? sgn : 0 );
            u32 b0 = ( (u32)X.pac[ f0 >> 2 ] >> ( ( ~(u32)f0 & 3 ) << 1 ) ) & 3;
            u32 b1 = ( (u32)X.pac[ f1 >> 2 ] >> ( ( ~(u32)f1 & 3 ) << 1 ) ) & 3;
            const u32 flip = comp ? 0x00030003u : 0u; // complement of a 2-bit code = code ^ 3
            return ( b0 | b1 << 16 ) ^ flip;
        }
    };
    __device__ Q qfetch( u32 s ) const
    {
        const DpJob& j = jobs[ s ];
        return Q{ reads + j.read_off, j.q_from, j.q_to, j.rev };
    }
    __device__ T tfetch( u32 s ) const
    {
        const DpJob& j = jobs[ s ];
        return T{ X, j.win_begin, j.r_from, j.r_to, j.rev };
    }
};
} // namespace

// Longest jobs first: a persistent launch whose waves pull jobs from a queue ends when its LAST job ends, and a long job
// taken late is a tail with one busy wave.  The lists of the exact register kernels (long-read batches: 10^4..10^6 jobs of
// 10^3..10^8 cells) are therefore sorted by descending direction-matrix size before the launch (LPT rule).
__global__ void k_job_cost( PipeFetch F, const u32* list, u32 n, u32* key )
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n )
        return;
    const KswJobView J = F.view( list[ i ] );
    const u64 c = ksw_p_bytes( J.qlen, J.tlen, J.w ) >> 6;
    key[ i ] = c > 0xffffffffull ? 0xffffffffu : (u32)c;
}

// ops capacity of a set: |Q| + sum of its jobs' cigar lengths + 8 * seeds + 16 (see nw.h)
__global__ void k_ops_caps( const HSet* sets, const SetInfo* info, const u32* set_read, const u64* roff,
                            const ma_ez* ez, u32 n_sets, u64* caps )
{
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if( s >= n_sets )
        return;
    const HSet hs = sets[ s ];
    const u32 rd = set_read[ s ];
    u64 c = ( roff[ rd + 1 ] - roff[ rd ] ) + 8ull * hs.cnt + 16;
    for( u32 k = 0; k < info[ s ].n_jobs; k++ )
        c += (u64)ez[ 2 * hs.off + k ].n_cigar;
    caps[ s ] = info[ s ].valid ? c : 0;
}

struct StitchSink
{
    static const bool STITCH = true;
    const ma_ez* ez;
    const u64* cig_off;
    const u32* cig_pool;
    u32 k;
    MA_HD void job( u32, u32, u32, u32, i32, i32, i32, u32 )
    {}
    MA_HD KswResult next( )
    {
        KswResult R;
        R.max_q = ez[ k ].max_q;
        R.max_t = ez[ k ].max_t;
        R.n_cigar = (u32)ez[ k ].n_cigar;
        R.cigar = cig_pool + cig_off[ k ];
        k++;
        return R;
    }
};

struct StitchKernelArgs
{
    IndexView X;
    NwParams P;
    u32 n_sets;
    const HSet* sets;
    const u32* set_read;
    const SetInfo* info;
    const ma_seed* hpool;
    const uint8_t* reads;
    const u64* roff;
    const ma_ez* ez;
    const u64* cig_off;
    const u32* cig_pool;
    const u64* ops_off; // exclusive scan of caps
    const u64* ops_cap;
    u64* ops;
    AlnHeader* hdr;
    unsigned long long* ctr;
    u32 lanes; // sets per wavefront (lanes_per_wave)
    u32 wave_split; // long reads: sets that span >= 1024 query bases go to k_stitch_wave
};

// what k_stitch_wave keeps in LDS of the set it walks: the next 64 seeds and the records of the next 64 jobs (with the first
// four cigar entries of each), loaded by the 64 lanes at once -- step by step each of them is a memory round trip that all
// lanes wait for (a 50 kb alignment: ~5 k seeds and ~5 k gap fills)
struct StitchWaveCache
{
    u64 sq[ 64 ], sr[ 64 ], sl[ 64 ];
    i32 jq[ 64 ], jt[ 64 ];
    u32 jn[ 64 ];
    u64 joff[ 64 ];
    uint4 jc[ 64 ];
};
struct StitchSinkWave : StitchSink
{
    static const bool WAVE = true;
    StitchWaveCache* C;
    u32 seedBase = 0x80000000u, jobBase = 0x80000000u, nJobs = 0;
    __device__ void seed( const ma_seed* S, u32 n, u32 kk, u64& q, u64& r, u64& l )
    {
        if( kk - seedBase >= 64u )
        {
            const u32 lane = threadIdx.x & 63, i = kk + lane;
            __syncthreads( );
            if( i < n )
            {
                const ma_seed x = S[ i ];
                C->sq[ lane ] = (u64)x.q_start, C->sr[ lane ] = (u64)x.r_start, C->sl[ lane ] = (u64)x.len;
            }
            seedBase = kk;
            __syncthreads( );
        }
        const u32 d = kk - seedBase;
        q = C->sq[ d ], r = C->sr[ d ], l = C->sl[ d ];
    }
    __device__ KswResult next( )
    {
        if( k - jobBase >= 64u )
        {
            const u32 lane = threadIdx.x & 63, i = k + lane;
            __syncthreads( );
            if( i < nJobs )
            {
                const ma_ez e = ez[ i ];
                const u64 off = cig_off[ i ];
                C->jq[ lane ] = e.max_q, C->jt[ lane ] = e.max_t, C->jn[ lane ] = (u32)e.n_cigar, C->joff[ lane ] = off;
                uint4 c = make_uint4( 0, 0, 0, 0 );
                const u32* p = cig_pool + off;
                if( e.n_cigar > 0 )
                    c.x = p[ 0 ];
                if( e.n_cigar > 1 )
                    c.y = p[ 1 ];
                if( e.n_cigar > 2 )
                    c.z = p[ 2 ];
                if( e.n_cigar > 3 )
                    c.w = p[ 3 ];
                C->jc[ lane ] = c;
            }
            jobBase = k;
            __syncthreads( );
        }
        const u32 d = k - jobBase;
        KswResult R;
        R.max_q = C->jq[ d ], R.max_t = C->jt[ d ], R.n_cigar = C->jn[ d ];
        R.cigar = cig_pool + C->joff[ d ];
        const uint4 c = C->jc[ d ];
        R.first[ 0 ] = c.x, R.first[ 1 ] = c.y, R.first[ 2 ] = c.z, R.first[ 3 ] = c.w;
        R.cached = true;
        k++;
        return R;
    }
};
// Sets whose walk is long enough to be worth a wavefront of their own (k_stitch_wave): the seeds span >= 1024 query bases.
// (A batch of 20 k reads of 50 kb has 2 * 10^5 sets; the ~10 % that span the read are ~all of the bases to compare, and as
// lanes of the one-set-per-lane kernel each of them kept its wavefront busy for its whole length: 69 ms.)
__device__ __forceinline__ bool stitch_is_big( const StitchKernelArgs& A, u32 s )
{
    if( !A.wave_split )
        return false;
    const HSet hs = A.sets[ s ];
    if( hs.cnt == 0 || !A.info[ s ].valid )
        return false;
    const ma_seed first = A.hpool[ hs.off ], last = A.hpool[ hs.off + hs.cnt - 1 ];
    return (u64)last.q_start + (u64)last.len >= (u64)first.q_start + 1024;
}
__device__ __forceinline__ void stitch_sink_setup( StitchSink&, const SetInfo&, StitchWaveCache* )
{}
__device__ __forceinline__ void stitch_sink_setup( StitchSinkWave& sink, const SetInfo& I, StitchWaveCache* cache )
{
    sink.C = cache;
    sink.nJobs = I.n_jobs;
}
template <typename SINK> __device__ __forceinline__ u64 stitch_set( const StitchKernelArgs& A, u32 s, StitchWaveCache* cache )
{
    const HSet hs = A.sets[ s ];
    const u32 rd = A.set_read[ s ];
    const SetInfo I = A.info[ s ];
    AlnHeader h;
    h.begin_ref = h.end_ref = 0;
    h.begin_q = h.end_q = 0;
    h.score = 0;
    h.length = 0;
    h.ops_off = A.ops_off[ s ];
    h.n_ops = 0;
    h.ops_cap = (u32)A.ops_cap[ s ];
    h.soc_index = hs.soc;
    h.secondary = h.supplementary = 0;
    h.mapq = NAN;
    u32 err = 0;
    if( I.valid )
    {
        h.begin_ref = h.end_ref = I.win_begin;
        NwWindow W;
        W.begin_ref = I.win_begin;
        W.end_ref = I.win_end;
        W.valid = true;
        SINK sink;
        sink.ez = A.ez + 2 * hs.off, sink.cig_off = A.cig_off + 2 * hs.off, sink.cig_pool = A.cig_pool, sink.k = 0;
        stitch_sink_setup( sink, I, cache );
        NwWalk<SINK> walk{ A.X, A.P, sink, A.reads + A.roff[ rd ], I.win_begin, AlnBuilder{ &h, A.ops + h.ops_off, &err } };
        walk.run( A.hpool + hs.off, hs.cnt, A.roff[ rd + 1 ] - A.roff[ rd ], W );
    }
    if( !sink_is_wave<SINK>::value || ( threadIdx.x & 63 ) == 0 )
    {
        A.hdr[ s ] = h;
        if( err )
            atomicOr( (unsigned long long*)&A.ctr[ CTR_ERR ], (unsigned long long)err );
    }
    return h.n_ops;
}
__global__ void __launch_bounds__( 64 ) __attribute__( ( amdgpu_waves_per_eu( 6 ) ) ) k_stitch( StitchKernelArgs A )
{
    const u32 s = blockIdx.x * A.lanes + threadIdx.x;
    u64 nOps = 0;
    if( threadIdx.x < A.lanes && s < A.n_sets && !stitch_is_big( A, s ) )
        nOps = stitch_set<StitchSink>( A, s, nullptr );
    // exact size of the ops download (all alignments): one atomic per wave
    const u64 total = wave_sum_u64( nOps );
    if( ( threadIdx.x & 63 ) == 0 && total )
        atomicAdd( &A.ctr[ CTR_OPS_ALL ], (unsigned long long)total );
}
// one set per wavefront: the sets k_stitch left out
__global__ void __launch_bounds__( 64 ) k_stitch_wave( StitchKernelArgs A )
{
    const u32 s = blockIdx.x;
    if( !stitch_is_big( A, s ) )
        return;
    __shared__ StitchWaveCache cache;
    const u64 nOps = stitch_set<StitchSinkWave>( A, s, &cache );
    if( threadIdx.x == 0 && nOps )
        atomicAdd( &A.ctr[ CTR_OPS_ALL ], (unsigned long long)nOps );
}

// per read: NeedlemanWunsch::execute's final sort + MappingQuality::execute
__global__ void k_finish( NwParams P, u32 n_reads, const u64* hset_off, const u64* roff, AlnHeader* hdr, const u64* ops,
                          u32* order, u32* mq_order, u32* mq_cnt, unsigned long long* ctr, int nw_sort )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    const u64 b = r < n_reads ? hset_off[ r ] : 0;
    const u32 n = r < n_reads ? (u32)( hset_off[ r + 1 ] - b ) : 0;
    u32 m = 0;
    u64 opsMq = 0;
    if( r < n_reads )
    {
        m = finish_read( P, hdr + b, ops, n, roff[ r + 1 ] - roff[ r ], order + b, mq_order + b, nw_sort != 0 );
        mq_cnt[ r ] = m;
        for( u32 k = 0; k < m; k++ )
            opsMq += hdr[ b + mq_order[ b + k ] ].n_ops;
    }
    // one atomic per wave and quantity
    const u64 al = wave_sum_u64( m ? 1 : 0 ), am = wave_sum_u64( m ), om = wave_sum_u64( opsMq );
    if( ( threadIdx.x & 63 ) == 0 && al )
    {
        atomicAdd( &ctr[ CTR_N_ALIGNED ], (unsigned long long)al );
        atomicAdd( &ctr[ CTR_OPS_MQ ], (unsigned long long)om );
        atomicAdd( &ctr[ CTR_ALN_MQ ], (unsigned long long)am );
    }
}

// ---- results in the order and layout of the C ABI, packed on the device so that a download is three plain copies:
// per read its alignments (NeedlemanWunsch order, or the MappingQuality selection), their ops as (type, length) pairs
__global__ void k_aln_sizes( u32 n_reads, const u64* hset_off, const AlnHeader* hdr, const u32* order, const u32* mq_cnt, int mq,
                             u64* cnt, u64* nops )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if( r >= n_reads )
        return;
    const u64 b = hset_off[ r ];
    const u32 c = mq ? mq_cnt[ r ] : (u32)( hset_off[ r + 1 ] - b );
    u64 o = 0;
    for( u32 k = 0; k < c; k++ )
        o += hdr[ b + order[ b + k ] ].n_ops;
    cnt[ r ] = c;
    nops[ r ] = o;
}
__global__ void k_aln_pack( u32 n_reads, const u64* hset_off, const AlnHeader* hdr, const u32* order, const u64* pool, int mq,
                            const u64* aln_off, const u64* ops_off, ma_alignment* alns, u64* ops )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if( r >= n_reads )
        return;
    const u64 b = hset_off[ r ];
    const u32 c = (u32)( aln_off[ r + 1 ] - aln_off[ r ] );
    u64 po = ops_off[ r ];
    for( u32 k = 0; k < c; k++ )
    {
        const AlnHeader& h = hdr[ b + order[ b + k ] ];
        ma_alignment a;
        a.begin_ref = (i64)h.begin_ref;
        a.end_ref = (i64)h.end_ref;
        a.begin_q = (i64)h.begin_q;
        a.end_q = (i64)h.end_q;
        a.score = h.score;
        a.soc_index = h.soc_index;
        a.n_ops = h.n_ops;
        a.ops_off = po;
        a.secondary = mq ? h.secondary : 0;
        a.supplementary = mq ? h.supplementary : 0;
        a.mapq = mq ? h.mapq : 0.0;
        alns[ aln_off[ r ] + k ] = a;
        for( u32 j = 0; j < h.n_ops; j++ )
        {
            const u64 o = pool[ h.ops_off + j ];
            ops[ 2 * ( po + j ) ] = op_type( o );
            ops[ 2 * ( po + j ) + 1 ] = op_len( o );
        }
        po += h.n_ops;
    }
}

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
    DevBuf taskA, taskB, taskCnt, taskKey, taskKey2, taskPerm, taskPerm2; // area tasks of long reads (seed_tasks)
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
    if( b->kswSide.fork )
    {
        (void)hipEventDestroy( b->kswSide.fork );
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

__global__ void k_max_qlen( const u64* roff, u64 n, unsigned long long* out )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i < n )
        atomicMax( out, (unsigned long long)( roff[ i + 1 ] - roff[ i ] ) );
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
        hipLaunchKernelGGL( k_max_qlen, dim3( (unsigned)( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream, b->d_roff, n,
                            b->ctr.as<unsigned long long>( ) );
    if( read_ctr( b ) )
        return 1;
    b->max_qlen = (u32)b->hctr[ 0 ];
    b->stage_done = 0;
    return 0;
}

static SeedParams seed_params( const ma_params& P )
{
    SeedParams S;
    S.technique = (u32)P.seeding_technique;
    S.min_seed_len = (u32)P.min_seed_len;
    S.min_amb = (u32)P.min_ambiguity;
    S.max_amb = (u32)P.max_ambiguity;
    S.min_seed_size_drop = (u32)P.min_seed_size_drop;
    S.disable_heuristics = (u32)P.disable_heuristics;
    S.rel_min_seed_size_amount = P.rel_min_seed_size_amount;
    S.genome_size_disable = P.genome_size_disable;
    S.window_begin = S.window_end = nullptr;
    S.smem_compact = 0;
    S.smem_merge = 0;
    return S;
}
// reads that stay in HBM are read through a 16-byte register window (seed_qbyte): the bounds of the reads array
static void seed_window( SeedParams& S, const ma_batch* b, bool on )
{
    const u64 bytes = b->n_bases + ( b->reads_external ? 0 : 64 ); // the batch's own copy is padded
    if( const char* e = getenv( "MA_SEED_WINDOW" ) ) // tuning hook
        on = on && atoi( e ) != 0;
    if( on && bytes >= 16 )
    {
        S.window_begin = b->d_reads;
        S.window_end = b->d_reads + bytes;
    }
}

static int seed_mems( ma_batch* b )
{
    const u64 n = b->n_reads, nb = b->n_bases;
    if( b->segOff.reserve( ( n + 1 ) * 8 ) || b->segCnt.reserve( ( n + 1 ) * 4 ) || b->memsCnt.reserve( ( nb + 2 ) * 8 ) ||
        b->memsOff.reserve( ( nb + 2 ) * 8 ) )
        return 1;
    MemsArgs A;
    A.X = b->idx->v;
    A.P = seed_params( b->P );
    A.reads = b->d_reads;
    A.roff = b->d_roff;
    A.n_reads = (u32)n;
    A.n_bases = nb;
    A.cnt = b->memsCnt.as<u64>( );
    A.off = b->memsOff.as<u64>( );
    A.pool = nullptr;
    A.pool_read = nullptr;
    A.ctr = b->ctr.as<unsigned long long>( );
    EvTimer t( b, 0 );
    u64 total = 0;
    if( nb )
    {
        hipLaunchKernelGGL( k_mems<false>, dim3( (unsigned)( ( nb + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream, A );
        MA_HIP( hipMemsetAsync( (char*)b->memsCnt.p + nb * 8, 0, 8, b->stream ) );
        if( scan_exclusive<u64>( b, b->memsCnt.as<u64>( ), b->memsOff.as<u64>( ), nb + 1 ) )
            return 1;
        MA_HIP( hipMemcpyAsync( &total, (char*)b->memsOff.p + nb * 8, 8, hipMemcpyDeviceToHost, b->stream ) );
        if( batch_wait( b ) )
        return 1;
    }
    else
        MA_HIP( hipMemsetAsync( b->memsOff.p, 0, 16, b->stream ) );
    b->segPoolCap = std::max<u64>( total + 1024, b->segPoolCap );
    if( b->segPool.reserve( b->segPoolCap * sizeof( ma_segment ) ) || b->segRead.reserve( b->segPoolCap * 4 ) )
        return 1;
    A.pool = b->segPool.as<ma_segment>( );
    A.pool_read = b->segRead.as<u32>( );
    if( nb && total )
        hipLaunchKernelGGL( k_mems<true>, dim3( (unsigned)( ( nb + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream, A );
    hipLaunchKernelGGL( k_mems_finish, dim3( (unsigned)( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream, A.X, A.P, b->d_roff, (u32)n,
                        b->memsOff.as<u64>( ), b->segPool.as<ma_segment>( ), b->segOff.as<u64>( ), b->segCnt.as<u32>( ) );
    const unsigned long long used = total;
    MA_HIP( hipMemcpyAsync( b->ctr.as<unsigned long long>( ) + CTR_SEG_USED, &used, 8, hipMemcpyHostToDevice, b->stream ) );
    if( batch_wait( b ) )
        return 1;
    MA_HIP( hipGetLastError( ) );
    b->stage_done = 1;
    return 0;
}

// maxSpan seeding of long reads as area tasks (k_seed_tasks); returns 2 when the task arrays were too small (the caller
// falls back to the read-per-lane kernel)
// First-attempt size of the segment pool.  Measured: 0.017 - 0.019 maxSpan segments per base (150 bp, 10 kb and 50 kb
// reads against GRCh38-like references); the pool takes 1/16 per base + 16 per read (>3x that), 24 bytes each plus the
// sort keys of the task kernel.  It used to be 1/2 per base: 45 GB for a 2 Gbase batch of which 0.9 GB were used, which
// kept a second long-read batch from being in flight on the same GPU.
static u64 seg_pool_heuristic( u64 n_bases, u64 n_reads )
{
    return std::max<u64>( n_bases / 16 + 16 * n_reads, 1024 );
}

static int seed_tasks( ma_batch* b )
{
    const u64 n = b->n_reads, nb = b->n_bases;
    int levels = 2;
    for( u32 q = b->max_qlen; q > 1; q >>= 1 )
        levels++;
    if( levels >= MA_TASK_KEY_BITS / 2 )
        return 2;
    const u64 taskCap = nb / 16 + 2 * n + 1024;
    b->segPoolCap = std::max( seg_pool_heuristic( nb, n ), b->segPoolMin );
    if( b->segOff.reserve( ( n + 1 ) * 8 ) || b->segCnt.reserve( ( n + 1 ) * 4 ) || b->taskA.reserve( taskCap * sizeof( SeedTask ) ) ||
        b->taskB.reserve( taskCap * sizeof( SeedTask ) ) || b->taskCnt.reserve( 64 * 8 ) ||
        b->segPool.reserve( b->segPoolCap * sizeof( ma_segment ) ) || b->segRead.reserve( b->segPoolCap * 4 ) ||
        b->stage.reserve( b->segPoolCap * sizeof( ma_segment ) ) || b->taskKey.reserve( b->segPoolCap * 8 ) ||
        b->taskKey2.reserve( b->segPoolCap * 8 ) || b->taskPerm.reserve( b->segPoolCap * 4 ) || b->taskPerm2.reserve( b->segPoolCap * 4 ) )
        return 1;
    unsigned long long* cnt = b->taskCnt.as<unsigned long long>( );
    MA_HIP( hipMemsetAsync( cnt, 0, 64 * 8, b->stream ) );
    TaskKernelArgs A;
    A.X = b->idx->v;
    A.P = seed_params( b->P );
    seed_window( A.P, b, true );
    A.slow_batch = 4;
    if( const char* e = getenv( "MA_SEED_SLOW_BATCH" ) ) // tuning hook
        A.slow_batch = (u32)std::max( 1, atoi( e ) );
    A.reads = b->d_reads;
    A.roff = b->d_roff;
    A.task_cap = taskCap;
    A.pool = b->stage.as<ma_segment>( ); // unsorted
    A.pool_key = b->taskKey.as<u64>( );
    A.pool_cap = b->segPoolCap;
    A.ctr = b->ctr.as<unsigned long long>( );
    {
        EvTimer t( b, 0 );
        hipLaunchKernelGGL( k_task_roots, dim3( (unsigned)( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream, b->d_roff, (u32)n,
                            b->taskA.as<SeedTask>( ), cnt );
        for( int lv = 0; lv < levels; lv++ )
        {
            A.in = ( lv & 1 ) ? b->taskB.as<SeedTask>( ) : b->taskA.as<SeedTask>( );
            A.out = ( lv & 1 ) ? b->taskA.as<SeedTask>( ) : b->taskB.as<SeedTask>( );
            A.nIn = cnt + lv;
            A.nOut = cnt + lv + 1;
            MA_HIP( hipMemsetAsync( A.ctr + CTR_NEXT_READ, 0, 8, b->stream ) );
            hipLaunchKernelGGL( k_seed_tasks, dim3( 2048 ), dim3( 256 ), 0, b->stream, A );
        }
    }
    MA_HIP( hipGetLastError( ) );
    if( read_ctr( b ) )
        return 1;
    const u32 err = (u32)b->hctr[ CTR_ERR ];
    const u64 ns = b->hctr[ CTR_SEG_USED ];
    if( ( err & MA_ERR_STACK_OVERFLOW ) )
    {
        MA_HIP( hipMemsetAsync( b->ctr.p, 0, CTR_COUNT * 8, b->stream ) );
        return 2; // task array too small: the classic kernel takes over
    }
    if( ( err & MA_ERR_SEG_OVERFLOW ) || ns > b->segPoolCap )
    {
        b->segPoolMin = ns + 1024; // counted need: run again
        MA_HIP( hipMemsetAsync( b->ctr.p, 0, CTR_COUNT * 8, b->stream ) );
        return seed_tasks( b );
    }
    MA_HIP( hipMemsetAsync( b->segCnt.p, 0, ( n + 1 ) * 4, b->stream ) );
    if( ns )
    {
        hipLaunchKernelGGL( k_iota32, dim3( (unsigned)( ( ns + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream, b->taskPerm.as<u32>( ), ns );
        size_t tb = 0;
        MA_HIP( hipcub::DeviceRadixSort::SortPairs( nullptr, tb, b->taskKey.as<u64>( ), b->taskKey2.as<u64>( ), b->taskPerm.as<u32>( ),
                                                    b->taskPerm2.as<u32>( ), (int)ns, 0, 64, b->stream ) );
        if( b->cubTmp.reserve( tb + 256 ) )
            return 1;
        MA_HIP( hipcub::DeviceRadixSort::SortPairs( b->cubTmp.p, tb, b->taskKey.as<u64>( ), b->taskKey2.as<u64>( ), b->taskPerm.as<u32>( ),
                                                    b->taskPerm2.as<u32>( ), (int)ns, 0, 64, b->stream ) );
        const dim3 grid( (unsigned)( ( ns + 255 ) / 256 ) ), block( 256 );
        hipLaunchKernelGGL( k_task_permute, grid, block, 0, b->stream, b->stage.as<ma_segment>( ), b->taskKey2.as<u64>( ),
                            b->taskPerm2.as<u32>( ), ns, b->segPool.as<ma_segment>( ), b->segRead.as<u32>( ) );
        hipLaunchKernelGGL( k_task_ranges, grid, block, 0, b->stream, b->segRead.as<u32>( ), ns, b->segOff.as<u64>( ), b->segCnt.as<u32>( ) );
    }
    hipLaunchKernelGGL( k_task_finish, dim3( (unsigned)( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream, b->idx->v, seed_params( b->P ),
                        b->d_roff, (u32)n, b->segPool.as<ma_segment>( ), b->segOff.as<u64>( ), b->segCnt.as<u32>( ) );
    MA_HIP( hipGetLastError( ) );
    b->stage_done = 1;
    return 0;
}

int ma_seed_batch( ma_batch* b )
{
    if( !b || !b->d_roff )
        return fail( "ma_seed_batch: no reads set" );
    MA_BIND_DEVICE( b->device );
    const u64 n = b->n_reads;
    MA_HIP( hipMemsetAsync( b->ctr.p, 0, CTR_COUNT * 8, b->stream ) );
    b->nSegs = b->nSeeds = b->nHsets = b->nHseeds = 0;
    if( n == 0 )
    {
        b->stage_done = 1;
        return 0;
    }
    if( b->P.seeding_technique == 2 )
        return seed_mems( b );
    // few long reads: one lane per AREA of the recursion instead of one per read.  With >= 128 k reads in the batch the
    // read-per-lane kernel already fills the machine and is faster (10 kb x 200 k reads: 157 vs 184 ms; the level-by-level
    // walk pays a tail per level), with 20 k reads of 50 kb the task kernel is 6.5x faster (83 vs 546 ms).
    // MA_SEED_TASKS=0 / 1 forces the choice (tests, tuning)
    {
        bool tasks = b->P.seeding_technique == 0 && b->max_qlen > 240 && n < 131072;
        if( const char* e = getenv( "MA_SEED_TASKS" ) )
            tasks = b->P.seeding_technique == 0 && atoi( e ) != 0;
        if( tasks )
        {
            const int rc = seed_tasks( b );
            if( rc != 2 )
                return rc;
        }
    }
    const bool smem = b->P.seeding_technique == 1;
    const u32 worst_cap = ( smem ? 6 : 2 ) * b->max_qlen + 8; // segments one read can emit at most
    const u32 smem_cap = smem ? ( ( b->max_qlen + 3 ) & ~1u ) : 0; // even: a lane's list of 16-byte compact entries stays 16-byte aligned
    b->segPoolCap = seg_pool_heuristic( b->n_bases, n );
    if( smem )
        b->segPoolCap *= 4;
    // the pool size above is a heuristic (a read can emit up to 2x / 6x its length in segments): a batch that needs more
    // is seeded again with the counted need (segPoolMin, kept for the later batches of this object)
    b->segPoolCap = std::max( b->segPoolCap, b->segPoolMin );
    if( const char* e = getenv( "MA_SEG_POOL_CAP" ) ) // test hook: force a (too) small pool on the first attempt
        if( b->segPoolMin == 0 )
            b->segPoolCap = (u64)std::max( 1, atoi( e ) );
    // Resident lanes: up to 8 waves per SIMD on 256 CUs, bounded by the reads and by a staging budget of a third of
    // the free HBM.  A lane walks its read serially, so lanes in flight are what hides the gather latency; for long
    // reads the worst-case staging (0.8 MB per 10 kb read) would leave too few of them, so the first attempt stages
    // a quarter of a segment per base (>10x what reads produce: 242 segments per 10 kb read) and the
    // stage is repeated with the worst case if any read overflowed.
    size_t freeB = 0, totalB = 0;
    MA_HIP( hipMemGetInfo( &freeB, &totalB ) );
    const u64 have = b->stage.cap + b->smemA.cap + b->smemB.cap; // already ours
    const u64 budget = std::max<u64>( 8ull << 30, ( (u64)freeB + have ) / 3 );
    u64 want = std::min<u64>( 256ull * 2048, ( n + 255 ) / 256 * 256 );
    if( const char* e = getenv( "MA_SEED_LANES" ) ) // tuning hook: resident lanes of the read-per-lane kernels
        want = std::min<u64>( want, std::max<u64>( 256, (u64)atoll( e ) / 256 * 256 ) );
    u32 seg_cap = worst_cap;
    if( want * ( (u64)worst_cap * sizeof( ma_segment ) + 2ull * smem_cap * sizeof( ma_segment ) ) > budget )
        seg_cap = std::min<u32>( worst_cap, ( smem ? 3 : 1 ) * ( b->max_qlen / 4 ) + 64 );
    if( const char* e = getenv( "MA_SEED_STAGE_CAP" ) ) // test hook: force a (too) small first attempt
        seg_cap = std::min<u32>( worst_cap, (u32)std::max( 1, atoi( e ) ) );
    for( int attempt = 0; attempt < 3; attempt++ )
    {
        const u64 lane_bytes = (u64)seg_cap * sizeof( ma_segment ) + 2ull * smem_cap * sizeof( ma_segment );
        const u64 lanes = std::min<u64>( want, std::max<u64>( 256, ( budget / lane_bytes ) / 256 * 256 ) );
        if( b->stage.reserve( lanes * seg_cap * sizeof( ma_segment ) ) ||
            ( smem && ( b->smemA.reserve( lanes * smem_cap * sizeof( ma_segment ) ) ||
                        b->smemB.reserve( lanes * smem_cap * sizeof( ma_segment ) ) ) ) ||
            b->segPool.reserve( b->segPoolCap * sizeof( ma_segment ) ) || b->segRead.reserve( b->segPoolCap * 4 ) ||
            b->segOff.reserve( n * 8 ) || b->segCnt.reserve( n * 4 ) || b->seedStack.reserve( lanes * 2 * MA_SEED_STACK * 4 ) )
            return 1;
        SeedKernelArgs A;
        A.X = b->idx->v;
        A.P = seed_params( b->P );
        A.reads = b->d_reads;
        A.roff = b->d_roff;
        A.n_reads = (u32)n;
        A.stage = b->stage.as<ma_segment>( );
        A.seg_cap = seg_cap;
        A.smem_a = smem ? b->smemA.as<ma_segment>( ) : nullptr;
        A.smem_b = smem ? b->smemB.as<ma_segment>( ) : nullptr;
        A.smem_cap = smem_cap;
        A.stack = b->seedStack.as<u32>( );
        A.pool = b->segPool.as<ma_segment>( );
        A.pool_read = b->segRead.as<u32>( );
        A.pool_cap = b->segPoolCap;
        A.seg_off = b->segOff.as<u64>( );
        A.seg_cnt = b->segCnt.as<u32>( );
        A.ctr = b->ctr.as<unsigned long long>( );
        {
            EvTimer t( b, 0 );
            // reads up to 240 bases are staged in LDS (256 lanes x q_lds bytes <= 64 KB)
            const u32 qb = (u32)( ( b->max_qlen + 7 ) / 8 * 8 + 4 );
            A.q_lds = qb * 256 <= 64 * 1024 ? qb : 0;
            seed_window( A.P, b, A.q_lds == 0 );
            A.P.smem_compact = smem && b->max_qlen < 2048 && b->idx->v.n < ( 1ull << 35 ) ? 1 : 0;
            if( const char* e = getenv( "MA_SMEM_COMPACT" ) ) // tuning / test hook
                A.P.smem_compact = A.P.smem_compact && atoi( e ) != 0 ? 1 : 0;
            A.P.smem_merge = smem && A.P.min_amb == 0 ? 1 : 0;
            if( const char* e = getenv( "MA_SMEM_MERGE" ) ) // test hook: 0 = keep every entry like the reference's lists
                A.P.smem_merge = A.P.smem_merge && atoi( e ) != 0 ? 1 : 0;
            // measured per 1 M x 150 bp reads: maxSpan 8.98 ms (1) / 8.56 (4) / 8.87 (8); SMEMs 114 ms (4) / 96 (8) / 96 (16) / 101 (32)
            // 200 k x 10 kb reads (k_seed_long: a transition costs several memory round trips in a row): 4: 149 ms, 8: 140, 16: 130, 24: 134, 32: 144
            A.slow_batch = A.P.technique == 0 ? ( A.q_lds ? 4 : 16 ) : 8;
            if( const char* e = getenv( "MA_SEED_SLOW_BATCH" ) ) // tuning hook
                A.slow_batch = (u32)std::max( 1, atoi( e ) );
            if( A.q_lds )
                hipLaunchKernelGGL( smem ? k_seed<true> : k_seed<false>, dim3( (unsigned)( lanes / 256 ) ), dim3( 256 ), A.q_lds * 256, b->stream, A );
            else
            {
                if( const char* e = getenv( "MA_SEED_LONG_JUMP" ) ) // A/B + test hook: 0 = walk every run step by step
                    if( atoi( e ) == 0 )
                        A.X.kmer_k = 0;
                hipLaunchKernelGGL( smem ? k_seed_long<true> : k_seed_long<false>, dim3( (unsigned)( lanes / 256 ) ), dim3( 256 ), 0, b->stream, A );
            }
        }
        MA_HIP( hipGetLastError( ) );
        // did every read fit its staging area, and all segments the pool?
        if( read_ctr( b ) )
            return 1;
        if( !( (u32)b->hctr[ CTR_ERR ] & MA_ERR_SEG_OVERFLOW ) )
            break;
        bool retry = false;
        if( b->hctr[ CTR_SEG_USED ] > b->segPoolCap ) // the pool pointer counts every segment, stored or not
        {
            b->segPoolMin = b->segPoolCap = b->hctr[ CTR_SEG_USED ] + 1024;
            retry = true;
        }
        if( seg_cap < worst_cap )
        {
            seg_cap = worst_cap;
            retry = true;
        }
        if( !retry || attempt == 2 )
            break; // surfaces as an error in the next stage
        MA_HIP( hipMemsetAsync( b->ctr.p, 0, CTR_COUNT * 8, b->stream ) );
    }
    b->stage_done = 1;
    return 0;
}

int ma_extract_seeds_batch( ma_batch* b )
{
    if( !b || b->stage_done < 1 )
        return fail( "ma_extract_seeds_batch: run ma_seed_batch first" );
    MA_BIND_DEVICE( b->device );
    const u64 n = b->n_reads;
    if( n == 0 )
    {
        b->socGiven = false, b->stage_done = 2;
        return 0;
    }
    if( read_ctr( b ) || check_err( b, "ma_seed_batch" ) )
        return 1;
    b->nSegs = b->hctr[ CTR_SEG_USED ];
    const u64 ns = b->nSegs;
    if( b->segSeedCnt.reserve( ( ns + 1 ) * 8 ) || b->segSeedOff.reserve( ( ns + 2 ) * 8 ) ||
        b->seedOff.reserve( n * 8 ) || b->seedCnt.reserve( n * 4 ) )
        return 1;
    EvTimer t( b, 1 );
    u64 total = 0;
    if( ns )
    {
        hipLaunchKernelGGL( k_seg_seed_counts, dim3( (unsigned)( ( ns + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream,
                            b->segPool.as<ma_segment>( ), ns, (u32)b->P.min_seed_len, (u32)b->P.max_ambiguity,
                            b->segSeedCnt.as<u64>( ) );
        MA_HIP( hipMemsetAsync( (char*)b->segSeedCnt.p + ns * 8, 0, 8, b->stream ) );
        if( scan_exclusive<u64>( b, b->segSeedCnt.as<u64>( ), b->segSeedOff.as<u64>( ), ns + 1 ) )
            return 1;
        MA_HIP( hipMemcpyAsync( &total, (char*)b->segSeedOff.p + ns * 8, 8, hipMemcpyDeviceToHost, b->stream ) );
        if( batch_wait( b ) )
        return 1;
    }
    b->nSeeds = total;
    if( b->seeds.reserve( ( total + 1 ) * sizeof( ma_seed ) ) )
        return 1;
    hipLaunchKernelGGL( k_read_seed_ranges, dim3( (unsigned)( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream,
                        b->segOff.as<u64>( ), b->segCnt.as<u32>( ), b->segSeedOff.as<u64>( ), ns, total, (u32)n,
                        b->seedOff.as<u64>( ), b->seedCnt.as<u32>( ) );
    if( total )
    {
        if( b->seedRow.reserve( ( total + 1 ) * 8 ) || b->seedSteps.reserve( ( total + 4 ) * 4 ) || b->seedSeg.reserve( ( total + 1 ) * 4 ) )
            return 1;
        unsigned long long* c = b->ctr.as<unsigned long long>( );
        hipLaunchKernelGGL( k_seed_rows, dim3( (unsigned)( ( ns + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream,
                            b->segPool.as<ma_segment>( ), b->segSeedOff.as<u64>( ), ns, b->seedRow.as<i64>( ),
                            b->seedSeg.as<u32>( ) );
        const u64 lanes = std::min<u64>( 256ull * 2048, ( total + 255 ) / 256 * 256 );
        hipLaunchKernelGGL( k_lf_walk, dim3( (unsigned)( lanes / 256 ) ), dim3( 256 ), 0, b->stream, b->idx->v,
                            b->seedRow.as<i64>( ), b->seedSteps.as<u32>( ), total, c + CTR_NEXT_SEED );
        hipLaunchKernelGGL( k_seed_final, dim3( (unsigned)( ( total + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream, b->idx->v,
                            b->segPool.as<ma_segment>( ), b->segRead.as<u32>( ), b->segSeedOff.as<u64>( ),
                            b->seedRow.as<i64>( ), b->seedSteps.as<u32>( ), b->seedSeg.as<u32>( ), total, b->d_roff,
                            b->seeds.as<ma_seed>( ), c );
    }
    MA_HIP( hipGetLastError( ) );
    b->socGiven = false, b->stage_done = 2;
    return 0;
}

// glibc srandom_r + 310 discards (stdlib/random_r.c): state after srand(seed)
static void glibc_srand_ring( u32 seed, u32 ring[ 31 ] )
{
    if( seed == 0 )
        seed = 1;
    i32 word = (i32)seed;
    ring[ 0 ] = (u32)word;
    for( int i = 1; i < 31; i++ )
    {
        const long hi = word / 127773, lo = word % 127773;
        word = (i32)( 16807 * lo - 2836 * hi );
        if( word < 0 )
            word += 2147483647;
        ring[ i ] = (u32)word;
    }
    int f = 3, r = 0;
    for( int i = 0; i < 310; i++ )
    {
        ring[ f ] += ring[ r ];
        if( ++f >= 31 )
            f = 0;
        if( ++r >= 31 )
            r = 0;
    }
    // after 310 = 10*31 steps f and r are back at 3 and 0
}

static ChainParams chain_params( const ma_params& P )
{
    ChainParams C;
    C.max_num_soc = (u32)P.max_num_soc;
    C.min_num_soc = (u32)P.min_num_soc;
    C.harm_score_min = (u32)P.harm_score_min;
    C.max_score_lookahead = (u32)P.max_score_lookahead;
    C.switch_qlen = (u32)P.switch_qlen;
    C.min_delta_dist = (u32)P.min_delta_dist;
    C.sv_penalty = (u32)P.sv_penalty;
    C.match = (u32)P.match;
    C.gap = (u32)P.gap;
    C.extend = (u32)P.extend;
    C.disable_heuristics = (u32)P.disable_heuristics;
    C.soc_width = (u32)P.soc_width;
    C.genome_size_disable = P.genome_size_disable;
    C.harm_score_min_rel = P.harm_score_min_rel;
    C.soc_score_decrease_tol = P.soc_score_decrease_tol;
    C.score_diff_tol = P.score_diff_tol;
    C.max_delta_dist = P.max_delta_dist;
    glibc_srand_ring( P.srand_seed, C.rng_ring );
    C.libm_probe = (u32)P.libm_probe;
    return C;
}

// Lanes of a wavefront that get a read / a seed set in the one-item-per-lane kernels (k_chain, k_dp_enum, k_stitch).
// Their lanes run long data-dependent loops (std::sort emulation, RANSAC, the walk over the seeds of an alignment), so
// the lanes of a wave diverge and are executed one after the other: a wave costs about the SUM of its lanes.  A batch
// of 1 M short reads fills the machine with full waves; a batch of 20 k long reads is only 313 full waves on 1024
// SIMDs, each serialising 64 lanes (50 kb reads: k_chain 545 ms).  Fewer items per wave spread the same lanes over
// ~4 waves per SIMD.
static u32 lanes_per_wave( u64 items )
{
    if( const char* e = getenv( "MA_LANES_PER_WAVE" ) ) // tuning hook
        return (u32)std::min( 64, std::max( 1, atoi( e ) ) );
    if( items >= 131072 )
        return 64; // >= 2 full waves per SIMD: measured no gain from thinner waves (10 kb x 200 k reads)
    const u64 waves = 256ull * 4 * 4;
    return (u32)std::max<u64>( 1, std::min<u64>( 64, ( items + waves - 1 ) / waves ) );
}

int ma_chain_batch( ma_batch* b )
{
    if( !b || b->stage_done < 2 )
        return fail( "ma_chain_batch: run ma_extract_seeds_batch first" );
    MA_BIND_DEVICE( b->device );
    const u64 n = b->n_reads;
    if( n == 0 )
    {
        b->stage_done = 3;
        return 0;
    }
    const u64 ts = b->nSeeds + 1;
    const u32 set_cap = 2 * (u32)b->P.max_num_soc;
    b->hpoolCap = 3 * ts + 1024;
    if( b->cWork.reserve( ts * sizeof( ma_seed ) ) || b->cMax.reserve( ts * sizeof( SoCEntry ) ) ||
        b->cMm.reserve( ts * sizeof( RefMinMax ) ) || b->cA.reserve( ts * sizeof( ma_seed ) ) ||
        b->cB.reserve( ts * sizeof( ma_seed ) ) || b->cOut.reserve( ts * sizeof( ma_seed ) ) ||
        b->cSh1.reserve( ts * sizeof( Shadow ) ) || b->cSh2.reserve( ts * sizeof( Shadow ) ) ||
        b->cVx.reserve( 3 * ts * 8 ) || b->cVy.reserve( 3 * ts * 8 ) || b->cMed.reserve( 6 * ts * 8 ) ||
        b->cInl.reserve( 3 * ts * 4 ) || b->cBest.reserve( 3 * ts * 4 ) ||
        b->hpool.reserve( b->hpoolCap * sizeof( ma_seed ) ) || b->hlocal.reserve( ( 3 * ts + 16 ) * sizeof( ma_seed ) ) ||
        b->hseedCnt.reserve( ( n + 2 ) * 8 ) || b->hseedOff.reserve( ( n + 2 ) * 8 ) ||
        b->setTab.reserve( n * set_cap * sizeof( HSet ) ) ||
        b->nsets.reserve( ( n + 1 ) * 4 ) || b->hsetOff.reserve( ( n + 2 ) * 8 ) )
        return 1;
    ChainKernelArgs A;
    A.X = b->idx->v;
    A.P = chain_params( b->P );
    A.n_reads = (u32)n;
    A.roff = b->d_roff;
    A.seed_off = b->seedOff.as<u64>( );
    A.seed_cnt = b->seedCnt.as<u32>( );
    A.seeds = b->seeds.as<ma_seed>( );
    A.work = b->cWork.as<ma_seed>( );
    A.maxima = b->cMax.as<SoCEntry>( );
    A.mm = b->cMm.as<RefMinMax>( );
    A.setA = b->cA.as<ma_seed>( );
    A.setB = b->cB.as<ma_seed>( );
    A.outA = b->cOut.as<ma_seed>( );
    A.sh1 = b->cSh1.as<Shadow>( );
    A.sh2 = b->cSh2.as<Shadow>( );
    A.vX = b->cVx.as<double>( );
    A.vY = b->cVy.as<double>( );
    A.med = b->cMed.as<double>( );
    A.inl = b->cInl.as<i32>( );
    A.best = b->cBest.as<i32>( );
    A.hpool = b->hpool.as<ma_seed>( );
    A.hpool_cap = b->hpoolCap;
    A.hlocal = b->hlocal.as<ma_seed>( );
    A.sets = b->setTab.as<HSet>( );
    A.set_cap = set_cap;
    A.nsets = b->nsets.as<u32>( );
    A.ctr = b->ctr.as<unsigned long long>( );
    A.queue = b->socGiven ? b->socIn.as<ma_soc>( ) : nullptr;
    A.queue_cnt = b->socGiven ? b->socInCnt.as<u32>( ) : nullptr;
    A.pre_nmx = nullptr;
    A.pre_sorted = nullptr;
    // long reads (thousands of seeds per read): the sweep's two std::sort calls run as wave-cooperative kernels on arrays in
    // LDS, the window sweep between them and the rest of the stage stay one read per lane (MA_CHAIN_WAVE_SORT=0: all in k_chain)
    const bool waveSortOn = []( ) { // (read on every call: the tests switch it inside one process)
        const char* e = getenv( "MA_CHAIN_WAVE_SORT" );
        return !e || atoi( e ) != 0;
    }( );
    const bool waveSort = waveSortOn && !b->socGiven && b->max_qlen > 254 && b->nSeeds >= 64;
    {
        EvTimer t( b, 2 );
        A.lanes = lanes_per_wave( n );
        if( waveSort )
        {
            if( b->preNmx.reserve( ( n + 1 ) * 4 ) || b->preSorted.reserve( ( n + 1 ) * 4 ) )
                return 1;
            MA_HIP( hipMemsetAsync( b->preSorted.p, 0, ( n + 1 ) * 4, b->stream ) );
            // test hooks: MA_WSORT_MIN / MA_WSORT_SMALL move the thresholds so that small test reads take both launches
            const u32 wsMin = []( ) { const char* e = getenv( "MA_WSORT_MIN" ); return e ? (u32)std::max( 17, atoi( e ) ) : MA_WSORT_MIN; }( );
            const u32 wsSmall = []( ) { const char* e = getenv( "MA_WSORT_SMALL" ); return e ? (u32)std::min<int>( std::max( 17, atoi( e ) ), MA_WSORT_SMALL ) : MA_WSORT_SMALL; }( );
            const u32 ldsSmall = (u32)ws::scratch_bytes( wsSmall ), ldsLarge = (u32)ws::scratch_bytes( MA_WSORT_LARGE );
            MA_HIP( hipFuncSetAttribute( (const void*)k_sort_seeds_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsLarge ) );
            for( int mode = 0; mode < 2; mode++ )
            {
                // reads of up to MA_WSORT_SMALL seeds (several wavefronts per CU), then the larger ones (one per CU)
                hipLaunchKernelGGL( k_sort_seeds_wave, dim3( (unsigned)n ), dim3( 64 ), ldsSmall, b->stream, (u32)n, A.seed_off, A.seed_cnt, A.seeds,
                                    A.work, A.setA, b->preSorted.as<u32>( ), mode, 0u, wsSmall, wsMin, wsSmall );
                hipLaunchKernelGGL( k_sort_seeds_wave, dim3( (unsigned)n ), dim3( 64 ), ldsLarge, b->stream, (u32)n, A.seed_off, A.seed_cnt, A.seeds,
                                    A.work, A.setA, b->preSorted.as<u32>( ), mode, wsSmall + 1, 0xffffffffu, std::max( wsSmall + 1, wsMin ), MA_WSORT_LARGE );
                if( mode == 0 )
                    hipLaunchKernelGGL( k_soc_windows, dim3( (unsigned)( ( n + A.lanes - 1 ) / A.lanes ) ), dim3( 64 ), 0, b->stream, A.X, A.P, (u32)n,
                                        A.lanes, A.roff, A.seed_off, A.seed_cnt, A.work, A.maxima, A.mm, A.setA, b->preSorted.as<u32>( ),
                                        b->preNmx.as<u32>( ) );
            }
            A.pre_nmx = b->preNmx.as<u32>( );
            A.pre_sorted = b->preSorted.as<u32>( );
        }
        hipLaunchKernelGGL( k_chain, dim3( (unsigned)( ( n + A.lanes - 1 ) / A.lanes ) ), dim3( 64 ), 0, b->stream, A );
    }
    MA_HIP( hipGetLastError( ) );
    // CSR of sets per read: widen counts to u64 via a scan over u32->u64 transform
    {
        size_t tb = 0;
        auto in = hipcub::TransformInputIterator<u64, hipcub::CastOp<u64>, const u32*>( b->nsets.as<u32>( ),
                                                                                      hipcub::CastOp<u64>( ) );
        MA_HIP( hipMemsetAsync( (char*)b->nsets.p + n * 4, 0, 4, b->stream ) );
        MA_HIP( hipcub::DeviceScan::ExclusiveSum( nullptr, tb, in, b->hsetOff.as<u64>( ), (int)( n + 1 ), b->stream ) );
        if( b->cubTmp.reserve( tb + 256 ) )
            return 1;
        MA_HIP( hipcub::DeviceScan::ExclusiveSum( b->cubTmp.p, tb, in, b->hsetOff.as<u64>( ), (int)( n + 1 ),
                                                  b->stream ) );
    }
    hipLaunchKernelGGL( k_hseed_counts, dim3( (unsigned)( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream,
                        b->setTab.as<HSet>( ), set_cap, b->nsets.as<u32>( ), (u32)n, b->hseedCnt.as<u64>( ) );
    MA_HIP( hipMemsetAsync( (char*)b->hseedCnt.p + n * 8, 0, 8, b->stream ) );
    if( scan_exclusive<u64>( b, b->hseedCnt.as<u64>( ), b->hseedOff.as<u64>( ), n + 1 ) )
        return 1;
    u64 nh = 0, nhs = 0;
    MA_HIP( hipMemcpyAsync( &nh, (char*)b->hsetOff.p + n * 8, 8, hipMemcpyDeviceToHost, b->stream ) );
    MA_HIP( hipMemcpyAsync( &nhs, (char*)b->hseedOff.p + n * 8, 8, hipMemcpyDeviceToHost, b->stream ) );
    if( read_ctr( b ) || check_err( b, "ma_chain_batch" ) )
        return 1;
    b->nHsets = nh;
    b->nHseeds = nhs;
    if( b->hsetFlat.reserve( ( nh + 1 ) * sizeof( HSet ) ) || b->hsetRead.reserve( ( nh + 1 ) * 4 ) ||
        b->hdense.reserve( ( nhs + 1 ) * sizeof( ma_seed ) ) )
        return 1;
    hipLaunchKernelGGL( k_hset_flatten, dim3( (unsigned)( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream,
                        b->setTab.as<HSet>( ), set_cap, b->nsets.as<u32>( ), b->hsetOff.as<u64>( ), (u32)n,
                        b->hseedOff.as<u64>( ), b->seedOff.as<u64>( ), b->hlocal.as<ma_seed>( ), b->hpool.as<ma_seed>( ),
                        b->hdense.as<ma_seed>( ), b->hsetFlat.as<HSet>( ), b->hsetRead.as<u32>( ) );
    MA_HIP( hipGetLastError( ) );
    b->stage_done = 3;
    return 0;
}

static NwParams nw_params( const ma_params& P )
{
    NwParams N;
    N.max_gap_area = (u32)P.max_gap_area;
    N.padding = (u32)P.padding;
    N.bandwidth_ext = (u32)P.bandwidth_ext;
    N.min_bandwidth_gap = (u32)P.min_bandwidth_gap;
    N.zdrop = (u32)P.zdrop;
    N.sv_penalty = (u32)P.sv_penalty;
    N.match = (u32)P.match;
    N.mismatch = (u32)P.mismatch;
    N.gap = (u32)P.gap;
    N.extend = (u32)P.extend;
    N.kq = (i32)(int8_t)P.gap;
    N.ke = (i32)(int8_t)P.extend;
    N.min_alignment_score = (u32)P.min_alignment_score;
    N.report_n_best = (u32)P.report_n_best;
    N.max_supplementary = (u32)P.max_supplementary;
    N.max_overlap_supplementary = P.max_overlap_supplementary;
    return N;
}

// The DP kernels are bound by VALU issue: two batches' DP stages running at the same time only slow each other down
// (measured: each takes ~1.8x as long), while a DP stage next to another batch's memory-bound seeding / chaining kernels
// does overlap.  With several batches in flight per device (own streams, own host threads) the DP stages therefore take
// turns when MA_DP_EXCLUSIVE=1: one at a time per device.  Measured (tools/overlap_matrix.sh): no gain over free overlap --
// a DP stage is slowed just as much by another batch's seeding / chaining kernels -- so the default is off.
static std::mutex& dp_turn( int device )
{
    static std::mutex turn[ 64 ];
    return turn[ device & 63 ];
}
static bool dp_exclusive( )
{
    const char* e = getenv( "MA_DP_EXCLUSIVE" ); // (read on every call: tools/overlap_matrix.py switches it inside one process)
    return e && atoi( e ) != 0;
}

// MA_DP_ONE_STREAM=1: all kernel classes of a DP stage back to back on the batch's stream, as before round 3 (A/B hook)
static bool dp_one_stream( )
{
    const char* e = getenv( "MA_DP_ONE_STREAM" ); // (read on every call: the tests switch it inside one process)
    return e && atoi( e ) != 0;
}

int ma_dp_batch( ma_batch* b )
{
    if( !b || b->stage_done < 3 )
        return fail( "ma_dp_batch: run ma_chain_batch first" );
    MA_BIND_DEVICE( b->device );
    const u64 n = b->n_reads, nh = b->nHsets, nhs = b->nHseeds;
    if( b->mqCnt.reserve( ( n + 1 ) * 4 ) )
        return 1;
    MA_HIP( hipMemsetAsync( b->mqCnt.p, 0, ( n + 1 ) * 4, b->stream ) );
    // the counters this stage owns start from zero on EVERY call (the stage API is public: a second ma_dp_batch on the same
    // batch must not double the job counts and the download sizes get_alns reads); [0, 8) and CTR_NEXT_SEED belong to the
    // earlier stages
    MA_HIP( hipMemsetAsync( b->ctr.as<unsigned long long>( ) + CTR_CIG_USED, 0, ( CTR_NEXT_SEED - CTR_CIG_USED ) * 8, b->stream ) );
    MA_HIP( hipMemsetAsync( b->ctr.as<unsigned long long>( ) + CTR_OPS_ALL, 0, ( CTR_COUNT - CTR_OPS_ALL ) * 8, b->stream ) );
    if( n == 0 || nh == 0 )
    {
        b->nJobSlots = 0;
        b->stage_done = 4;
        return read_ctr( b );
    }
    const u64 nSlots = 2 * nhs;
    b->nJobSlots = nSlots;
    if( b->jobs.reserve( ( nSlots + 2 ) * sizeof( DpJob ) ) || b->info.reserve( nh * sizeof( SetInfo ) ) ||
        b->ez.reserve( ( nSlots + 2 ) * sizeof( ma_ez ) ) || b->clsLists.reserve( ( ( KSW_N_CLASSES + 1 ) * nSlots + 2 ) * 4 ) || b->cigOff.reserve( ( nSlots + 2 ) * 8 ) ||
        b->opsCap.reserve( ( nh + 1 ) * 8 ) || b->opsOff.reserve( ( nh + 2 ) * 8 ) ||
        b->hdr.reserve( nh * sizeof( AlnHeader ) ) || b->order.reserve( nh * 4 ) || b->mqOrder.reserve( nh * 4 ) )
        return 1;
    const NwParams NP = nw_params( b->P );
    DpKernelArgs D;
    D.X = b->idx->v;
    D.P = NP;
    D.n_sets = (u32)nh;
    D.sets = b->hsetFlat.as<HSet>( );
    D.set_read = b->hsetRead.as<u32>( );
    D.hpool = b->hdense.as<ma_seed>( );
    D.reads = b->d_reads;
    D.roff = b->d_roff;
    D.jobs = b->jobs.as<DpJob>( );
    D.info = b->info.as<SetInfo>( );
    D.ctr = b->ctr.as<unsigned long long>( );
    D.lists = b->clsLists.as<u32>( );
    D.list_stride = nSlots;
    D.SC = KswScoring{ b->P.match, b->P.mismatch, b->P.gap, b->P.extend, b->P.gap2, b->P.extend2 };
    {
        EvTimer t( b, 3 );
        // zero-fill: a slot is a job iff q_to > q_from (pool regions of dropped sets stay empty)
        MA_HIP( hipMemsetAsync( b->ez.p, 0, ( nSlots + 2 ) * sizeof( ma_ez ), b->stream ) );
        MA_HIP( hipMemsetAsync( b->jobs.p, 0, ( nSlots + 2 ) * sizeof( DpJob ), b->stream ) );
        D.lanes = 64; // this kernel's per-wave work is the list building, not the lanes' walks (50 kb: 8.6 ms full waves, 55 ms thin)
        hipLaunchKernelGGL( k_dp_enum, dim3( (unsigned)( ( nh + D.lanes - 1 ) / D.lanes ) ), dim3( 64 ), 0, b->stream, D );
    }
    MA_HIP( hipGetLastError( ) );
    if( read_ctr( b ) || check_err( b, "ma_dp_batch(enumerate)" ) )
        return 1;
    const u64 nJobs = b->hctr[ CTR_N_JOBS ];
    if( nJobs )
    {
        KswSizing S;
        S.state = b->hctr[ CTR_MAX_STATE ];
        S.h = b->hctr[ CTR_MAX_H ];
        S.p = b->hctr[ CTR_MAX_P ];
        S.cig = b->hctr[ CTR_MAX_CIG ];
        S.qlen = b->hctr[ CTR_MAX_QLEN ];
        for( int k = 0; k < KSW_N_CLASSES; k++ )
        {
            S.cls[ k ] = b->hctr[ CTR_CLS0 + k ];
            S.pc[ k ] = b->hctr[ CTR_MAX_PC0 + k ];
            S.cigc[ k ] = b->hctr[ CTR_MAX_CIGC0 + k ];
        }
        S.pRedo = b->hctr[ CTR_MAX_P_REDO ];
        S.cigRedo = b->hctr[ CTR_MAX_CIG_REDO ];
        // every wave of the ksw launches may leave one partly used 4096-word reservation per class launch
        b->cigPoolCap = std::max<u64>( 64 * nJobs + ( 1 << 20 ), b->n_bases / 2 ) + 4096ull * 256 * 32 * 4;
        b->cigPoolCap = std::max( b->cigPoolCap, b->cigPoolMin );
        if( const char* e = getenv( "MA_CIG_POOL_CAP" ) ) // test hook: force a (too) small pool on the first attempt
            if( b->cigPoolMin == 0 )
                b->cigPoolCap = (u64)std::max( 1, atoi( e ) );
        KswScoring SC{ b->P.match, b->P.mismatch, b->P.gap, b->P.extend, b->P.gap2, b->P.extend2 };
        unsigned long long* c = b->ctr.as<unsigned long long>( );
        PipeFetch F{ b->idx->v, b->jobs.as<DpJob>( ), b->d_reads };
        std::unique_lock<std::mutex> xTurn( dp_turn( b->device ), std::defer_lock );
        if( dp_exclusive( ) )
            xTurn.lock( ); // released when the launches below have drained (read_ctr synchronises the stream)
        // the pool size is a heuristic as well: if the cigars did not fit, the DP stage is run again with the counted need
        for( int attempt = 0; attempt < 2; attempt++ )
        {
            if( b->cigPool.reserve( b->cigPoolCap * 4 ) )
                return 1;
            KswOut O;
            O.ez = b->ez.as<ma_ez>( );
            O.cig_off = b->cigOff.as<u64>( );
            O.cig_pool = b->cigPool.as<u32>( );
            O.cig_pool_cap = b->cigPoolCap;
            O.cig_used = c + CTR_CIG_USED;
            O.cells = c + CTR_CELLS;
            O.njobs = c + CTR_KSW_JOBS;
            O.err = (u32*)( c + CTR_ERR );
            O.path = c + CTR_PATH_BYTES;
            O.cig_chunk = 4096;
            O.cig_words = c + CTR_CIG_WORDS;
            {
                EvTimer t( b, 4 );
                // long reads: every kernel class on its own stream (ksw_launch.h), longest jobs first
                const bool longReads = b->max_qlen > 254 && !dp_one_stream( );
                if( longReads && !b->kswSide.ready( ) )
                {
                    MA_HIP( hipEventCreateWithFlags( &b->kswSide.fork, hipEventDisableTiming ) );
                    for( int l = 0; l < 3; l++ )
                    {
                        MA_HIP( hipStreamCreateWithFlags( &b->kswSide.stream[ l ], hipStreamNonBlocking ) );
                        MA_HIP( hipEventCreateWithFlags( &b->kswSide.join[ l ], hipEventDisableTiming ) );
                    }
                }
                if( longReads )
                    for( int k = 0; k < 4; k++ )
                    {
                        const u64 nk = S.cls[ k ];
                        if( nk < 2048 )
                            continue;
                        u32* list = b->clsLists.as<u32>( ) + (u64)k * nSlots;
                        if( b->sortKey.reserve( nk * 4 ) || b->sortKey2.reserve( nk * 4 ) || b->sortVal2.reserve( nk * 4 ) )
                            return 1;
                        hipLaunchKernelGGL( k_job_cost, dim3( (unsigned)( ( nk + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream, F, list, (u32)nk,
                                            b->sortKey.as<u32>( ) );
                        size_t tb = 0;
                        MA_HIP( hipcub::DeviceRadixSort::SortPairsDescending( nullptr, tb, b->sortKey.as<u32>( ), b->sortKey2.as<u32>( ), list,
                                                                              b->sortVal2.as<u32>( ), (int)nk, 0, 32, b->stream ) );
                        if( b->cubTmp.reserve( tb + 256 ) )
                            return 1;
                        MA_HIP( hipcub::DeviceRadixSort::SortPairsDescending( b->cubTmp.p, tb, b->sortKey.as<u32>( ), b->sortKey2.as<u32>( ), list,
                                                                              b->sortVal2.as<u32>( ), (int)nk, 0, 32, b->stream ) );
                        MA_HIP( hipMemcpyAsync( list, b->sortVal2.p, nk * 4, hipMemcpyDeviceToDevice, b->stream ) );
                    }
                if( ksw_run_all( F, SC, (u32)nSlots, S, b->kswScratch, (unsigned int*)( c + CTR_NEXT_SLOTS ), O, b->stream,
                                 b->clsLists.as<u32>( ), nSlots, (unsigned int*)( c + CTR_N_REDO ), (unsigned int*)( c + CTR_NEXT_BIG ),
                                 longReads ? &b->kswSide : nullptr ) )
                    return 1;
                MA_HIP( hipGetLastError( ) );
            }
            if( read_ctr( b ) )
                return 1;
            if( !( (u32)b->hctr[ CTR_ERR ] & MA_ERR_CIGAR_OVERFLOW ) || attempt == 1 )
                break;
            b->cigPoolMin = b->cigPoolCap = b->hctr[ CTR_CIG_USED ] + 4096ull * 256 * 32 * 4;
            MA_HIP( hipMemsetAsync( c + CTR_ERR, 0, ( CTR_KSW_JOBS - CTR_ERR + 1 ) * 8, b->stream ) ); // ERR, CIG_USED, CELLS, KSW_JOBS
            MA_HIP( hipMemsetAsync( c + CTR_PATH_BYTES, 0, 8, b->stream ) );
            MA_HIP( hipMemsetAsync( c + CTR_NEXT_SLOTS, 0, ( CTR_NEXT_SEED - CTR_NEXT_SLOTS ) * 8, b->stream ) ); // queues, N_REDO, CIG_WORDS
            MA_HIP( hipMemsetAsync( c + CTR_NEXT_BIG, 0, 16, b->stream ) );
            MA_HIP( hipMemsetAsync( b->ez.p, 0, ( nSlots + 2 ) * sizeof( ma_ez ), b->stream ) );
        }
        if( check_err( b, "ma_dp_batch(ksw)" ) )
            return 1;
    }
    {
        EvTimer t( b, 5 );
        hipLaunchKernelGGL( k_ops_caps, dim3( (unsigned)( ( nh + 255 ) / 256 ) ), dim3( 256 ), 0, b->stream,
                            b->hsetFlat.as<HSet>( ), b->info.as<SetInfo>( ), b->hsetRead.as<u32>( ), b->d_roff,
                            b->ez.as<ma_ez>( ), (u32)nh, b->opsCap.as<u64>( ) );
        MA_HIP( hipMemsetAsync( (char*)b->opsCap.p + nh * 8, 0, 8, b->stream ) );
        if( scan_exclusive<u64>( b, b->opsCap.as<u64>( ), b->opsOff.as<u64>( ), nh + 1 ) )
            return 1;
        u64 totalOps = 0;
        MA_HIP( hipMemcpyAsync( &totalOps, (char*)b->opsOff.p + nh * 8, 8, hipMemcpyDeviceToHost, b->stream ) );
        if( batch_wait( b ) )
        return 1;
        b->nOpsCap = totalOps;
        if( b->ops.reserve( ( totalOps + 2 ) * 8 ) )
            return 1;
        StitchKernelArgs T;
        T.X = b->idx->v;
        T.P = NP;
        T.n_sets = (u32)nh;
        T.sets = b->hsetFlat.as<HSet>( );
        T.set_read = b->hsetRead.as<u32>( );
        T.info = b->info.as<SetInfo>( );
        T.hpool = b->hdense.as<ma_seed>( );
        T.reads = b->d_reads;
        T.roff = b->d_roff;
        T.ez = b->ez.as<ma_ez>( );
        T.cig_off = b->cigOff.as<u64>( );
        T.cig_pool = b->cigPool.as<u32>( );
        T.ops_off = b->opsOff.as<u64>( );
        T.ops_cap = b->opsCap.as<u64>( );
        T.ops = b->ops.as<u64>( );
        T.hdr = b->hdr.as<AlnHeader>( );
        T.ctr = b->ctr.as<unsigned long long>( );
        T.lanes = lanes_per_wave( nh );
        T.wave_split = b->max_qlen >= 1024 ? 1 : 0;
        if( const char* e = getenv( "MA_STITCH_WAVE" ) ) // A/B + test hook
            T.wave_split = T.wave_split && atoi( e ) != 0 ? 1 : 0;
        hipLaunchKernelGGL( k_stitch, dim3( (unsigned)( ( nh + T.lanes - 1 ) / T.lanes ) ), dim3( 64 ), 0, b->stream, T );
        if( T.wave_split )
            hipLaunchKernelGGL( k_stitch_wave, dim3( (unsigned)nh ), dim3( 64 ), 0, b->stream, T );
        hipLaunchKernelGGL( k_finish, dim3( (unsigned)( ( n + 63 ) / 64 ) ), dim3( 64 ), 0, b->stream, NP, (u32)n,
                            b->hsetOff.as<u64>( ), b->d_roff, b->hdr.as<AlnHeader>( ), b->ops.as<u64>( ),
                            b->order.as<u32>( ), b->mqOrder.as<u32>( ), b->mqCnt.as<u32>( ),
                            b->ctr.as<unsigned long long>( ), 1 );
    }
    MA_HIP( hipGetLastError( ) );
    b->stage_done = 4;
    return 0;
}

// MappingQuality::execute (mappingQuality.cpp:11-131) ALONE, for alignments that were computed elsewhere (the reference's
// NeedlemanWunsch in a mixed graph): per read its alignments in the order NeedlemanWunsch::execute left them
// (needlemanWunsch.h:131-132), ops as (type, length) pairs like ma_batch_get_alignments returns them.  Afterwards
// ma_batch_get_mapq_alignments serves the MappingQuality selection (ma_batch_get_alignments: the input, unchanged).
static int reset_ctr( ma_batch* b );
int ma_batch_set_alignments( ma_batch* b, const uint64_t* aln_off, const ma_alignment* alns, const uint64_t* ops )
{
    if( !b || !b->d_roff || !aln_off )
        return fail( "ma_batch_set_alignments: no reads set or null argument" );
    const u64 n = b->n_reads, na = aln_off[ n ];
    if( na && !alns )
        return fail( "ma_batch_set_alignments: null argument" );
    MA_BIND_DEVICE( b->device );
    u64 no = 0;
    for( u64 i = 0; i < na; i++ )
        no += alns[ i ].n_ops;
    if( no && !ops )
        return fail( "ma_batch_set_alignments: null argument" );
    std::vector<AlnHeader> h( na + 1 );
    std::vector<u64> pk( no + 1 );
    u64 w = 0;
    for( u64 i = 0; i < na; i++ )
    {
        const ma_alignment& a = alns[ i ];
        AlnHeader& x = h[ i ];
        x.begin_ref = (u64)a.begin_ref, x.end_ref = (u64)a.end_ref, x.begin_q = (u64)a.begin_q, x.end_q = (u64)a.end_q;
        x.score = a.score;
        x.length = 0;
        x.ops_off = w;
        x.n_ops = x.ops_cap = a.n_ops;
        x.soc_index = a.soc_index;
        x.secondary = a.secondary, x.supplementary = a.supplementary; // Alignment::larger reads them (all 0 after the DP stage)
        x.mapq = a.mapq;
        for( u32 k = 0; k < a.n_ops; k++ )
        {
            const u64 t = ops[ 2 * ( a.ops_off + k ) ], l = ops[ 2 * ( a.ops_off + k ) + 1 ];
            if( t > MT_DEL )
                return fail( "ma_batch_set_alignments: unknown match type" );
            x.length += l;
            pk[ w++ ] = op_pack( (u32)t, l );
        }
    }
    if( reset_ctr( b ) || b->hsetOff.reserve( ( n + 2 ) * 8 ) || b->hdr.reserve( ( na + 1 ) * sizeof( AlnHeader ) ) ||
        b->ops.reserve( ( no + 2 ) * 8 ) || b->order.reserve( ( na + 1 ) * 4 ) || b->mqOrder.reserve( ( na + 1 ) * 4 ) ||
        b->mqCnt.reserve( ( n + 1 ) * 4 ) )
        return 1;
    MA_HIP( hipMemcpyAsync( b->hsetOff.p, aln_off, ( n + 1 ) * 8, hipMemcpyHostToDevice, b->stream ) );
    if( na )
        MA_HIP( hipMemcpyAsync( b->hdr.p, h.data( ), na * sizeof( AlnHeader ), hipMemcpyHostToDevice, b->stream ) );
    if( no )
        MA_HIP( hipMemcpyAsync( b->ops.p, pk.data( ), no * 8, hipMemcpyHostToDevice, b->stream ) );
    MA_HIP( hipMemsetAsync( b->mqCnt.p, 0, ( n + 1 ) * 4, b->stream ) );
    const unsigned long long all = no;
    MA_HIP( hipMemcpyAsync( b->ctr.as<unsigned long long>( ) + CTR_OPS_ALL, &all, 8, hipMemcpyHostToDevice, b->stream ) );
    if( n && na )
        hipLaunchKernelGGL( k_finish, dim3( (unsigned)( ( n + 63 ) / 64 ) ), dim3( 64 ), 0, b->stream, nw_params( b->P ), (u32)n,
                            b->hsetOff.as<u64>( ), b->d_roff, b->hdr.as<AlnHeader>( ), b->ops.as<u64>( ), b->order.as<u32>( ),
                            b->mqOrder.as<u32>( ), b->mqCnt.as<u32>( ), b->ctr.as<unsigned long long>( ), 0 );
    MA_HIP( hipGetLastError( ) );
    if( batch_wait( b ) )
        return 1; // the host vectors go out of scope
    b->nHsets = na; // one alignment per harmonized set: the bookkeeping get_alns walks
    b->nHseeds = 0;
    b->nJobSlots = 0;
    b->stage_done = 4;
    return 0;
}

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
    out[ 4 ] = b->hctr[ CTR_CELLS ];
    out[ 5 ] = b->hctr[ CTR_KSW_JOBS ];
    out[ 6 ] = b->hctr[ CTR_SEQ_BYTES ];
    out[ 7 ] = b->hctr[ CTR_PATH_BYTES ] + 4 * b->hctr[ CTR_CIG_WORDS ];
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

static int get_alns( ma_batch* b, bool mq, uint64_t* aln_off, ma_alignment* alns, uint64_t* ops )
{
    if( !b || b->stage_done < 4 )
        return fail( "ma_batch_get_alignments: stage not run" );
    MA_BIND_DEVICE( b->device );
    if( ma_batch_sync( b ) )
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
    if( aln_off )
        MA_HIP( hipMemcpyAsync( aln_off, b->outAlnOff.p, ( n + 1 ) * 8, hipMemcpyDeviceToHost, b->stream ) );
    if( alns && totalA )
        MA_HIP( hipMemcpyAsync( alns, b->outAlns.p, totalA * sizeof( ma_alignment ), hipMemcpyDeviceToHost, b->stream ) );
    if( ops && totalO )
        MA_HIP( hipMemcpyAsync( ops, b->outOpsPairs.p, totalO * 16, hipMemcpyDeviceToHost, b->stream ) );
    if( batch_wait( b ) )
        return 1;
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
int ma_debug_seed_prof( unsigned long long* out )
{
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( g_seed_prof ), 8 * 8 ) );
    return 0;
}
#endif

} // extern "C"
