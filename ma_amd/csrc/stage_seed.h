// stage_seed.h -- kernels of the seeding stage (BinarySeeding, binarySeeding.cpp:32-178): k_seed / k_seed_long (one read per lane,
// maxSpan and SMEM state machines of seeding.h), k_mems (MEMs, one lane per start position), k_seed_tasks (one lane per
// area of the maxSpan recursion for few long reads) with its ordering pass.  Textually part of pipeline.hip.
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
// entries of each SMEM list a lane of k_seed_tasks_smem keeps in LDS: 3 waves per SIMD = 3 workgroups per CU x 48 KB.  (The read-per-lane
// kernel k_seed_long<true> keeps its lists in HBM: with 4 heads in LDS it spilled 8 registers and ran 2140 ms for 1848 on 50 kb x 10 k reads.)
#define MA_SMEM_LDS_HEADS_TASK 6
// One read per lane at a time; lanes refill from a global queue, so a wavefront keeps stepping 64
// reads in lockstep through extend_backward until the batch is exhausted.
// LONG: the reads stay in HBM (longer than 240 bases) and are read through the register window of seed_qbyte.
template <bool LONG, bool SM, bool MS = !SM> __device__ __forceinline__ void seed_kernel_body( const SeedKernelArgs& A )
{
    const u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    SeedScratch S;
    S.stage = A.stage + (u64)lane * A.seg_cap;
    S.seg_cap = A.seg_cap;
    // a lane's list: smem_cap entries of 16 bytes (compact) or 40 bytes
    const u64 smemWords = (u64)A.smem_cap * ( A.P.smem_compact ? 2u : 5u );
    S.smem_a = A.smem_a ? (ma_segment*)( (u64*)A.smem_a + (u64)lane * smemWords ) : nullptr;
    S.smem_b = A.smem_b ? (ma_segment*)( (u64*)A.smem_b + (u64)lane * smemWords ) : nullptr;
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

// ---- task-parallel maxSpan / SMEM seeding for long reads -----------------------------------------------------
// procesInterval (binarySeeding.cpp:32-84) is a binary recursion: the centre of an area is extended, then the part left
// of the covered interval and the part right of it are processed independently.  A read-per-lane walk leaves a 50 kb
// read on ONE lane (20 k reads = 1.2 wavefronts per CU); here every AREA is a task.  The tree is walked level by level
// (the centre is the middle of its area, so an area halves from level to level: depth <= log2(read length) + 1); the
// lanes of a level's launch pull tasks from the level's array, extend, append their 0..2 segments to the pool with the
// task's PRE-ORDER key -- node before its left subtree before its right subtree, two bits per level: exactly the order in
// which the recursion pushes segments -- and append the child areas to the next level's array.  A stable sort by
// (read, key) then restores the reference's segment order.
// Round 6: the same for SMEM seeding (template SM; the Nanopore preset: binarySeeding.h:261-452 under the
// same recursion, binarySeeding.cpp:41-83).  A centre's smemExtension emits a variable number of segments -- they leave the lane's
// staging area in emission order under ONE key, and the sort is stable -- and needs the lane's two pending lists; both are sized
// for what reads produce (seg_cap, smem_cap), and a task that outgrows either raises MA_ERR_SMEM_OVERFLOW: the host then runs the
// read-per-lane kernel, which sizes them for the worst case.
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
    // SMEM tasks: per-lane staging of a centre's segments and the two pending lists (seeding.h).  (Round 6 also built and measured
    // "areas of up to L bases walked as whole subtrees by one lane" -- slower for every L, DESIGN.md section 3.1, commit 6600a6f.)
    ma_segment* stage;
    u32 seg_cap;
    ma_segment* smem_a;
    ma_segment* smem_b;
    u32 smem_cap;
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
template <bool SM> __device__ __forceinline__ void seed_tasks_body( const TaskKernelArgs& A )
{
    constexpr bool MS = !SM;
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
    if( SM )
    {
        const u64 lane = (u64)blockIdx.x * blockDim.x + threadIdx.x;
        const u64 smemWords = (u64)A.smem_cap * ( A.P.smem_compact ? 2u : 5u );
        S.stage = A.stage + lane * A.seg_cap;
        S.seg_cap = A.seg_cap;
        S.smem_a = (ma_segment*)( (u64*)A.smem_a + lane * smemWords );
        S.smem_b = (ma_segment*)( (u64*)A.smem_b + lane * smemWords );
        S.smem_cap = A.smem_cap;
    }
    __shared__ ulonglong2 sHeads[ SM ? 2 * MA_SMEM_LDS_HEADS_TASK * 256 : 1 ]; // the heads of the two SMEM lists (seeding.h: SeedScratch::lds)
    if( SM && A.P.smem_compact )
    {
        S.lds = (u64*)sHeads + 2 * threadIdx.x;
        S.lds_n = A.smem_cap < MA_SMEM_LDS_HEADS_TASK ? A.smem_cap : MA_SMEM_LDS_HEADS_TASK;
        S.lds_stride = 256;
    }
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
            if( SM && flush && ( L.err & ( MA_ERR_SEG_OVERFLOW | MA_ERR_SMEM_OVERFLOW ) ) )
                L.err = ( L.err & ~(u32)MA_ERR_SEG_OVERFLOW ) | MA_ERR_SMEM_OVERFLOW; // the task outgrew its staging area or a list
            const u32 ns = flush ? ( L.nseg < S.seg_cap ? L.nseg : S.seg_cap ) : 0;
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
                        A.pool[ so + k ] = SM ? S.stage[ k ] : mine[ k ];
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
        bool ext = act && seed_try<true, SM, MS>( L, A.P, c, SM ? &S : nullptr );
        {
            const unsigned long long sm = __ballot( act && !ext );
            if( sm && ( (u32)__popcll( sm ) >= A.slow_batch || __ballot( ext ) == 0 ) )
                if( act && !ext )
                    ext = seed_prepare<true, true, SM, MS>( L, A.P, S, A.X, c );
        }
        if( ext )
        {
            i64 ok[ 3 ];
            u32 nb;
            seed_prefetch<true>( L, A.P );
            extend_backward( A.X, L.ik, c, ok, nb );
            L.steps++;
            L.blocks += nb;
            seed_apply<SM, MS>( L, A.P, S, ok );
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
__global__ void __launch_bounds__( 256 ) k_seed_tasks( TaskKernelArgs A )
{
    seed_tasks_body<false>( A );
}
// 3 waves per SIMD: at 4 (128 VGPRs) the kernel spills 51 VGPRs inside the step loop -- 50 kb x 10 k reads: 439-469 ms against 368 ms
// with 168 VGPRs and no spill (profiles/r06_smem_tasks_tuning.txt)
#if !defined( MA_TASKS_SMEM_WAVES )
#define MA_TASKS_SMEM_WAVES 3
#endif
__global__ void __launch_bounds__( 256 ) __attribute__( ( amdgpu_waves_per_eu( MA_TASKS_SMEM_WAVES ) ) ) k_seed_tasks_smem( TaskKernelArgs A )
{
    seed_tasks_body<true>( A );
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
