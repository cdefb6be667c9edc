// launch_seed.h -- host side of the seeding stage: ma_seed_batch (C ABI) with its three launch paths (read per lane, MEMs, area
// tasks), pool sizing and counted retries.  Textually part of pipeline.hip (inside its extern "C" block).
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

// maxSpan / SMEM seeding of long reads as area tasks (k_seed_tasks, k_seed_tasks_smem); returns 2 when the task arrays -- or, SMEMs,
// a lane's staging area or pending lists -- were too small (the caller falls back to the read-per-lane kernel)
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
    const bool smem = b->P.seeding_technique == 1;
    const u64 taskCap = nb / 16 + 2 * n + 1024;
    b->segPoolCap = std::max( ( smem ? 4 : 1 ) * seg_pool_heuristic( nb, n ), b->segPoolMin );
    // SMEM tasks: what ONE centre needs, not what a read needs -- a centre's forward run pushes an entry per change of the interval's
    // size (a few dozen before the match ends; a read-per-lane list is sized for the read's length) and emits a handful of segments;
    // MA_SEED_TASK_CAPS="<segments>,<list entries>": test hook that forces the fallback
    u32 segCapT = 96, smemCapT = 384;
    if( const char* e = getenv( "MA_SEED_TASK_CAPS" ) )
    {
        int a = 0, c = 0;
        if( sscanf( e, "%d,%d", &a, &c ) == 2 && a > 0 && c > 0 )
            segCapT = (u32)a, smemCapT = (u32)( ( c + 1 ) & ~1 );
    }
    const u32 smem_compact = smem && b->max_qlen < ( 1u << MA_SMEM_QZ_BITS ) && b->idx->v.n < ( 1ull << 35 ) && !( getenv( "MA_SMEM_COMPACT" ) && atoi( getenv( "MA_SMEM_COMPACT" ) ) == 0 ) ? 1 : 0;
    const u64 smemEntry = smem_compact ? 16 : sizeof( ma_segment );
    unsigned blocksT = smem ? 1024 : 2048; // (the SMEM kernel: 4 waves per SIMD)
    if( const char* e = getenv( "MA_SEED_TASK_BLOCKS" ) ) // tuning hook
        blocksT = (unsigned)std::max( 64, std::min( 4096, atoi( e ) ) );
    const u64 lanesT = (u64)blocksT * 256;
    if( smem && ( b->taskStage.reserve( lanesT * segCapT * sizeof( ma_segment ) ) || b->smemA.reserve( lanesT * smemCapT * smemEntry ) ||
                  b->smemB.reserve( lanesT * smemCapT * smemEntry ) ) )
        return 1;
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
    A.stage = smem ? b->taskStage.as<ma_segment>( ) : nullptr;
    A.seg_cap = segCapT;
    A.smem_a = smem ? b->smemA.as<ma_segment>( ) : nullptr;
    A.smem_b = smem ? b->smemB.as<ma_segment>( ) : nullptr;
    A.smem_cap = smemCapT;
    if( smem )
    {
        A.P.smem_compact = smem_compact;
        A.P.smem_merge = A.P.min_amb == 0 ? 1 : 0;
        if( const char* e = getenv( "MA_SMEM_MERGE" ) ) // test hook: 0 = keep every entry like the reference's lists
            A.P.smem_merge = A.P.smem_merge && atoi( e ) != 0 ? 1 : 0;
        A.X.kmer_k = 0; // (the K-mer table serves the maxSpan runs that start from a single base)
        A.slow_batch = 8;
        if( const char* e = getenv( "MA_SEED_SLOW_BATCH" ) )
            A.slow_batch = (u32)std::max( 1, atoi( e ) );
    }
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
            if( smem )
                hipLaunchKernelGGL( k_seed_tasks_smem, dim3( blocksT ), dim3( 256 ), 0, b->stream, A );
            else
                hipLaunchKernelGGL( k_seed_tasks, dim3( blocksT ), dim3( 256 ), 0, b->stream, A );
        }
    }
    MA_HIP( hipGetLastError( ) );
    if( read_ctr( b ) )
        return 1;
    const u32 err = (u32)b->hctr[ CTR_ERR ];
    const u64 ns = b->hctr[ CTR_SEG_USED ];
    if( ( err & ( MA_ERR_STACK_OVERFLOW | MA_ERR_SMEM_OVERFLOW ) ) )
    {
        MA_HIP( hipMemsetAsync( b->ctr.p, 0, CTR_COUNT * 8, b->stream ) );
        return 2; // task array (SMEMs: a lane's staging area or lists) too small: the classic kernel takes over
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
    // walk pays a tail per level), with 20 k reads of 50 kb the task kernel is 6.5x faster (83 vs 546 ms).  100 k reads of 10 kb
    // (`pacbio` parameter set) are past the crossover as well: 87 ms per lane, 98 ms as tasks -- the limit is 64 k reads.
    // MA_SEED_TASKS=0 / 1 forces the choice (tests, tuning)
    {
        // Round 6: SMEM seeding (Nanopore preset) as tasks too -- one read per lane left 10 k x 50 kb reads on 10 k of 262 k lanes
        bool tasks = b->P.seeding_technique <= 1 && b->max_qlen > 240 && n < 65536;
        if( const char* e = getenv( "MA_SEED_TASKS" ) )
            tasks = b->P.seeding_technique <= 1 && atoi( e ) != 0;
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
    // SMEM pending lists as 16-byte entries (seeding.h: smem_pack) unless the text or a read is too long for the packed fields
    u32 smem_compact = smem && b->max_qlen < ( 1u << MA_SMEM_QZ_BITS ) && b->idx->v.n < ( 1ull << 35 ) ? 1 : 0;
    if( const char* e = getenv( "MA_SMEM_COMPACT" ) ) // tuning / test hook
        smem_compact = smem_compact && atoi( e ) != 0 ? 1 : 0;
    const u64 smem_entry = smem_compact ? 16 : sizeof( ma_segment );
    u32 seg_cap = worst_cap;
    if( want * ( (u64)worst_cap * sizeof( ma_segment ) + 2ull * smem_cap * smem_entry ) > budget )
        seg_cap = std::min<u32>( worst_cap, ( smem ? 3 : 1 ) * ( b->max_qlen / 4 ) + 64 );
    if( const char* e = getenv( "MA_SEED_STAGE_CAP" ) ) // test hook: force a (too) small first attempt
        seg_cap = std::min<u32>( worst_cap, (u32)std::max( 1, atoi( e ) ) );
    for( int attempt = 0; attempt < 3; attempt++ )
    {
        const u64 lane_bytes = (u64)seg_cap * sizeof( ma_segment ) + 2ull * smem_cap * smem_entry;
        const u64 lanes = std::min<u64>( want, std::max<u64>( 256, ( budget / lane_bytes ) / 256 * 256 ) );
        if( b->stage.reserve( lanes * seg_cap * sizeof( ma_segment ) ) ||
            ( smem && ( b->smemA.reserve( lanes * smem_cap * smem_entry ) || b->smemB.reserve( lanes * smem_cap * smem_entry ) ) ) ||
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
            A.P.smem_compact = smem_compact;
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
