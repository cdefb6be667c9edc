// launch_dp.h -- host side of the NeedlemanWunsch + MappingQuality stage: ma_dp_batch (enumeration, the ksw launches of
// ksw_launch.h on their streams, stitch, finish), ma_batch_set_alignments.  Textually part of pipeline.hip.
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
        b->ez.reserve( ( nSlots + 2 ) * sizeof( ma_ez ) ) || b->clsLists.reserve( ( ( KSW_N_CLASSES + 2 ) * nSlots + 2 ) * 4 ) || b->cigOff.reserve( ( nSlots + 2 ) * 8 ) ||
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
    D.SC.grp = ksw_grp_env( );
    D.SC.band_mis = ksw_band_mis_env( );
    D.SC.band_long = ksw_bandl_env( );
    // The band of 24 is for batches of short reads.  Long reads have millions of short extension jobs between their seeds whose
    // query rarely follows the main diagonal (50 kb at 10 % errors: 92 k of 11 M jobs pass the pre-filter, which then costs more than
    // the band saves: k_dp_enum 4.9 -> 20.8 ms; measured in round 5).  Their LONG extension jobs go to the band of 120 (band_long).
    if( D.SC.grp >= 1000 && b->max_qlen > 1000 )
        D.SC.grp = 1;
    D.ez = b->ez.as<ma_ez>( );
    D.cig_off = b->cigOff.as<u64>( );
    if( b->sortKey2.reserve( nSlots + 64 ) ) // (free until the job lists are sorted)
        return 1;
    D.cls_cache = b->sortKey2.as<uint8_t>( );
    {
        // 1 x 1 gap fills are answered by the enumeration when the worst score cannot lose to a gap (stage_dp.h); MA_DP_1X1=0: A/B hook
        int8_t q = (int8_t)b->P.gap, e = (int8_t)b->P.extend, q2 = (int8_t)b->P.gap2, e2 = (int8_t)b->P.extend2;
        if( q2 + e2 < q + e )
            std::swap( q, q2 ), std::swap( e, e2 );
        const int mis = -std::abs( (int)(int8_t)b->P.mismatch ), worst = std::min( mis, -(int)e2 );
        const int alt = std::max( -2 * ( q + e ), -( q2 + e2 ) - ( q + e ) );
        const bool untouched = -std::min( mis, 0 ) > 2 * ( q + e ); // kswcpp returns at once (kswcpp_core.h:340-341)
        const char* env = getenv( "MA_DP_1X1" );
        D.one_by_one = ( !env || atoi( env ) != 0 ) && !untouched && worst >= alt && q >= 0 && e >= 1 && q2 >= 0 && e2 >= 1 ? 1u : 0u;
    }
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
        if( getenv( "MA_DP_CLASS_REPORT" ) ) // diagnostics: jobs per kernel class of this DP stage
        {
            fprintf( stderr, "dp classes:" );
            for( int k = 0; k < KSW_N_CLASSES; k++ )
                fprintf( stderr, " %llu", (unsigned long long)S.cls[ k ] );
            fprintf( stderr, "\n" );
        }
        S.bandlN = b->hctr[ CTR_MAX_BANDL ];
        S.pRedo = b->hctr[ CTR_MAX_P_REDO ];
        S.cigRedo = b->hctr[ CTR_MAX_CIG_REDO ];
        // every wave of the ksw launches may leave one partly used 4096-word reservation per class launch
        b->cigPoolCap = std::max<u64>( 64 * nJobs + ( 1 << 20 ), b->n_bases / 2 ) + 4096ull * 256 * 32 * 4;
        b->cigPoolCap = std::max( b->cigPoolCap, b->cigPoolMin );
        if( const char* e = getenv( "MA_CIG_POOL_CAP" ) ) // test hook: force a (too) small pool on the first attempt
            if( b->cigPoolMin == 0 )
                b->cigPoolCap = (u64)std::max( 1, atoi( e ) );
        const KswScoring SC = D.SC;
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
            // pool word 0 = the cigar 1M of the 1 x 1 gap fills the enumeration answered; the kernels' reservations start at 1
            MA_HIP( hipMemsetD32Async( (hipDeviceptr_t)b->cigPool.p, (int)( 1u << 4 ), 1, b->stream ) );
            MA_HIP( hipMemsetD32Async( (hipDeviceptr_t)( c + CTR_CIG_USED ), 1, 1, b->stream ) );
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
                hipStream_t dpStream = b->stream;
#if defined( MA_EXP_DP_PRIO )
                // Experiment (round 6): with several batches in flight the persistent DP waves of one batch keep another batch's seeding
                // kernel out of the SIMDs.  On a stream of the lowest priority the dispatcher should prefer the other batches' kernels
                // whenever a DP launch ends and wave slots come free.
                if( !b->dpLow )
                {
                    int lo = 0, hi = 0;
                    MA_HIP( hipDeviceGetStreamPriorityRange( &lo, &hi ) );
                    MA_HIP( hipStreamCreateWithPriority( &b->dpLow, hipStreamNonBlocking, lo ) );
                    MA_HIP( hipEventCreateWithFlags( &b->dpFork, hipEventDisableTiming ) );
                    MA_HIP( hipEventCreateWithFlags( &b->dpJoin, hipEventDisableTiming ) );
                }
                dpStream = b->dpLow;
                MA_HIP( hipEventRecord( b->dpFork, b->stream ) );
                MA_HIP( hipStreamWaitEvent( dpStream, b->dpFork, 0 ) );
                struct Rejoin
                {
                    ma_batch* b;
                    ~Rejoin( )
                    {
                        (void)hipEventRecord( b->dpJoin, b->dpLow );
                        (void)hipStreamWaitEvent( b->stream, b->dpJoin, 0 );
                    }
                } xRejoin{ b };
#endif
                // long reads: every kernel class on its own stream (ksw_launch.h), longest jobs first
                const bool longReads = b->max_qlen > 254 && !dp_one_stream( );
                if( longReads && !b->kswSide.ready( ) )
                {
                    MA_HIP( hipEventCreateWithFlags( &b->kswSide.fork, hipEventDisableTiming ) );
                    MA_HIP( hipEventCreateWithFlags( &b->kswSide.band, hipEventDisableTiming ) );
                    for( int l = 0; l < 3; l++ )
                    {
                        MA_HIP( hipStreamCreateWithFlags( &b->kswSide.stream[ l ], hipStreamNonBlocking ) );
                        MA_HIP( hipEventCreateWithFlags( &b->kswSide.join[ l ], hipEventDisableTiming ) );
                    }
                }
                if( longReads )
                    for( int k : { 0, 1, 2, 3, KSW_CLS_BANDL, KSW_CLS_BANDL + 1 } )
                    {
                        const u64 nk = S.cls[ k ];
                        if( nk < 2048 )
                            continue;
                        u32* list = b->clsLists.as<u32>( ) + (u64)k * nSlots;
                        if( b->sortKey.reserve( nk * 4 ) || b->sortKey2.reserve( nk * 4 ) || b->sortVal2.reserve( nk * 4 ) )
                            return 1;
                        hipLaunchKernelGGL( k_job_cost, dim3( (unsigned)( ( nk + 255 ) / 256 ) ), dim3( 256 ), 0, dpStream, F, list, (u32)nk,
                                            b->sortKey.as<u32>( ) );
                        size_t tb = 0;
                        MA_HIP( hipcub::DeviceRadixSort::SortPairsDescending( nullptr, tb, b->sortKey.as<u32>( ), b->sortKey2.as<u32>( ), list,
                                                                              b->sortVal2.as<u32>( ), (int)nk, 0, 32, dpStream ) );
                        if( b->cubTmp.reserve( tb + 256 ) )
                            return 1;
                        MA_HIP( hipcub::DeviceRadixSort::SortPairsDescending( b->cubTmp.p, tb, b->sortKey.as<u32>( ), b->sortKey2.as<u32>( ), list,
                                                                              b->sortVal2.as<u32>( ), (int)nk, 0, 32, dpStream ) );
                        MA_HIP( hipMemcpyAsync( list, b->sortVal2.p, nk * 4, hipMemcpyDeviceToDevice, dpStream ) );
                    }
                // the wavefront-sharing classes: sets of about equally long jobs (MA_KSW_GRP_SORT=0: A/B hook)
                if( []( ) { const char* e = getenv( "MA_KSW_GRP_SORT" ); return !e || atoi( e ) != 0; }( ) )
                {
                    GrpSortArgs G;
                    u64 most = 0, all = 0;
                    for( int k = 0; k < KSW_GRP_SORT_LISTS; k++ ) // (qlen <= 254 here: a bin per length)
                    {
                        const int cls = KSW_CLS_GRP0 + k;
                        G.n[ k ] = (u32)S.cls[ cls ];
                        G.list[ k ] = b->clsLists.as<u32>( ) + (u64)cls * nSlots;
                        most = std::max<u64>( most, S.cls[ cls ] );
                        all += S.cls[ cls ];
                    }
                    if( all >= 8192 )
                    {
                        if( b->sortVal2.reserve( all * 4 + 64 ) || b->sortKey.reserve( KSW_GRP_SORT_LISTS * 2 * KSW_GRP_BINS * 4 ) )
                            return 1;
                        u64 at = 0;
                        for( int k = 0; k < KSW_GRP_SORT_LISTS; k++ )
                        {
                            G.tmp[ k ] = b->sortVal2.as<u32>( ) + at;
                            at += G.n[ k ];
                        }
                        G.hist = b->sortKey.as<u32>( );
                        MA_HIP( hipMemsetAsync( G.hist, 0, KSW_GRP_SORT_LISTS * 2 * KSW_GRP_BINS * 4, dpStream ) );
                        const dim3 grid( (unsigned)std::min<u64>( 1024, ( most + 255 ) / 256 ), KSW_GRP_SORT_LISTS );
                        hipLaunchKernelGGL( k_grp_hist, grid, dim3( 256 ), 0, dpStream, F, G );
                        hipLaunchKernelGGL( k_grp_scatter, grid, dim3( 256 ), 0, dpStream, F, G );
                    }
                }
                if( ksw_run_all( F, SC, (u32)nSlots, S, b->kswScratch, (unsigned int*)( c + CTR_NEXT_SLOTS ), O, dpStream,
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
            // (the result records are NOT cleared: every listed job writes its own again, and the records of the 1 x 1 gap
            // fills, which the enumeration wrote, must stay)
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
