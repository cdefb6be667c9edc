// launch_chain.h -- host side of the SoC sweep + Harmonization: ma_chain_batch (C ABI), thin waves for small batches, the
// four-launch form for long reads.  Textually part of pipeline.hip.
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
            const u32 ldsSmall = (u32)ws::scratch_bytes( wsSmall );
            for( int mode = 0; mode < 2; mode++ )
            {
                // reads of up to MA_WSORT_SMALL seeds with their arrays in LDS, then the larger ones with their arrays in global memory
                // (setB: free until k_chain); reads of more than MA_WSORT_HUGE seeds are copied by that launch and sorted by their lane
                hipLaunchKernelGGL( k_sort_seeds_wave, dim3( (unsigned)n ), dim3( 64 ), ldsSmall, b->stream, (u32)n, A.seed_off, A.seed_cnt, A.seeds,
                                    A.work, A.setA, b->preSorted.as<u32>( ), mode, 0u, wsSmall, wsMin, wsSmall, (ma_seed*)nullptr );
                hipLaunchKernelGGL( k_sort_seeds_wave, dim3( (unsigned)n ), dim3( 64 ), 0, b->stream, (u32)n, A.seed_off, A.seed_cnt, A.seeds, A.work,
                                    A.setA, b->preSorted.as<u32>( ), mode, wsSmall + 1, 0xffffffffu, std::max( wsSmall + 1, wsMin ), MA_WSORT_HUGE, A.setB );
                if( mode == 0 )
                {
                    // the sweep: one wavefront per read the kernels above sorted, one lane per read for the others
                    // (MA_SOC_WAVE=0: all by their lane; 2: the wave kernel's fallback to the lane form on every read -- tests)
                    const int waveSweep = []( ) { const char* e = getenv( "MA_SOC_WAVE" ); return e ? atoi( e ) : 1; }( );
                    if( waveSweep )
                        hipLaunchKernelGGL( k_soc_windows_wave, dim3( (unsigned)n ), dim3( 64 ), 0, b->stream, A.X, A.P, (u32)n, A.roff, A.seed_off,
                                            A.seed_cnt, A.work, A.maxima, A.mm, A.setA, A.setB, b->preSorted.as<u32>( ), b->preNmx.as<u32>( ),
                                            waveSweep == 2 ? 1 : 0 );
                    hipLaunchKernelGGL( k_soc_windows, dim3( (unsigned)( ( n + A.lanes - 1 ) / A.lanes ) ), dim3( 64 ), 0, b->stream, A.X, A.P, (u32)n,
                                        A.lanes, A.roff, A.seed_off, A.seed_cnt, A.work, A.maxima, A.mm, A.setA, b->preSorted.as<u32>( ),
                                        b->preNmx.as<u32>( ), waveSweep );
                }
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
