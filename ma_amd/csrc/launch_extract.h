// launch_extract.h -- host side of seed extraction: ma_extract_seeds_batch (C ABI).  Textually part of pipeline.hip.
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
