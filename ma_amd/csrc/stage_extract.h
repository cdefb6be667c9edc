// stage_extract.h -- kernels of seed extraction (SegmentVector::forEachSeed segment.h:316-349, FMIndex::bwt_sa fMIndex.h:788-814,
// ExtractSeeds stripOfConsideration.h:97-157): k_seg_seed_counts, k_read_seed_ranges, k_seed_rows, k_lf_walk, k_seed_final.
// Textually part of pipeline.hip.
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
