// stage_chain.h -- kernels of the SoC sweep and Harmonization (stripOfConsideration.cpp:12-161, harmonization.cpp:14-555): k_chain,
// the wave-cooperative sorts and the window sweep of long reads (k_sort_seeds_wave, k_soc_windows), k_soc_dump, and the
// compaction of the harmonized sets (k_hseed_counts, k_hset_flatten).  Textually part of pipeline.hip.
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
#define MA_WSORT_SMALL 1024u // reads with up to this many seeds: arrays in LDS, 15 KB per wavefront (13 n + 1.9 KB)
// More seeds, up to the 16-bit positions of wave_sort.h: the same sort on arrays in GLOBAL memory (the read's part of a scratch buffer
// the stage uses later).  Round 6: the LDS form of the large reads (up to 8192 seeds in 100 KB) ran one wavefront per CU, and half of
// the Nanopore preset's 50 kb reads (8400 seeds on average) fell to the lane-serial sort behind it; the global form runs as many
// wavefronts per CU as registers allow and is faster from about 1000 seeds on (chain stage: Nanopore preset 379 -> 141 ms,
// 50 kb reads under the default preset 113 -> 92 ms); reads of more seeds than MA_WSORT_HUGE -> the lane-serial sort of chain.h.
#define MA_WSORT_HUGE 65535u
// One wavefront per read.  mode 0: work = seeds sorted by delta (reads outside [nMin, nMax] of this launch are left alone,
// reads it owns but cannot sort are copied unsorted); mode 1: work re-sorted by reference position in place (via tmp).
// gscratch (or null): 40 n bytes per read carved by seed offset -- the arrays of a read that does not fit the LDS (13 n + 1.9 KB of it;
// the barriers of wave_sort.h order a wavefront's accesses to global memory as they order those to LDS)
__global__ void __launch_bounds__( 64 ) k_sort_seeds_wave( u32 n_reads, const u64* seed_off, const u32* seed_cnt, const ma_seed* seeds,
                                                          ma_seed* work, ma_seed* tmp, u32* sorted, int mode, u32 nMin, u32 nMax,
                                                          u32 nSortMin, u32 nSortMax, ma_seed* gscratch )
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
    ws::Scratch S = gscratch != nullptr ? ws::carve( (uint8_t*)( gscratch + off ), n ) : ws::carve( lds, doSort ? n : 1 );
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
                                                      const u32* sorted, u32* pre_nmx, int waveSweep )
{
    const u32 r = blockIdx.x * lanes + threadIdx.x;
    if( threadIdx.x >= lanes || r >= n_reads )
        return;
    const u64 off = seed_off[ r ];
    const bool byDelta = ( sorted[ r ] & 1u ) != 0;
    if( byDelta && waveSweep )
        return; // k_soc_windows_wave's
    // (tmp is the keyed sort's scratch, free when the wave-cooperative kernel did the sort: 40 n bytes for the 12 (n + 1) of the prefix
    // sums.  Reads sorted in here -- 10 kb: ~240 seeds -- gain nothing: building the sums costs what they save, 55.3 vs 57.7 ms)
    pre_nmx[ r ] = soc_windows( X, P, work + off, seed_cnt[ r ], (u32)( roff[ r + 1 ] - roff[ r ] ), maxima + off, mm + off, tmp + off,
                                byDelta, byDelta && seed_cnt[ r ] >= 2 ? (u64*)( tmp + off ) : nullptr );
}

// Round 6: the window sweep of a read whose seeds k_sort_seeds_wave sorted by delta, ONE WAVEFRONT per read (VERDICT rounds 3-5: "the
// chain stage wave-cooperatively" -- this is its sweep; the sorts are wave_sort.h, harmonization stays one read per lane).  The lane
// form (soc_windows, chain.h) pays five to eight dependent loads from global memory per seed: 6300 cycles per seed on the 8400 seeds
// of a 50 kb Nanopore read.  Here:
//   1. all lanes: contig id of every seed, prefix sums of (len, ambiguity) by wave scans -- the same integers as the lane form's;
//   2. all lanes: the window end of every S by binary search.  The reference's end pointer never moves back
//      (stripOfConsideration.cpp:77-113): E(S) = first e >= E(S-1) with delta[e] > delta[S] + strip or contig[e] != contig[S].  With the
//      seeds sorted by delta and contig ids that do not decrease along them (checked; otherwise lane 0 runs the lane form), the
//      condition is monotone in e for every S and in S for every e, so E(S) is the first failing e behind S;
//   3. the strip stack of push_back_no_overlap (soc.h:362-404), which is serial: every lane runs it redundantly on wave-uniform values,
//      64 windows at a time read from registers (readlane), the top two strips of the stack in registers with the prefix sums at
//      their ends -- the re-sums of the reference's loops are differences of those -- and lane 0 writes what changes;
//   4. make_heap by lane 0 (few strips), the reference rectangles of the strips by all lanes.
// scr: 16 n bytes of the read's part of a free scratch buffer (delta | contig id | window end); tmp: the prefix sums, as in soc_windows.
__device__ __forceinline__ u64 wave_uniform_u64( u64 v, int l )
{
    return ( (u64)(u32)__builtin_amdgcn_readlane( (int)( v >> 32 ), l ) << 32 ) | (u32)__builtin_amdgcn_readlane( (int)(u32)v, l );
}
struct SoCTop // a strip of the stack in registers, with the prefix sums at its ends
{
    SoCEntry e;
    u64 plb, ple;
    u32 pab, pae;
};
__global__ void __launch_bounds__( 64 ) k_soc_windows_wave( IndexView X, ChainParams P, u32 n_reads, const u64* roff, const u64* seed_off,
                                                           const u32* seed_cnt, ma_seed* work, SoCEntry* maxima, RefMinMax* mm, ma_seed* tmp,
                                                           ma_seed* scr, const u32* sorted, u32* pre_nmx, int forceLaneForm )
{
    const u32 r = blockIdx.x;
    if( r >= n_reads || ( sorted[ r ] & 1u ) == 0 )
        return;
    const int lane = threadIdx.x & 63;
    const u64 off = seed_off[ r ];
    const u32 n = seed_cnt[ r ];
    const u32 qlen = (u32)( roff[ r + 1 ] - roff[ r ] );
    const ma_seed* s = work + off;
    SoCEntry* mx = maxima + off;
    if( n == 0 )
    {
        if( lane == 0 )
            pre_nmx[ r ] = 0;
        return;
    }
    u64* pl = (u64*)( tmp + off ); // n + 1
    u32* pa = (u32*)( pl + n + 1 ); // n + 1
    u64* dl = (u64*)( scr + off ); // n
    u32* cid = (u32*)( dl + n ); // n
    u32* winE = cid + n; // n
    double fMinLen = mmax( (double)P.harm_score_min_rel * (double)(u64)qlen, (double)(u64)P.harm_score_min );
    if( P.genome_size_disable >= X.n )
        fMinLen = 0;
    const u64 strip = P.soc_width != 0 ? (u64)P.soc_width : ( (u64)P.match * (u64)qlen - (u64)P.gap ) / (u64)P.extend;
    const u64 minScore = (u64)fMinLen;
    // ---- 1. contig ids, prefix sums
    bool monotone = true;
    {
        SeqIdCache cache;
        u64 carryL = 0;
        u32 carryA = 0, prevCid = 0;
        if( lane == 0 )
            pl[ 0 ] = 0, pa[ 0 ] = 0;
        for( u32 base = 0; base < n; base += 64 )
        {
            const u32 i = base + lane;
            u64 vl = 0;
            u32 va = 0, c = 0;
            if( i < n )
            {
                vl = (u64)s[ i ].len, va = s[ i ].ambiguity;
                c = (u32)cache.get( X, (u64)s[ i ].r_start );
                dl[ i ] = (u64)s[ i ].delta;
                cid[ i ] = c;
            }
            const u32 left = (u32)__shfl_up( (int)c, 1, 64 );
            if( i < n && c < ( lane == 0 ? prevCid : left ) )
                monotone = false;
            prevCid = (u32)__shfl( (int)c, 63, 64 ); // (lanes past n hold 0, but then this was the last chunk)
            for( int d = 1; d < 64; d <<= 1 )
            {
                const u64 ol = ( (u64)(u32)__shfl_up( (int)( vl >> 32 ), d, 64 ) << 32 ) | (u32)__shfl_up( (int)(u32)vl, d, 64 );
                const u32 oa = (u32)__shfl_up( (int)va, d, 64 );
                if( lane >= d )
                    vl += ol, va += oa;
            }
            vl += carryL, va += carryA;
            if( i < n )
                pl[ i + 1 ] = vl, pa[ i + 1 ] = va;
            carryL = wave_uniform_u64( vl, 63 );
            carryA = (u32)__builtin_amdgcn_readlane( (int)va, 63 );
        }
    }
    __syncthreads( );
    if( __ballot( !monotone ) != 0 || forceLaneForm )
    {
        // contig ids that go down along the deltas (seeds around a contig border): the lane form, which takes them one by one
        if( lane == 0 )
            pre_nmx[ r ] = soc_windows( X, P, work + off, n, qlen, mx, mm + off, tmp + off, true, n >= 2 ? (u64*)( tmp + off ) : nullptr );
        return;
    }
    // ---- 2. window ends
    for( u32 S = lane; S < n; S += 64 )
    {
        const u64 lim = dl[ S ] + strip;
        const u32 c = cid[ S ];
        u32 lo = S + 1, hi = n;
        while( lo < hi )
        {
            const u32 mid = lo + ( hi - lo ) / 2;
            if( dl[ mid ] > lim || cid[ mid ] != c )
                hi = mid;
            else
                lo = mid + 1;
        }
        winE[ S ] = lo;
    }
    __syncthreads( );
    // ---- 3. the strip stack (wave-uniform)
    u32 nmx = 0;
    SoCTop top, second;
    bool haveSecond = false;
    top.e.accLen = 0, top.e.amb = top.e.cnt = top.e.b = top.e.e = 0, top.plb = top.ple = 0, top.pab = top.pae = 0;
    second = top;
    bool more = true;
    for( u32 S0 = 0; S0 < n && more; S0 += 64 )
    {
        const u32 Sl = S0 + lane;
        u32 vE = 0, vPaS = 0, vPaE = 0;
        u64 vPlS = 0, vPlE = 0;
        if( Sl < n )
        {
            vE = winE[ Sl ];
            vPlS = pl[ Sl ], vPaS = pa[ Sl ];
            vPlE = pl[ vE ], vPaE = pa[ vE ];
        }
        const int cnt = (int)( n - S0 < 64u ? n - S0 : 64u );
        for( int l = 0; l < cnt && more; l++ )
        {
            const u32 S = S0 + (u32)l;
            const u32 E = (u32)__builtin_amdgcn_readlane( (int)vE, l );
            const u64 plS = wave_uniform_u64( vPlS, l ), plE = wave_uniform_u64( vPlE, l );
            const u32 paS = (u32)__builtin_amdgcn_readlane( (int)vPaS, l ), paE = (u32)__builtin_amdgcn_readlane( (int)vPaE, l );
            more = E != n; // the reference's loop ends with the window that reaches the last seed
            SoCEntry cur;
            cur.accLen = plE - plS, cur.amb = paE - paS, cur.cnt = E - S, cur.b = cur.e = 0;
            if( !( (double)cur.accLen >= fMinLen ) )
                continue;
            // push_back_no_overlap( cur, [S, E) )
            u32 itS = S;
            u64 plI = plS;
            u32 paI = paS;
            bool drop = false;
            while( nmx > 0 && top.e.e > itS )
            {
                if( soc_less( top.e, cur ) )
                {
                    // the strip on top keeps what lies before itS
                    if( itS < top.e.b )
                        top.e.accLen = 0, top.e.amb = 0, top.e.cnt = 0;
                    else
                        top.e.accLen = plI - top.plb, top.e.amb = paI - top.pab, top.e.cnt = itS - top.e.b;
                    top.e.e = itS;
                    top.ple = plI, top.pae = paI;
                    if( top.e.accLen < minScore || top.e.accLen == 0 )
                    {
                        nmx--;
                        if( nmx > 0 )
                        {
                            if( haveSecond )
                                top = second, haveSecond = false;
                            else
                            {
                                __syncthreads( ); // lane 0's stores
                                top.e = mx[ nmx - 1 ];
                                top.plb = pl[ top.e.b ], top.pab = pa[ top.e.b ];
                                top.ple = pl[ top.e.e ], top.pae = pa[ top.e.e ];
                            }
                        }
                    }
                    else if( lane == 0 )
                        mx[ nmx - 1 ] = top.e;
                }
                else
                {
                    // cur keeps what lies behind the strip on top
                    const u32 be = top.e.e;
                    if( E < be )
                        cur.accLen = 0, cur.amb = 0, cur.cnt = 0;
                    else
                        cur.accLen = plE - top.ple, cur.amb = paE - top.pae, cur.cnt = E - be;
                    itS = be, plI = top.ple, paI = top.pae;
                    if( cur.accLen < minScore || cur.accLen == 0 )
                    {
                        drop = true;
                        break;
                    }
                }
            }
            if( drop )
                continue;
            cur.b = itS, cur.e = E;
            if( nmx > 0 )
                second = top, haveSecond = true;
            top.e = cur, top.plb = plI, top.pab = paI, top.ple = plE, top.pae = paE;
            if( lane == 0 )
                mx[ nmx ] = cur;
            nmx++;
        }
    }
    __syncthreads( );
    // ---- 4. the heap, the reference rectangles (soc.h:196-231)
    if( lane == 0 )
        ss::make_heap( mx, (i64)nmx, SoCHeapOrder( ) );
    __syncthreads( );
    for( u32 k = 0; k < nmx; k++ )
    {
        const u32 b = mx[ k ].b, e = mx[ k ].e;
        u64 lo = (u64)s[ b ].r_start, hi = lo;
        for( u32 i = b + (u32)lane; i < e; i += 64 )
        {
            const u64 x = (u64)s[ i ].r_start;
            lo = mmin( lo, x ), hi = mmax( hi, x );
        }
        for( int d = 32; d >= 1; d >>= 1 )
        {
            const u64 ol = ( (u64)(u32)__shfl_xor( (int)( lo >> 32 ), d, 64 ) << 32 ) | (u32)__shfl_xor( (int)(u32)lo, d, 64 );
            const u64 oh = ( (u64)(u32)__shfl_xor( (int)( hi >> 32 ), d, 64 ) << 32 ) | (u32)__shfl_xor( (int)(u32)hi, d, 64 );
            lo = mmin( lo, ol ), hi = mmax( hi, oh );
        }
        if( lane == 0 )
            mm[ off + k ].lo = lo, mm[ off + k ].hi = hi;
    }
    if( lane == 0 )
        pre_nmx[ r ] = nmx;
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
