// stage_dp.h -- kernels of the NeedlemanWunsch stage around the kswcpp kernels of ksw_launch.h (needlemanWunsch.cpp:82-877,
// mappingQuality.cpp:11-131): k_dp_enum (job enumeration), k_job_cost, k_ops_caps, k_stitch / k_stitch_wave (the walk that
// assembles the alignments), k_finish (sort + MappingQuality).  Textually part of pipeline.hip.
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
    // 1 x 1 gap fills (a single mismatch between two seeds: 30 % of the DP calls of a 150 bp batch).  NeedlemanWunsch::ksw
    // (needlemanWunsch.cpp:82-169) runs kswcpp on them globally and reads the cigar only; for a 1 x 1 matrix kswcpp's first cell
    // compares the score s with the four gap terms -2(q+e), -(q2+e2)-(q+e) (x, y, x2, y2 initialised to -q-e / -q2-e2 plus the
    // first-row / first-column u, v = -q-e, kswcpp_core.h:562-585,653-699) and the left-aligned variant keeps s on ties: whenever
    // the WORST score (mismatch, or -e2 for an N) is not below them the back-trace is one M, whatever the bases.  one_by_one = that
    // holds for the scoring in use: the enumeration writes the result (ez as kswcpp leaves it for a global call, the cigar word
    // 1M = pool word 0, which the DP stage reserves) and lists no job.
    u32 one_by_one;
    ma_ez* ez;
    u64* cig_off;
    // class of every job slot as the first pass over a wave's jobs decided it (the pre-filters of the band kernels walk the job's first
    // bases: once, not once per pass -- 50 kb: k_dp_enum 26 -> 15 ms)
    uint8_t* cls_cache;
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
    // per-class scratch sizes: only the classes whose launches are sized by their jobs (the query-stationary classes from
    // KSW_CLS_GRP0 on have a fixed scratch per wave)
    u32 pcl[ KSW_CLS_GRP0 ], cgl[ KSW_CLS_GRP0 ], pRedo = 0, cgRedo = 0, bandlN = 0;
    // Append the jobs to the per-class lists with ONE round trip to the list counters per wave: a first pass over the lanes'
    // jobs counts the wave's jobs per class (lane c holds class c's count), lane c reserves class c's list space, a second
    // pass writes the entries.  (A reservation per class and round -- each waiting for its atomic's return -- made the kernel
    // latency-bound when the classes went from 7 to 13: 1.35 -> 2.4 ms per 1 M reads.)
    {
        const u32 mine = sink.n < sink.cap ? sink.n : sink.cap;
        const u32 rounds = (u32)wave_max_u64( mine );
        const int lane = threadIdx.x & 63;
        // scratch per wave of each class's launch: the classes differ by orders of magnitude (a 50 kb end extension
        // needs 27 MB of direction bytes, a gap between two seeds a few KB), and a launch sized for the largest job of
        // the whole batch would leave most of the machine without waves
#pragma unroll
        for( int c = 0; c < KSW_CLS_GRP0; c++ )
            pcl[ c ] = cgl[ c ] = 0;
        auto classOf = [ & ]( u32 k, u32& pj, u32& cj, u32& pk8, bool second ) -> int {
            const DpJob& j = A.jobs[ sink.slot0 + k ];
            const i32 ql = (i32)( j.q_to - j.q_from ), tl = (i32)( j.r_to - j.r_from );
            pj = cj = pk8 = 0;
            if( A.one_by_one && ql == 1 && tl == 1 && j.flag == 0 && j.zdrop < 0 )
                return -2; // answered here (first pass)
            int cls = second ? (int)A.cls_cache[ sink.slot0 + k ] : ksw_job_class_pipe( A.SC, ql, tl, j.w, j.zdrop, j.flag );
            if( !second && A.SC.grp >= 1000 && ( cls == KSW_CLS_GRP0 || cls == KSW_CLS_GRP0 + 1 ) )
            {
                // the proven narrow band (ksw_band.h) is tried on the jobs whose query follows the target's main diagonal
                const uint8_t* qb = A.reads + j.read_off;
                auto qf = [ & ]( i32 i ) -> u32 { return j.rev ? qb[ j.q_to - 1 - (u32)i ] : qb[ j.q_from + (u32)i ]; };
                auto tf = [ & ]( i32 i ) -> u32 { return text_base( A.X, j.win_begin + ( j.rev ? j.r_to - 1 - (u32)i : j.r_from + (u32)i ) ); };
                if( !ksw_band_likely( qf, tf, ql, tl, A.SC.band_mis ) )
                {
                    KswScoring S1 = A.SC;
                    S1.grp = 1;
                    cls = ksw_job_class_pipe( S1, ql, tl, j.w, j.zdrop, j.flag );
                }
            }
            if( !second && ( cls == KSW_CLS_BANDL || cls == KSW_CLS_BANDL + 1 ) )
            {
                // the band of 120 is tried on the long extensions whose first bases follow the target with few edits (ksw_bandl_likely)
                const uint8_t* qb = A.reads + j.read_off;
                auto qf = [ & ]( i32 i ) -> u32 { return j.rev ? qb[ j.q_to - 1 - (u32)i ] : qb[ j.q_from + (u32)i ]; };
                auto tf = [ & ]( i32 i ) -> u32 { return text_base( A.X, j.win_begin + ( j.rev ? j.r_to - 1 - (u32)i : j.r_from + (u32)i ) ); };
                if( !ksw_bandl_likely( qf, tf, ql, tl ) )
                {
                    KswScoring S1 = A.SC;
                    S1.band_long = 0;
                    cls = ksw_job_class_pipe( S1, ql, tl, j.w, j.zdrop, j.flag );
                }
            }
            if( !second )
                A.cls_cache[ sink.slot0 + k ] = (uint8_t)cls;
            const u64 pk = ksw_p_bytes( ql, tl, j.w );
            // 256-byte units (the query-stationary classes from KSW_CLS_GRP0 on have a fixed scratch per wave: ksw_grp.h)
            pj = cls >= KSW_CLS_GRP0 ? 0u : (u32)( ( ( cls >= 5 ? ksw_ext_p_bytes( ql, tl, cls - 4 ) : pk ) + 255 ) >> 8 );
            cj = (u32)( ql + tl + 2 );
            pk8 = (u32)( ( pk + 255 ) >> 8 );
            return cls;
        };
        u32 cntV = 0; // lane c: jobs of class c among this wave's
        u32 n11 = 0; // this lane's 1 x 1 gap fills
        for( u32 k = 0; k < rounds; k++ )
        {
            int cls = -1;
            u32 pj = 0, cj = 0, pk8 = 0;
            if( k < mine )
            {
                cls = classOf( k, pj, cj, pk8, false );
                if( cls >= 5 )
                {
                    pRedo = max( pRedo, pk8 );
                    cgRedo = max( cgRedo, cj );
                }
                if( cls == KSW_CLS_BANDL || cls == KSW_CLS_BANDL + 1 )
                {
                    const DpJob& jj = A.jobs[ sink.slot0 + k ];
                    bandlN = max( bandlN, min( jj.q_to - jj.q_from, jj.r_to - jj.r_from ) );
                }
                if( A.SC.grp >= 1000 && ( cls == KSW_CLS_GRP0 || cls == KSW_CLS_GRP0 + 1 ) )
                {
                    // a job on the narrow band that fails its checks goes on to k_ksw_ext<1> / <2>: their launches are sized for it too
                    const DpJob& jj = A.jobs[ sink.slot0 + k ];
                    const i32 ql = (i32)( jj.q_to - jj.q_from ), tl = (i32)( jj.r_to - jj.r_from );
                    const int e = ksw_ext_slots( A.SC, ql, tl, jj.w, jj.zdrop, jj.flag );
                    const u32 pe = (u32)( ( ksw_ext_p_bytes( ql, tl, e ) + 255 ) >> 8 );
#pragma unroll
                    for( int c = 5; c < 7; c++ )
                        if( c == 4 + e )
                        {
                            pcl[ c ] = max( pcl[ c ], pe );
                            cgl[ c ] = max( cgl[ c ], cj );
                        }
                }
                if( cls == -2 )
                {
                    ma_ez rz; // what a global kswcpp call leaves (kswcpp_core.h:328-338, 796-835): only the cigar is set
                    rz.max = 0, rz.zdropped = 0, rz.max_q = rz.max_t = rz.mqe_t = rz.mte_q = -1;
                    rz.mqe = rz.mte = rz.score = (i32)0x80000000;
                    rz.reach_end = 0, rz.n_cigar = 1;
                    A.ez[ sink.slot0 + k ] = rz;
                    A.cig_off[ sink.slot0 + k ] = 0; // pool word 0 = 1M
                    n11++;
                }
            }
#pragma unroll
            for( int c = 0; c < KSW_CLS_GRP0; c++ )
                if( cls == c )
                {
                    pcl[ c ] = max( pcl[ c ], pj );
                    cgl[ c ] = max( cgl[ c ], cj );
                }
            unsigned long long todo = __ballot( cls >= 0 );
            while( todo )
            {
                const int c = __builtin_amdgcn_readlane( cls, __ffsll( (long long)todo ) - 1 );
                const unsigned long long m = __ballot( cls == c );
                todo &= ~m;
                if( lane == c )
                    cntV += (u32)__popcll( m );
            }
        }
        {
            const u64 all11 = wave_sum_u64( n11 );
            if( lane == 0 && all11 )
                atomicAdd( &A.ctr[ CTR_N_1X1 ], (unsigned long long)all11 );
        }
        unsigned long long baseV = 0;
        if( lane < KSW_N_CLASSES && cntV )
            baseV = atomicAdd( &A.ctr[ CTR_CLS0 + lane ], (unsigned long long)cntV );
        u32 runV = 0; // lane c: entries of class c written so far
        for( u32 k = 0; k < rounds; k++ )
        {
            int cls = -1;
            u32 pj, cj, pk8;
            if( k < mine )
                cls = classOf( k, pj, cj, pk8, true );
            unsigned long long todo = __ballot( cls >= 0 );
            while( todo )
            {
                const int c = __builtin_amdgcn_readlane( cls, __ffsll( (long long)todo ) - 1 );
                const unsigned long long m = __ballot( cls == c );
                todo &= ~m;
                const u64 base = ( (u64)(u32)__builtin_amdgcn_readlane( (int)( baseV >> 32 ), c ) << 32 ) | (u32)__builtin_amdgcn_readlane( (int)(u32)baseV, c );
                const u32 run = (u32)__builtin_amdgcn_readlane( (int)runV, c );
                if( cls == c )
                    A.lists[ (u64)c * A.list_stride + base + run + __popcll( m & ( ( 1ull << lane ) - 1 ) ) ] = sink.slot0 + k;
                if( lane == c )
                    runV += (u32)__popcll( m );
            }
        }
    }
    // one atomic per wave and quantity instead of eight per job
    const u64 st = wave_max_u64( sink.mx_state ), h = wave_max_u64( sink.mx_h ), p = wave_max_u64( sink.mx_p );
    const u64 cg = wave_max_u64( sink.mx_cig ), ql = wave_max_u64( sink.mx_qlen );
    const u64 nj = wave_sum_u64( sink.n_jobs ), sb = wave_sum_u64( sink.seq_bytes );
    u64 pcW[ KSW_CLS_GRP0 ], cgW[ KSW_CLS_GRP0 ];
#pragma unroll
    for( int c = 0; c < KSW_CLS_GRP0; c++ )
    {
        pcW[ c ] = wave_max_u64( pcl[ c ] );
        cgW[ c ] = wave_max_u64( cgl[ c ] );
    }
    const u64 pRedoW = wave_max_u64( pRedo ), cgRedoW = wave_max_u64( cgRedo ), bandlW = wave_max_u64( bandlN );
    if( ( threadIdx.x & 63 ) == 0 && nj )
    {
        atomicMax( &A.ctr[ CTR_MAX_STATE ], (unsigned long long)st );
        atomicMax( &A.ctr[ CTR_MAX_H ], (unsigned long long)h );
        atomicMax( &A.ctr[ CTR_MAX_P ], (unsigned long long)p );
        atomicMax( &A.ctr[ CTR_MAX_CIG ], (unsigned long long)cg );
        atomicMax( &A.ctr[ CTR_MAX_QLEN ], (unsigned long long)ql );
#pragma unroll
        for( int c = 0; c < KSW_CLS_GRP0; c++ )
        {
            if( pcW[ c ] )
                atomicMax( &A.ctr[ CTR_MAX_PC0 + c ], (unsigned long long)pcW[ c ] << 8 );
            if( cgW[ c ] )
                atomicMax( &A.ctr[ CTR_MAX_CIGC0 + c ], (unsigned long long)cgW[ c ] );
        }
        if( bandlW )
            atomicMax( &A.ctr[ CTR_MAX_BANDL ], (unsigned long long)bandlW );
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

// Jobs that share a wavefront run in lock-step until the longest of them is done (ksw_grp.h): the lists of those classes are
// ordered by query length, longest first, so that the jobs of a set are about equally long (unsorted: 1.7x the diagonals four
// jobs need).  A counting sort over the <= 256 possible lengths, all lists in two launches (a radix sort per list was 45
// launches per step): k_grp_hist copies every list aside and counts its lengths, k_grp_scatter puts the entries back at
// start-of-its-length + a running index.  The order among jobs of equal length is arbitrary; no result depends on it.
#define KSW_GRP_SORT_LISTS 6
#define KSW_GRP_BINS 256 // one bin per query length: the lists of the narrow band hold queries of up to 254 bases (= the block size)
struct GrpSortArgs
{
    u32* list[ KSW_GRP_SORT_LISTS ]; // in place
    u32* tmp[ KSW_GRP_SORT_LISTS ];
    u32 n[ KSW_GRP_SORT_LISTS ];
    u32* hist; // KSW_GRP_SORT_LISTS x 2 x 256 words, zeroed: [l][0][len] = jobs of that length, [l][1][len] = running index
};
// (same-address device atomics serialise in L2 at ~9 ns each: a block counts in LDS and touches every global counter once)
__global__ void k_grp_hist( PipeFetch F, GrpSortArgs A )
{
    const int l = blockIdx.y;
    __shared__ u32 h[ KSW_GRP_BINS ];
    h[ threadIdx.x ] = 0; // (256 threads, KSW_GRP_BINS = 256)
    __syncthreads( );
    for( u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < A.n[ l ]; i += gridDim.x * blockDim.x )
    {
        const u32 slot = A.list[ l ][ i ];
        A.tmp[ l ][ i ] = slot;
        atomicAdd( &h[ min( (u32)F.view( slot ).qlen, KSW_GRP_BINS - 1u ) ], 1u );
    }
    __syncthreads( );
    if( h[ threadIdx.x ] )
        atomicAdd( A.hist + ( l * 2 + 0 ) * KSW_GRP_BINS + threadIdx.x, h[ threadIdx.x ] );
}
__global__ void k_grp_scatter( PipeFetch F, GrpSortArgs A )
{
    const int l = blockIdx.y;
    __shared__ u32 start[ KSW_GRP_BINS ]; // where this block's entries of each length go: start of the length (longest first) + the block's reservation
    __shared__ u32 h[ KSW_GRP_BINS ];
    __shared__ u32 tot[ KSW_GRP_BINS ];
    h[ threadIdx.x ] = 0;
    tot[ threadIdx.x ] = A.hist[ ( l * 2 + 0 ) * KSW_GRP_BINS + threadIdx.x ];
    __syncthreads( );
    for( u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < A.n[ l ]; i += gridDim.x * blockDim.x )
        atomicAdd( &h[ min( (u32)F.view( A.tmp[ l ][ i ] ).qlen, KSW_GRP_BINS - 1u ) ], 1u );
    __syncthreads( );
    {
        u32 before = 0;
        for( u32 k = KSW_GRP_BINS - 1; k > threadIdx.x; k-- )
            before += tot[ k ];
        start[ threadIdx.x ] = before + ( h[ threadIdx.x ] ? atomicAdd( A.hist + ( l * 2 + 1 ) * KSW_GRP_BINS + threadIdx.x, h[ threadIdx.x ] ) : 0u );
    }
    __syncthreads( );
    h[ threadIdx.x ] = 0;
    __syncthreads( );
    for( u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < A.n[ l ]; i += gridDim.x * blockDim.x )
    {
        const u32 slot = A.tmp[ l ][ i ];
        const u32 len = min( (u32)F.view( slot ).qlen, KSW_GRP_BINS - 1u );
        A.list[ l ][ start[ len ] + atomicAdd( &h[ len ], 1u ) ] = slot;
    }
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
