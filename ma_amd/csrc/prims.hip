// prims.hip -- batched primitive operations behind the C ABI (kernel-level parity seams):
// FMIndex::extend_backward (fMIndex.cpp:21-101), FMIndex::bwt_sa (fMIndex.h:788-814) and
// kswcpp_dispatch (kswcpp.h:165-190).
#include "ksw_launch.h"
#include "fm_device.h"
#include <cstring>

using namespace ma;

__global__ void k_extend( IndexView X, const i64* ik, const uint8_t* c, u64 n, i64* ok )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n )
        return;
    i64 a[ 3 ] = { ik[ 3 * i ], ik[ 3 * i + 1 ], ik[ 3 * i + 2 ] }, o[ 3 ];
    u32 nb;
    extend_backward( X, a, c[ i ], o, nb );
    ok[ 3 * i ] = o[ 0 ];
    ok[ 3 * i + 1 ] = o[ 1 ];
    ok[ 3 * i + 2 ] = o[ 2 ];
}

__global__ void k_bwt_sa( IndexView X, const i64* rows, u64 n, i64* pos )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n )
        return;
    u32 steps;
    pos[ i ] = bwt_sa( X, rows[ i ], steps );
}

extern "C" int ma_extend_backward_batch( const ma_index* x, const int64_t* ik, const uint8_t* c, uint64_t n, int64_t* ok )
{
    if( !x )
        return fail( "ma_extend_backward_batch: null index" );
    MA_BIND_DEVICE( x->device );
    if( n == 0 )
        return 0;
    DevBuf dik, dc, dok;
    if( dik.reserve( n * 24 ) || dc.reserve( n ) || dok.reserve( n * 24 ) )
        return 1;
    MA_HIP( hipMemcpy( dik.p, ik, n * 24, hipMemcpyHostToDevice ) );
    MA_HIP( hipMemcpy( dc.p, c, n, hipMemcpyHostToDevice ) );
    hipLaunchKernelGGL( k_extend, dim3( (unsigned)( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, 0, x->v, dik.as<i64>( ),
                        dc.as<uint8_t>( ), n, dok.as<i64>( ) );
    MA_HIP( hipGetLastError( ) );
    MA_HIP( hipMemcpy( ok, dok.p, n * 24, hipMemcpyDeviceToHost ) );
    return 0;
}

extern "C" int ma_bwt_sa_batch( const ma_index* x, const int64_t* rows, uint64_t n, int64_t* pos )
{
    if( !x )
        return fail( "ma_bwt_sa_batch: null index" );
    MA_BIND_DEVICE( x->device );
    if( n == 0 )
        return 0;
    DevBuf dr, dp;
    if( dr.reserve( n * 8 ) || dp.reserve( n * 8 ) )
        return 1;
    MA_HIP( hipMemcpy( dr.p, rows, n * 8, hipMemcpyHostToDevice ) );
    hipLaunchKernelGGL( k_bwt_sa, dim3( (unsigned)( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, 0, x->v, dr.as<i64>( ), n,
                        dp.as<i64>( ) );
    MA_HIP( hipGetLastError( ) );
    MA_HIP( hipMemcpy( pos, dp.p, n * 8, hipMemcpyDeviceToHost ) );
    return 0;
}

namespace
{
struct ByteFetch
{
    static const bool EARLY = false; // ma_ksw_batch returns every ez field of kswcpp_dispatch
    const ma_ksw_job* jobs;
    const uint8_t* qb;
    const uint8_t* tb;
    __device__ bool valid( u32 ) const
    {
        return true;
    }
    __device__ KswJobView view( u32 s ) const
    {
        KswJobView v;
        v.qlen = jobs[ s ].qlen;
        v.tlen = jobs[ s ].tlen;
        v.w = jobs[ s ].w;
        v.zdrop = jobs[ s ].zdrop;
        v.flag = jobs[ s ].flag;
        return v;
    }
    struct Q
    {
        const uint8_t* p;
        __device__ u32 operator( )( i32 i ) const
        {
            return p[ i ];
        }
        static const bool CLEAN = false; // bytes as given: 4 (and above) = N
        __device__ u32 pair( i32 i ) const // cells i, i + 1 (the byte arrays are padded: reading one past the end is safe)
        {
            return (u32)p[ i ] | (u32)p[ i + 1 ] << 16;
        }
    };
    __device__ Q qfetch( u32 s ) const
    {
        return Q{ qb + jobs[ s ].q_off };
    }
    __device__ Q tfetch( u32 s ) const
    {
        return Q{ tb + jobs[ s ].t_off };
    }
};
struct ByteFetchPipe : ByteFetch
{
    static const bool EARLY = true; // what NeedlemanWunsch reads: max, max_q, max_t, cigar
};
} // namespace

// FETCH::EARLY selects the pipeline semantics (extension kernel + early stop, see ksw_ext.h)
template <typename FETCH>
static int ksw_batch_impl( const ma_params* P, const ma_ksw_job* jobs, uint64_t n, const uint8_t* q_bytes, uint64_t q_len,
                           const uint8_t* t_bytes, uint64_t t_len, ma_ez* ez, uint64_t* cigar_off, uint32_t* cigar,
                           uint64_t cigar_cap )
{
    if( !P || !jobs || !ez || !cigar_off )
        return fail( "ma_ksw_batch: null argument" );
    if( n == 0 )
        return 0;
    KswScoring SC{ P->match, P->mismatch, P->gap, P->extend, P->gap2, P->extend2 };
    SC.grp = ksw_grp_env( );
    SC.band_mis = ksw_band_mis_env( );
    SC.band_long = ksw_bandl_env( );
    KswSizing S;
    for( uint64_t i = 0; i < n; i++ )
        ksw_size_job( S, jobs[ i ].qlen, jobs[ i ].tlen, jobs[ i ].w );
    // pipeline semantics: per-class job lists as k_dp_enum builds them on the device
    std::vector<u32> lists;
    DevBuf dlists;
    if( FETCH::EARLY )
    {
        for( int k = 0; k < KSW_N_CLASSES; k++ )
            S.cls[ k ] = S.pc[ k ] = S.cigc[ k ] = 0;
        lists.assign( (size_t)( KSW_N_CLASSES + 2 ) * n, 0u );
        for( uint64_t i = 0; i < n; i++ )
        {
            const i32 ql = jobs[ i ].qlen, tl = jobs[ i ].tlen;
            if( ql <= 0 || tl <= 0 )
                continue;
            int c = ksw_job_class_pipe( SC, ql, tl, jobs[ i ].w, jobs[ i ].zdrop, jobs[ i ].flag );
            if( SC.grp >= 1000 && ( c == KSW_CLS_GRP0 || c == KSW_CLS_GRP0 + 1 ) && !getenv( "MA_KSW_BAND_ALL" ) )
            {
                // (k_dp_enum's pre-filter of the narrow band; MA_KSW_BAND_ALL=1: the tests try every eligible job)
                const uint8_t *qp = q_bytes + jobs[ i ].q_off, *tp = t_bytes + jobs[ i ].t_off;
                auto qf = [ & ]( i32 k ) -> u32 { return qp[ k ]; };
                auto tf = [ & ]( i32 k ) -> u32 { return tp[ k ]; };
                if( !ksw_band_likely( qf, tf, ql, tl, SC.band_mis ) )
                {
                    KswScoring S1 = SC;
                    S1.grp = 1;
                    c = ksw_job_class_pipe( S1, ql, tl, jobs[ i ].w, jobs[ i ].zdrop, jobs[ i ].flag );
                }
            }
            if( ( c == KSW_CLS_BANDL || c == KSW_CLS_BANDL + 1 ) && !getenv( "MA_KSW_BAND_ALL" ) )
            {
                const uint8_t *qp = q_bytes + jobs[ i ].q_off, *tp = t_bytes + jobs[ i ].t_off;
                auto qf = [ & ]( i32 k ) -> u32 { return qp[ k ]; };
                auto tf = [ & ]( i32 k ) -> u32 { return tp[ k ]; };
                if( !ksw_bandl_likely( qf, tf, ql, tl ) )
                {
                    KswScoring S1 = SC;
                    S1.band_long = 0;
                    c = ksw_job_class_pipe( S1, ql, tl, jobs[ i ].w, jobs[ i ].zdrop, jobs[ i ].flag );
                }
            }
            lists[ (size_t)c * n + S.cls[ c ]++ ] = (u32)i;
            const u64 pk = ksw_p_bytes( ql, tl, jobs[ i ].w ), cg = (u64)ql + tl + 2;
            S.pc[ c ] = std::max( S.pc[ c ], c >= KSW_CLS_GRP0 ? 0 : ( c >= 5 ? ksw_ext_p_bytes( ql, tl, c - 4 ) : pk ) );
            S.cigc[ c ] = std::max( S.cigc[ c ], cg );
            if( c == KSW_CLS_BANDL || c == KSW_CLS_BANDL + 1 )
                S.bandlN = std::max<u64>( S.bandlN, (u64)std::min( ql, tl ) );
            if( c >= 5 )
            {
                S.pRedo = std::max( S.pRedo, pk );
                S.cigRedo = std::max( S.cigRedo, cg );
            }
            if( SC.grp >= 1000 && ( c == KSW_CLS_GRP0 || c == KSW_CLS_GRP0 + 1 ) )
            {
                const int e = ksw_ext_slots( SC, ql, tl, jobs[ i ].w, jobs[ i ].zdrop, jobs[ i ].flag ); // its extension kernel, should it fail its checks
                S.pc[ 4 + e ] = std::max( S.pc[ 4 + e ], ksw_ext_p_bytes( ql, tl, e ) );
                S.cigc[ 4 + e ] = std::max( S.cigc[ 4 + e ], cg );
            }
        }
        if( dlists.reserve( lists.size( ) * 4 + 16 ) )
            return 1;
        MA_HIP( hipMemcpy( dlists.p, lists.data( ), lists.size( ) * 4, hipMemcpyHostToDevice ) );
    }
    DevBuf dj, dq, dt, dez, doff, dpool, dscr, dctr;
    if( dj.reserve( n * sizeof( ma_ksw_job ) ) || dq.reserve( q_len + 16 ) || dt.reserve( t_len + 16 ) ||
        dez.reserve( n * sizeof( ma_ez ) ) || doff.reserve( ( n + 1 ) * 8 ) || dpool.reserve( cigar_cap * 4 + 16 ) ||
        dctr.reserve( 256 ) )
        return 1;
    MA_HIP( hipMemcpy( dj.p, jobs, n * sizeof( ma_ksw_job ), hipMemcpyHostToDevice ) );
    MA_HIP( hipMemcpy( dq.p, q_bytes, q_len, hipMemcpyHostToDevice ) );
    MA_HIP( hipMemcpy( dt.p, t_bytes, t_len, hipMemcpyHostToDevice ) );
    MA_HIP( hipMemset( dctr.p, 0, 256 ) );
    MA_HIP( hipMemset( dez.p, 0, n * sizeof( ma_ez ) ) );
    MA_HIP( hipMemset( doff.p, 0, ( n + 1 ) * 8 ) );
    KswOut O;
    unsigned long long* ctr = dctr.as<unsigned long long>( );
    O.ez = dez.as<ma_ez>( );
    O.cig_off = doff.as<u64>( );
    O.cig_pool = dpool.as<u32>( );
    O.cig_pool_cap = cigar_cap;
    O.cig_used = ctr + 0;
    O.cells = ctr + 1;
    O.njobs = ctr + 2;
    O.err = (u32*)( ctr + 3 );
    O.path = nullptr;
    O.cig_words = nullptr;
    O.cig_chunk = 0; // dense pool: cigar_off[n] is the total
    unsigned int* next = (unsigned int*)( ctr + 4 ); // 28 x u32 launch queues (ctr[4..17])
    FETCH F;
    F.jobs = dj.as<ma_ksw_job>( );
    F.qb = dq.as<uint8_t>( );
    F.tb = dt.as<uint8_t>( );
    if( ksw_run_all( F, SC, (u32)n, S, dscr, next, O, 0, FETCH::EARLY ? dlists.as<u32>( ) : nullptr, n,
                     (unsigned int*)( ctr + 18 ), (unsigned int*)( ctr + 19 ) ) )
        return 1;
    MA_HIP( hipDeviceSynchronize( ) );
    unsigned long long h[ 8 ];
    MA_HIP( hipMemcpy( h, dctr.p, 64, hipMemcpyDeviceToHost ) );
    int rc = 0;
    if( ( (u32)h[ 3 ] ) & MA_ERR_CIGAR_OVERFLOW )
        rc = fail( "ma_ksw_batch: cigar capacity too small" );
    else
    {
        MA_HIP( hipMemcpy( ez, dez.p, n * sizeof( ma_ez ), hipMemcpyDeviceToHost ) );
        MA_HIP( hipMemcpy( cigar_off, doff.p, n * 8, hipMemcpyDeviceToHost ) );
        cigar_off[ n ] = h[ 0 ];
        if( cigar && h[ 0 ] )
            MA_HIP( hipMemcpy( cigar, dpool.p, h[ 0 ] * 4, hipMemcpyDeviceToHost ) );
    }
    return rc;
}

extern "C" int ma_ksw_batch( const ma_params* P, const ma_ksw_job* jobs, uint64_t n, const uint8_t* q_bytes,
                             uint64_t q_len, const uint8_t* t_bytes, uint64_t t_len, ma_ez* ez, uint64_t* cigar_off,
                             uint32_t* cigar, uint64_t cigar_cap )
{
    return ksw_batch_impl<ByteFetch>( P, jobs, n, q_bytes, q_len, t_bytes, t_len, ez, cigar_off, cigar, cigar_cap );
}

extern "C" int ma_ksw_ext_batch( const ma_params* P, const ma_ksw_job* jobs, uint64_t n, const uint8_t* q_bytes,
                                 uint64_t q_len, const uint8_t* t_bytes, uint64_t t_len, ma_ez* ez, uint64_t* cigar_off,
                                 uint32_t* cigar, uint64_t cigar_cap )
{
    return ksw_batch_impl<ByteFetchPipe>( P, jobs, n, q_bytes, q_len, t_bytes, t_len, ez, cigar_off, cigar, cigar_cap );
}

// ---- diagnostics: the device libm values the chaining stage decides with (chain.h: tan, sin, atan, log), evaluated
// exactly like there (same translation-unit flags, -ffp-contract=off), so a test can compare their bits with glibc's
// (harmonization.h:82-89 and ransac.cpp:112,131-135 run on glibc in the reference)
__global__ void k_libm_probe( int op, const double* in, u64 n, double* out )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n )
        return;
    const double x = in[ i ];
    out[ i ] = op == 0 ? tan( x ) : op == 1 ? sin( x ) : op == 2 ? atan( x ) : log( x );
}
namespace ma
{
int band_stats_of_prims( unsigned long long out[ 8 ] )
{
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( ma::g_band_stats ), 8 * 8 ) );
    return 0;
}
int band_long_stats_of_prims( unsigned long long out[ 8 ] )
{
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( ma::g_band_stats ), 8 * 8, 8 * 8 ) );
    return 0;
}
int dp_family_stats_of_prims( unsigned long long out[ 2 * KSW_N_FAMILIES ] )
{
    MA_HIP( hipMemcpyFromSymbol( out, HIP_SYMBOL( ma::g_dp_family ), 2 * KSW_N_FAMILIES * 8 ) );
    return 0;
}
} // namespace ma

extern "C" int ma_debug_libm( int op, const double* in, uint64_t n, double* out )
{
    if( !in || !out || op < 0 || op > 3 )
        return fail( "ma_debug_libm: bad argument" );
    if( n == 0 )
        return 0;
    DevBuf di, dout;
    if( di.reserve( n * 8 ) || dout.reserve( n * 8 ) )
        return 1;
    MA_HIP( hipMemcpy( di.p, in, n * 8, hipMemcpyHostToDevice ) );
    hipLaunchKernelGGL( k_libm_probe, dim3( (unsigned)( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, 0, op, di.as<double>( ), n,
                        dout.as<double>( ) );
    MA_HIP( hipGetLastError( ) );
    MA_HIP( hipMemcpy( out, dout.p, n * 8, hipMemcpyDeviceToHost ) );
    return 0;
}
