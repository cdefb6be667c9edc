// ksw_launch.h -- persistent-wave driver around ksw_wave_core: every 64-thread workgroup (= one
// wavefront) pulls DP job slots from an atomic counter until none are left.
#pragma once
#include "internal.h"
#include "ksw_wave.h"
#include "ksw_reg.h"
#include "ksw_ext.h"
#include "ksw_pk.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace ma
{
struct KswWaveScratch
{
    uint8_t* base; // per-wave HBM scratch
    u64 stride; // bytes per wave
    u64 state_cap; // bytes reserved for u|v|..|qr (used in HBM mode)
    u64 h_cap; // bytes for H
    u64 p_cap; // direction bytes
    u64 cig_cap; // cigar entries
    u32 use_lds; // state + H in dynamic LDS
};

struct KswOut
{
    ma_ez* ez; // per slot
    u64* cig_off; // per slot
    u32* cig_pool;
    u64 cig_pool_cap;
    unsigned long long* cig_used;
    unsigned long long* cells; // sum of band cells
    unsigned long long* njobs;
    unsigned long long* path; // back-trace steps
    unsigned long long* cig_words; // cigar words written (may be null)
    u32* err;
    u32 cig_chunk; // 0: one pool atomic per job (dense pool); else each wave reserves pool space this many words at a time
};

// Per-wave running totals and the wave's current cigar-pool reservation: the device-wide counters sit in one cache
// line, and same-line atomics serialise in L2 (~9 ns each), so they are touched once per wave, not once per job.
struct KswWaveAcc
{
    u64 cells = 0, njobs = 0, path = 0, cig_words = 0;
    u64 chunk_off = 0;
    u32 chunk_left = 0;
};
#define KSW_JOBS_PER_FETCH 8u // queue entries a wave takes per atomic

// job classes by the number of 128-cell ring slots they need (ksw_pk_slots): 1, 2, 3, <=5, else LDS kernel (class 3
// shares the launch slot of the widest ring; class 4 = ksw_wave.h)
#define KSW_S0 1
#define KSW_S1 2
#define KSW_S2 3
#define KSW_S3 5
#define KSW_N_CLASSES 7 // 0..3 exact register kernel (ksw_pk.h), 4 LDS kernel, 5 / 6 extension kernel (ksw_ext.h) with 1 / 2 slots
MA_HD int ksw_job_class( i32 qlen, i32 tlen, i32 w )
{
    if( qlen > 150000 )
        return 4; // the register kernels keep the reversed query in LDS (160 KB per CU)
    const i32 n = ksw_pk_slots( qlen, tlen, w ); // 128-cell slots of the two-cells-per-lane kernel (ksw_pk.h)
    return n <= KSW_S0 ? 0 : ( n <= KSW_S1 ? 1 : ( n <= KSW_S2 ? 2 : ( n <= KSW_S3 ? 3 : 4 ) ) );
}
// pipeline mode: extensions whose callers read only max_q / max_t / cigar go to the extension kernel
MA_HD int ksw_job_class_pipe( const KswScoring& SC, i32 qlen, i32 tlen, i32 w, i32 zdrop, i32 flag )
{
    const int e = ksw_ext_slots( SC, qlen, tlen, w, zdrop, flag );
    return e ? 4 + e : ksw_job_class( qlen, tlen, w );
}

// result record + cigar of one finished job -> output arrays (all 64 lanes call this)
__device__ __forceinline__ void ksw_publish( const KswOut& O, KswWaveAcc& A, u32 slot, const KswEz& ez, u32 nCig, u64 cells,
                                             u64 path, const u32* cig, unsigned long long* sOff )
{
    A.cells += cells;
    A.njobs += 1;
    A.path += path;
    A.cig_words += nCig;
    u64 off;
    if( O.cig_chunk == 0 || nCig > O.cig_chunk )
    {
        if( threadIdx.x == 0 )
            *sOff = atomicAdd( O.cig_used, (unsigned long long)nCig );
        __syncthreads( );
        off = *sOff;
        __syncthreads( );
    }
    else
    {
        if( nCig > A.chunk_left )
        {
            if( threadIdx.x == 0 )
                *sOff = atomicAdd( O.cig_used, (unsigned long long)O.cig_chunk );
            __syncthreads( );
            A.chunk_off = *sOff;
            A.chunk_left = O.cig_chunk;
            __syncthreads( );
        }
        off = A.chunk_off;
        A.chunk_off += nCig;
        A.chunk_left -= nCig;
    }
    const bool fits = off + nCig <= O.cig_pool_cap;
    if( threadIdx.x == 0 )
    {
        ma_ez r;
        r.max = (i32)ez.max;
        r.zdropped = ez.zdropped;
        r.max_q = ez.max_q;
        r.max_t = ez.max_t;
        r.mqe = ez.mqe;
        r.mqe_t = ez.mqe_t;
        r.mte = ez.mte;
        r.mte_q = ez.mte_q;
        r.score = ez.score;
        r.reach_end = ez.reach_end;
        r.n_cigar = (i32)nCig;
        O.ez[ slot ] = r;
        O.cig_off[ slot ] = off;
        if( !fits )
            atomicOr( O.err, MA_ERR_CIGAR_OVERFLOW );
    }
    if( fits )
        for( u32 i = threadIdx.x; i < nCig; i += 64 )
            O.cig_pool[ off + i ] = cig[ i ];
    __syncthreads( ); // the scratch cigar may be overwritten by the next job
}
__device__ __forceinline__ void ksw_flush( const KswOut& O, const KswWaveAcc& A )
{
    if( threadIdx.x == 0 && A.njobs )
    {
        atomicAdd( O.cells, (unsigned long long)A.cells );
        atomicAdd( O.njobs, (unsigned long long)A.njobs );
        if( O.path )
            atomicAdd( O.path, (unsigned long long)A.path );
        if( O.cig_words )
            atomicAdd( O.cig_words, (unsigned long long)A.cig_words );
    }
}
// next queue position of this wave: KSW_JOBS_PER_FETCH entries per atomic
__device__ __forceinline__ bool ksw_next( unsigned int* nextSlot, u32 n, u32& cur, u32& end, u32* sSlot, u32& at )
{
    if( cur == end )
    {
        if( threadIdx.x == 0 )
            *sSlot = atomicAdd( nextSlot, KSW_JOBS_PER_FETCH );
        __syncthreads( );
        cur = *sSlot;
        __syncthreads( );
        end = cur + KSW_JOBS_PER_FETCH < n ? cur + KSW_JOBS_PER_FETCH : n;
        if( cur >= n )
            return false;
    }
    at = cur++;
    return true;
}

// job source of a launch: mode 0 = list[0..n), 1 = every slot 0..n that is valid and of class cls,
// 2 = list[0..*nDev) filtered by class cls (the jobs the extension kernel handed back)
struct KswJobs
{
    const u32* list;
    u32 n;
    const unsigned int* nDev;
    int mode, cls;
};

#if defined( MA_KSW_PROF )
#define KSW_PROF_T( v ) const unsigned long long v = clock64( )
#define KSW_PROF_ADD( i, a, b ) prof[ i ] += ( b ) - ( a )
#define KSW_PROF_ARG , prof
#else
#define KSW_PROF_ARG
#define KSW_PROF_T( v )
#define KSW_PROF_ADD( i, a, b )
#endif

template <typename FETCH, int R>
__global__ void __launch_bounds__( 64 ) __attribute__( ( amdgpu_waves_per_eu( R == 1 ? 7 : 4 ) ) ) k_ksw_ext( FETCH F, KswScoring SC, const u32* list, u32 n, unsigned int* nextSlot,
                                                  uint8_t* scratch, u64 stride, u64 p_cap, u32 ldsBytes, KswOut O,
                                                  u32* redo, unsigned int* nRedo )
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) char lds[];
    __shared__ u32 sSlot;
    __shared__ unsigned long long sOff;
    uint8_t* my = scratch + (u64)blockIdx.x * stride;
    u32* cig = (u32*)( my + p_cap );
#if defined( MA_KSW_PROF )
    unsigned long long prof[ 16 ] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    KswWaveAcc acc;
    u32 qCur = 0, qEnd = 0;
    while( true )
    {
        KSW_PROF_T( t0 );
        u32 at;
        if( !ksw_next( nextSlot, n, qCur, qEnd, &sSlot, at ) )
            break;
        const u32 slot = list[ at ];
        const KswJobView J = F.view( slot );
        KswEz ez;
        u32 nCig = 0;
        u64 cells = 0, path = 0;
        auto qf = F.qfetch( slot );
        auto tf = F.tfetch( slot );
        bool ok;
        KSW_PROF_T( t1 );
        if( !( J.flag & KSW_EZ_EXTZ_ONLY ) )
        {
            if( J.flag & KSW_EZ_RIGHT )
                ok = ksw_ext_core<R, false, true>( SC, J, qf, tf, (uint8_t*)lds, ldsBytes, my, cig, ez, nCig, cells, path KSW_PROF_ARG );
            else
                ok = ksw_ext_core<R, true, true>( SC, J, qf, tf, (uint8_t*)lds, ldsBytes, my, cig, ez, nCig, cells, path KSW_PROF_ARG );
        }
        else if( J.flag & KSW_EZ_RIGHT )
            ok = ksw_ext_core<R, false, false>( SC, J, qf, tf, (uint8_t*)lds, ldsBytes, my, cig, ez, nCig, cells, path KSW_PROF_ARG );
        else
            ok = ksw_ext_core<R, true, false>( SC, J, qf, tf, (uint8_t*)lds, ldsBytes, my, cig, ez, nCig, cells, path KSW_PROF_ARG );
        if( !ok )
        {
            if( threadIdx.x == 0 )
                redo[ atomicAdd( nRedo, 1u ) ] = slot;
            continue;
        }
        KSW_PROF_T( t2 );
        ksw_publish( O, acc, slot, ez, nCig, cells, path, cig, &sOff );
        KSW_PROF_T( t3 );
        KSW_PROF_ADD( 0, t0, t1 );
        KSW_PROF_ADD( 1, t1, t2 );
        KSW_PROF_ADD( 2, t2, t3 );
        KSW_PROF_ADD( 3, 0ull, 1ull );
    }
    ksw_flush( O, acc );
#if defined( MA_KSW_PROF )
    if( threadIdx.x == 0 )
        for( int i = 0; i < 16; i++ )
            atomicAdd( &g_ksw_prof[ i ], prof[ i ] );
#endif
}

template <typename FETCH, int S>
__global__ void __launch_bounds__( 64 ) __attribute__( ( amdgpu_waves_per_eu( S >= 5 ? 4 : 3 ) ) ) k_ksw_pk( FETCH F, KswScoring SC, KswJobs JB, unsigned int* nextSlot,
                                                  uint8_t* scratch, u64 stride, u64 p_cap, u32 ldsBytes, KswOut O )
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) char lds[];
    __shared__ u32 sSlot;
    __shared__ unsigned long long sOff;
    uint8_t* my = scratch + (u64)blockIdx.x * stride;
    uint8_t* P = my;
    u32* cig = (u32*)( my + p_cap );
    const u32 n = JB.mode == 2 ? *JB.nDev : JB.n;
    KswWaveAcc acc;
    u32 qCur = 0, qEnd = 0;
    while( true )
    {
        u32 at;
        if( !ksw_next( nextSlot, n, qCur, qEnd, &sSlot, at ) )
            break;
        const u32 slot = JB.mode == 1 ? at : JB.list[ at ];
        if( JB.mode == 1 && !F.valid( slot ) )
            continue;
        const KswJobView J = F.view( slot );
        if( JB.mode != 0 && ksw_job_class( J.qlen, J.tlen, J.w ) != JB.cls )
            continue;
        KswEz ez;
        u32 nCig = 0;
        u64 cells = 0, path = 0;
        auto qf = F.qfetch( slot );
        auto tf = F.tfetch( slot );
        ksw_pk_core<S, FETCH::EARLY>( SC, J, qf, tf, (uint8_t*)lds, P, cig, ez, nCig, cells, path, ldsBytes );
        ksw_publish( O, acc, slot, ez, nCig, cells, path, cig, &sOff );
    }
    ksw_flush( O, acc );
}

template <typename FETCH>
__global__ void __launch_bounds__( 64 ) k_ksw( FETCH F, KswScoring SC, KswJobs JB, unsigned int* nextSlot,
                                              KswWaveScratch WS, u32 ldsBytes, KswOut O )
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) char lds[];
    __shared__ u32 sSlot;
    uint8_t* my = WS.base + (u64)blockIdx.x * WS.stride;
    KswMem M;
    if( WS.use_lds )
    {
        M.u = (int8_t*)lds;
        M.H = (void*)( lds + WS.state_cap );
        M.p = my;
        M.cig = (u32*)( my + WS.p_cap );
        M.stage = (uint8_t*)lds;
        M.stageBytes = ldsBytes;
    }
    else
    {
        M.u = (int8_t*)my;
        M.H = (void*)( my + WS.state_cap );
        M.p = my + WS.state_cap + WS.h_cap;
        M.cig = (u32*)( my + WS.state_cap + WS.h_cap + WS.p_cap );
        M.stage = nullptr;
        M.stageBytes = 0;
    }
    __shared__ unsigned long long sOff;
    const u32 n = JB.mode == 2 ? *JB.nDev : JB.n;
    KswWaveAcc acc;
    u32 qCur = 0, qEnd = 0;
    while( true )
    {
        u32 at;
        if( !ksw_next( nextSlot, n, qCur, qEnd, &sSlot, at ) )
            break;
        const u32 slot = JB.mode == 1 ? at : JB.list[ at ];
        if( JB.mode == 1 && !F.valid( slot ) )
            continue;
        const KswJobView J = F.view( slot );
        if( JB.mode != 0 && ksw_job_class( J.qlen, J.tlen, J.w ) != 4 )
            continue; // handled by a register-resident launch
        M.L = ( ( J.tlen + 15 ) / 16 ) * 16;
        KswEz ez;
        u32 nCig = 0;
        u64 cells = 0, path = 0;
        auto qf = F.qfetch( slot );
        auto tf = F.tfetch( slot );
        if( ksw_h16( SC, J.qlen, J.tlen ) )
            ksw_wave_core<int16_t, 8>( SC, J, qf, tf, M, ez, nCig, cells, path );
        else
            ksw_wave_core<int32_t, 4>( SC, J, qf, tf, M, ez, nCig, cells, path );
        ksw_publish( O, acc, slot, ez, nCig, cells, path, M.cig, &sOff );
    }
    ksw_flush( O, acc );
}

// sizes for a job population (host side)
struct KswSizing
{
    u64 state = 0, h = 0, p = 0, cig = 0;
    u64 qlen = 0; // longest query (LDS bytes of the register kernels)
    u64 cls[ KSW_N_CLASSES ] = { 0, 0, 0, 0, 0, 0, 0 }; // jobs per class
};
inline void ksw_size_job( KswSizing& S, i32 qlen, i32 tlen, i32 w )
{
    if( qlen <= 0 || tlen <= 0 )
        return;
    S.cls[ ksw_job_class( qlen, tlen, w ) ]++;
    S.qlen = S.qlen > (u64)qlen ? S.qlen : (u64)qlen;
    const u64 st = ksw_state_bytes( qlen, tlen );
    const u64 L = (u64)( ( tlen + 15 ) / 16 ) * 16;
    const u64 p = (u64)( (i64)qlen + tlen - 1 ) * (u64)( ksw_ncol( qlen, tlen, w ) * 16 ) + 16;
    S.state = S.state > st ? S.state : st;
    S.h = S.h > L * 4 ? S.h : L * 4;
    S.p = S.p > p ? S.p : p;
    const u64 c = (u64)qlen + tlen + 2;
    S.cig = S.cig > c ? S.cig : c;
}

// Plans the launch: fills WS (without base) and returns waves + dynamic LDS bytes
struct KswPlan
{
    KswWaveScratch ws;
    u32 waves;
    u32 lds_bytes;
};
inline KswPlan ksw_plan( const KswSizing& S, u64 nJobs, u64 scratch_budget_bytes )
{
    KswPlan P;
    memset( &P, 0, sizeof( P ) );
    auto al = []( u64 x ) { return ( x + 255 ) / 256 * 256; };
    P.ws.state_cap = al( S.state );
    P.ws.h_cap = al( S.h );
    P.ws.p_cap = al( S.p );
    P.ws.cig_cap = S.cig;
    const u64 ldsNeed = P.ws.state_cap + P.ws.h_cap;
    P.ws.use_lds = ldsNeed <= 64 * 1024 ? 1 : 0;
    P.lds_bytes = P.ws.use_lds ? (u32)ldsNeed : 0;
    P.ws.stride = al( ( P.ws.use_lds ? 0 : P.ws.state_cap + P.ws.h_cap ) + P.ws.p_cap + al( S.cig * 4 ) );
    u64 waves = 256ull * 16; // 16 single-wave workgroups per CU
    if( P.ws.use_lds )
    {
        const u64 perCu = ( 160 * 1024 ) / ( ldsNeed ? ldsNeed : 1 );
        waves = 256ull * ( perCu > 16 ? 16 : ( perCu < 1 ? 1 : perCu ) );
    }
    if( waves > nJobs )
        waves = nJobs;
    if( P.ws.stride * waves > scratch_budget_bytes )
        waves = scratch_budget_bytes / ( P.ws.stride ? P.ws.stride : 1 );
    if( waves < 1 )
        waves = 1;
    P.waves = (u32)waves;
    return P;
}

// Launches every class that has jobs.  `next` = 12 zeroed counters (one per launch).  `lists` (device, or null) holds
// the job slots of class k at lists + k * list_stride, SZ.cls[k] entries, and room for the jobs the extension
// kernel hands back at lists + KSW_N_CLASSES * list_stride (counted in *nRedo); without lists every launch scans
// nSlots and there are no extension-kernel classes.
#define KSW_REG_LDS 6144u // per-wave LDS of the exact register kernels: reversed query, later the back-trace staging block
#define KSW_EXT_LDS 4096u // extension kernel: back-trace staging only (8 waves per SIMD fit)
template <typename FETCH>
int ksw_run_all( const FETCH& F, const KswScoring& SC, u32 nSlots, const KswSizing& SZ, DevBuf& scratch,
                 unsigned int* next, KswOut O, hipStream_t stream, u32* lists = nullptr, u64 list_stride = 0,
                 unsigned int* nRedo = nullptr )
{
    auto al = []( u64 x ) { return ( x + 255 ) / 256 * 256; };
    u64 nJobs = 0;
    for( int k = 0; k < KSW_N_CLASSES; k++ )
        nJobs += SZ.cls[ k ];
    if( nJobs == 0 )
        return 0;
    const u64 nExt = SZ.cls[ 5 ] + SZ.cls[ 6 ];
    const u64 p_cap = al( SZ.p );
    const u64 regStride = al( p_cap + al( SZ.cig * 4 ) );
    KswPlan plan = ksw_plan( SZ, SZ.cls[ 4 ] ? SZ.cls[ 4 ] : 1, 24ull << 30 );
    u64 perCu = 32; // waves per CU of the persistent ksw launches (MA_KSW_WAVES_PER_CU: tuning hook)
    if( const char* e = getenv( "MA_KSW_WAVES_PER_CU" ) )
        perCu = (u64)std::max( 1, atoi( e ) );
    u64 regWaves = std::min<u64>( 256ull * perCu, nJobs );
    if( regStride * regWaves > ( 24ull << 30 ) )
        regWaves = std::max<u64>( 1, ( 24ull << 30 ) / regStride );
    u64 need = std::max<u64>( regStride * regWaves, SZ.cls[ 4 ] ? plan.ws.stride * plan.waves : 0 );
    // the per-wave scratch follows the LARGEST job of the batch, which varies a lot from batch to batch for long
    // reads: once it is in the GB range take the whole budget so that later batches never re-allocate mid-step
    if( need > ( 2ull << 30 ) )
        need = 24ull << 30;
    if( scratch.reserve( need ) )
        return 1;
    const u32 ldsReg = std::max<u32>( (u32)( ( ( std::min<u64>( SZ.qlen, 150000 ) + 15 ) / 16 ) * 16 + 64 ), KSW_REG_LDS );
    if( ldsReg > 48 * 1024 )
    {
        MA_HIP( hipFuncSetAttribute( (const void*)k_ksw_pk<FETCH, KSW_S0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsReg ) );
        MA_HIP( hipFuncSetAttribute( (const void*)k_ksw_pk<FETCH, KSW_S1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsReg ) );
        MA_HIP( hipFuncSetAttribute( (const void*)k_ksw_pk<FETCH, KSW_S2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsReg ) );
        MA_HIP( hipFuncSetAttribute( (const void*)k_ksw_pk<FETCH, KSW_S3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsReg ) );
    }
    uint8_t* base = scratch.as<uint8_t>( );
    u32* redo = lists ? lists + (u64)KSW_N_CLASSES * list_stride : nullptr;
    auto grid = [ & ]( u64 jobs ) { return dim3( (unsigned)std::max<u64>( 1, std::min<u64>( regWaves, jobs ) ) ); };
    // the launches run back to back on one stream, so they can share the scratch
    if( SZ.cls[ 5 ] )
        hipLaunchKernelGGL( ( k_ksw_ext<FETCH, 1> ), grid( SZ.cls[ 5 ] ), dim3( 64 ), KSW_EXT_LDS, stream, F, SC,
                            lists + 5 * list_stride, (u32)SZ.cls[ 5 ], next + 5, base, regStride, p_cap, KSW_EXT_LDS, O, redo,
                            nRedo );
    if( SZ.cls[ 6 ] )
        hipLaunchKernelGGL( ( k_ksw_ext<FETCH, 2> ), grid( SZ.cls[ 6 ] ), dim3( 64 ), KSW_EXT_LDS, stream, F, SC,
                            lists + 6 * list_stride, (u32)SZ.cls[ 6 ], next + 6, base, regStride, p_cap, KSW_EXT_LDS, O, redo,
                            nRedo );
    for( int pass = 0; pass < ( nExt ? 2 : 1 ); pass++ )
    {
        // pass 0: the classes' own jobs; pass 1: whatever the extension kernel handed back (usually nothing)
        auto jobs = [ & ]( int k ) {
            KswJobs JB;
            JB.list = pass ? redo : ( lists ? lists + (u64)k * list_stride : nullptr );
            JB.n = lists ? (u32)SZ.cls[ k ] : nSlots;
            JB.nDev = nRedo;
            JB.mode = pass ? 2 : ( lists ? 0 : 1 );
            JB.cls = k;
            return JB;
        };
        auto cnt = [ & ]( int k ) { return pass ? std::min<u64>( nExt, 256 * 4 ) : SZ.cls[ k ]; };
        unsigned int* nx = next + ( pass ? 7 : 0 );
        if( cnt( 0 ) )
            hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S0> ), grid( cnt( 0 ) ), dim3( 64 ), ldsReg, stream, F, SC, jobs( 0 ),
                                nx + 0, base, regStride, p_cap, ldsReg, O );
        if( cnt( 1 ) )
            hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S1> ), grid( cnt( 1 ) ), dim3( 64 ), ldsReg, stream, F, SC, jobs( 1 ),
                                nx + 1, base, regStride, p_cap, ldsReg, O );
        if( cnt( 2 ) )
            hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S2> ), grid( cnt( 2 ) ), dim3( 64 ), ldsReg, stream, F, SC, jobs( 2 ),
                                nx + 2, base, regStride, p_cap, ldsReg, O );
        if( cnt( 3 ) )
            hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S3> ), grid( cnt( 3 ) ), dim3( 64 ), ldsReg, stream, F, SC, jobs( 3 ),
                                nx + 3, base, regStride, p_cap, ldsReg, O );
        if( !pass && cnt( 4 ) ) // a handed-back job always fits a register kernel
        {
            plan.ws.base = base;
            if( plan.lds_bytes > 48 * 1024 )
                MA_HIP( hipFuncSetAttribute( (const void*)k_ksw<FETCH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)plan.lds_bytes ) );
            hipLaunchKernelGGL( k_ksw<FETCH>, dim3( plan.waves ), dim3( 64 ), plan.lds_bytes, stream, F, SC,
                                jobs( 4 ), nx + 4, plan.ws, plan.lds_bytes, O );
        }
    }
    MA_HIP( hipGetLastError( ) );
    return 0;
}
} // namespace ma
