// ksw_launch.h -- persistent-wave driver around ksw_wave_core: every 64-thread workgroup (= one
// wavefront) pulls DP job slots from an atomic counter until none are left.
#pragma once
#include "internal.h"
#include "ksw_wave.h"
#include "ksw_reg.h"
#include <algorithm>
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
    u32* err;
};

// job classes by the number of ring slots they need (ksw_need_slots): 1, 2, <=4, <=9, else LDS kernel (class 3
// shares the launch slot of the widest ring; class 4 = ksw_wave.h)
#define KSW_S0 1
#define KSW_S1 2
#define KSW_S2 3
#define KSW_S3 9
MA_HD int ksw_job_class( i32 qlen, i32 tlen, i32 w )
{
    const i32 n = ksw_need_slots( qlen, tlen, w );
    return n <= KSW_S0 ? 0 : ( n <= KSW_S1 ? 1 : ( n <= KSW_S2 ? 2 : ( n <= KSW_S3 ? 3 : 4 ) ) );
}

// `list` = the job slots of this class (n entries), or null: scan all n slots and skip other classes
template <typename FETCH, int S>
__global__ void __launch_bounds__( 64 ) k_ksw_reg( FETCH F, KswScoring SC, const u32* list, u32 n, unsigned int* nextSlot,
                                                  int cls, uint8_t* scratch, u64 stride, u64 p_cap, u32 ldsBytes, KswOut O )
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) char lds[];
    __shared__ u32 sSlot;
    __shared__ unsigned long long sOff;
    uint8_t* my = scratch + (u64)blockIdx.x * stride;
    uint8_t* P = my;
    u32* cig = (u32*)( my + p_cap );
    while( true )
    {
        if( threadIdx.x == 0 )
            sSlot = atomicAdd( nextSlot, 1u );
        __syncthreads( );
        const u32 at = sSlot;
        __syncthreads( );
        if( at >= n )
            break;
        const u32 slot = list ? list[ at ] : at;
        if( !list && !F.valid( slot ) )
            continue;
        const KswJobView J = F.view( slot );
        if( !list && ksw_job_class( J.qlen, J.tlen, J.w ) != cls )
            continue;
        KswEz ez;
        u32 nCig = 0;
        u64 cells = 0, path = 0;
        auto qf = F.qfetch( slot );
        auto tf = F.tfetch( slot );
        if( ksw_h16( SC, J.qlen, J.tlen ) )
            ksw_reg_core<S, int16_t, 8, FETCH::EARLY>( SC, J, qf, tf, (uint8_t*)lds, P, cig, ez, nCig, cells, path, ldsBytes );
        else
            ksw_reg_core<S, int32_t, 4, FETCH::EARLY>( SC, J, qf, tf, (uint8_t*)lds, P, cig, ez, nCig, cells, path, ldsBytes );
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
            sOff = atomicAdd( O.cig_used, (unsigned long long)nCig );
            O.cig_off[ slot ] = sOff;
            if( sOff + nCig > O.cig_pool_cap )
                atomicOr( O.err, MA_ERR_CIGAR_OVERFLOW );
            atomicAdd( O.cells, (unsigned long long)cells );
            atomicAdd( O.njobs, 1ull );
            if( O.path )
                atomicAdd( O.path, (unsigned long long)path );
        }
        __syncthreads( );
        const u64 off = sOff;
        if( off + nCig <= O.cig_pool_cap )
            for( u32 i = threadIdx.x; i < nCig; i += 64 )
                O.cig_pool[ off + i ] = cig[ i ];
        __syncthreads( );
    }
}

template <typename FETCH>
__global__ void __launch_bounds__( 64 ) k_ksw( FETCH F, KswScoring SC, const u32* list, u32 n, unsigned int* nextSlot,
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
    while( true )
    {
        if( threadIdx.x == 0 )
            sSlot = atomicAdd( nextSlot, 1u );
        __syncthreads( );
        const u32 at = sSlot;
        __syncthreads( );
        if( at >= n )
            break;
        const u32 slot = list ? list[ at ] : at;
        if( !list && !F.valid( slot ) )
            continue;
        const KswJobView J = F.view( slot );
        if( !list && ksw_job_class( J.qlen, J.tlen, J.w ) != 4 )
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
        // publish
        __shared__ unsigned long long sOff;
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
            sOff = atomicAdd( O.cig_used, (unsigned long long)nCig );
            O.cig_off[ slot ] = sOff;
            if( sOff + nCig > O.cig_pool_cap )
                atomicOr( O.err, MA_ERR_CIGAR_OVERFLOW );
            atomicAdd( O.cells, (unsigned long long)cells );
            atomicAdd( O.njobs, 1ull );
            if( O.path )
                atomicAdd( O.path, (unsigned long long)path );
        }
        __syncthreads( );
        const u64 off = sOff;
        if( off + nCig <= O.cig_pool_cap )
            for( u32 i = threadIdx.x; i < nCig; i += 64 )
                O.cig_pool[ off + i ] = M.cig[ i ];
        __syncthreads( );
    }
}

// sizes for a job population (host side)
struct KswSizing
{
    u64 state = 0, h = 0, p = 0, cig = 0;
    u64 qlen = 0; // longest query (LDS bytes of the register kernels)
    u64 cls[ 5 ] = { 0, 0, 0, 0, 0 }; // jobs per class
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

// Launches every class that has jobs. `next` = 5 zeroed counters (one per launch).  `lists` (device, or null) holds
// the job slots of class k at lists + k * list_stride, SZ.cls[k] entries; without it every launch scans nSlots.
#define KSW_REG_LDS 6144u // per-wave LDS of the ring kernels: reversed query, later the back-trace staging block
template <typename FETCH>
int ksw_run_all( const FETCH& F, const KswScoring& SC, u32 nSlots, const KswSizing& SZ, DevBuf& scratch,
                 unsigned int* next, KswOut O, hipStream_t stream, const u32* lists = nullptr, u64 list_stride = 0 )
{
    auto al = []( u64 x ) { return ( x + 255 ) / 256 * 256; };
    const u64 nJobs = SZ.cls[ 0 ] + SZ.cls[ 1 ] + SZ.cls[ 2 ] + SZ.cls[ 3 ] + SZ.cls[ 4 ];
    if( nJobs == 0 )
        return 0;
    const u64 p_cap = al( SZ.p );
    const u64 regStride = al( p_cap + al( SZ.cig * 4 ) );
    KswPlan plan = ksw_plan( SZ, SZ.cls[ 4 ] ? SZ.cls[ 4 ] : 1, 24ull << 30 );
    u64 regWaves = std::min<u64>( 256ull * 24, nJobs );
    if( regStride * regWaves > ( 24ull << 30 ) )
        regWaves = std::max<u64>( 1, ( 24ull << 30 ) / regStride );
    const u64 need = std::max<u64>( regStride * regWaves, SZ.cls[ 4 ] ? plan.ws.stride * plan.waves : 0 );
    if( scratch.reserve( need ) )
        return 1;
    const u32 ldsReg = std::max<u32>( (u32)( ( ( SZ.qlen + 15 ) / 16 ) * 16 + 64 ), KSW_REG_LDS );
    uint8_t* base = scratch.as<uint8_t>( );
    auto lst = [ & ]( int k ) { return lists ? lists + (u64)k * list_stride : (const u32*)nullptr; };
    auto cnt = [ & ]( int k ) { return lists ? (u32)SZ.cls[ k ] : nSlots; };
    // the launches run back to back on one stream, so they can share the scratch
    if( SZ.cls[ 0 ] )
        hipLaunchKernelGGL( ( k_ksw_reg<FETCH, KSW_S0> ), dim3( (unsigned)std::min<u64>( regWaves, SZ.cls[ 0 ] ) ), dim3( 64 ),
                            ldsReg, stream, F, SC, lst( 0 ), cnt( 0 ), next + 0, 0, base, regStride, p_cap, ldsReg, O );
    if( SZ.cls[ 1 ] )
        hipLaunchKernelGGL( ( k_ksw_reg<FETCH, KSW_S1> ), dim3( (unsigned)std::min<u64>( regWaves, SZ.cls[ 1 ] ) ), dim3( 64 ),
                            ldsReg, stream, F, SC, lst( 1 ), cnt( 1 ), next + 1, 1, base, regStride, p_cap, ldsReg, O );
    if( SZ.cls[ 2 ] )
        hipLaunchKernelGGL( ( k_ksw_reg<FETCH, KSW_S2> ), dim3( (unsigned)std::min<u64>( regWaves, SZ.cls[ 2 ] ) ), dim3( 64 ),
                            ldsReg, stream, F, SC, lst( 2 ), cnt( 2 ), next + 2, 2, base, regStride, p_cap, ldsReg, O );
    if( SZ.cls[ 3 ] )
        hipLaunchKernelGGL( ( k_ksw_reg<FETCH, KSW_S3> ), dim3( (unsigned)std::min<u64>( regWaves, SZ.cls[ 3 ] ) ), dim3( 64 ),
                            ldsReg, stream, F, SC, lst( 3 ), cnt( 3 ), next + 3, 3, base, regStride, p_cap, ldsReg, O );
    if( SZ.cls[ 4 ] )
    {
        plan.ws.base = base;
        if( plan.lds_bytes > 48 * 1024 )
            MA_HIP( hipFuncSetAttribute( (const void*)k_ksw<FETCH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)plan.lds_bytes ) );
        hipLaunchKernelGGL( k_ksw<FETCH>, dim3( plan.waves ), dim3( 64 ), plan.lds_bytes, stream, F, SC, lst( 4 ), cnt( 4 ),
                            next + 4, plan.ws, plan.lds_bytes, O );
    }
    MA_HIP( hipGetLastError( ) );
    return 0;
}
} // namespace ma
