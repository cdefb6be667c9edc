// ksw_launch.h -- persistent-wave driver around ksw_wave_core: every 64-thread workgroup (= one
// wavefront) pulls DP job slots from an atomic counter until none are left.
#pragma once
#include "internal.h"
#include "ksw_wave.h"
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

template <typename FETCH>
__global__ void __launch_bounds__( 64 ) k_ksw( FETCH F, KswScoring SC, u32 nSlots, unsigned int* nextSlot,
                                              KswWaveScratch WS, KswOut O )
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
    }
    else
    {
        M.u = (int8_t*)my;
        M.H = (void*)( my + WS.state_cap );
        M.p = my + WS.state_cap + WS.h_cap;
        M.cig = (u32*)( my + WS.state_cap + WS.h_cap + WS.p_cap );
    }
    while( true )
    {
        if( threadIdx.x == 0 )
            sSlot = atomicAdd( nextSlot, 1u );
        __syncthreads( );
        const u32 slot = sSlot;
        __syncthreads( );
        if( slot >= nSlots )
            break;
        if( !F.valid( slot ) )
            continue;
        const KswJobView J = F.view( slot );
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
};
inline void ksw_size_job( KswSizing& S, i32 qlen, i32 tlen, i32 w )
{
    if( qlen <= 0 || tlen <= 0 )
        return;
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
} // namespace ma
