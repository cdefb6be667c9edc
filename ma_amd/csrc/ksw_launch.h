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
// ... when the launch has jobs in plenty.  With few jobs per wave -- the long junk extensions of a 10 kb batch on k_ksw_pk<5> (8.5 k jobs
// of ~8 ms on 4 096 waves), the handful of jobs the band of 120 hands back -- eight at a time leaves most waves without work and
// makes the launch as long as eight jobs in a row (10 kb: the 150 handed-back jobs were a tail of 40 ms on 19 waves): one at a time.
__device__ __forceinline__ u32 ksw_fetch_size( u32 n )
{
    return n / gridDim.x >= 32u ? KSW_JOBS_PER_FETCH : 1u;
}

// job classes by the number of 128-cell ring slots they need (ksw_pk_slots): 1, 2, 3, <=5, else LDS kernel (class 3
// shares the launch slot of the widest ring; class 4 = ksw_wave.h)
#define KSW_S0 1
#define KSW_S1 2
#define KSW_S2 3
#define KSW_S3 5
#define KSW_N_CLASSES 15 // 0..3 exact register kernel (ksw_pk.h), 4 LDS kernel, 5 / 6 extension kernel (ksw_ext.h) with 1 / 2 slots,
                         // 7..12 the query-stationary extension kernel (ksw_grp.h): 4 jobs of up to 254 query bases on the proven narrow band
                         // (ksw_band.h) / 2 of up to 64 / 4 of up to 32 per wave, left / right; 13 / 14 LONG extension jobs on the proven band
                         // of 120, one per wave, left / right (ksw_band.h, G = 1)
#define KSW_CLS_GRP0 7
#define KSW_CLS_BANDL 13
MA_HD int ksw_job_class( i32 qlen, i32 tlen, i32 w )
{
    if( qlen > 150000 )
        return 4; // the register kernels keep the reversed query in LDS (160 KB per CU)
    const i32 n = ksw_pk_slots( qlen, tlen, w ); // 128-cell slots of the two-cells-per-lane kernel (ksw_pk.h)
    return n <= KSW_S0 ? 0 : ( n <= KSW_S1 ? 1 : ( n <= KSW_S2 ? 2 : ( n <= KSW_S3 ? 3 : 4 ) ) );
}
// per-wave scratch of one job: direction bytes of the exact kernels (n_col bytes per diagonal) / of the extension kernel
// (one ring row per diagonal)
MA_HD u64 ksw_p_bytes( i32 qlen, i32 tlen, i32 w )
{
    return (u64)ksw_max_diags( qlen, tlen, w ) * (u64)( ksw_ncol( qlen, tlen, w ) * 16 ) + 16;
}
MA_HD u64 ksw_ext_p_bytes( i32 qlen, i32 tlen, int R )
{
    return (u64)( (i64)qlen + tlen - 1 ) * (u64)( 128 * R ) + 16;
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
// cells computed and jobs finished per DP kernel FAMILY since the library was loaded (ma_debug_dp_family_stats; bench.py divides a
// family's VALU instructions of the PMC pass by ITS cells): 0 k_ksw_ext<1>, 1 k_ksw_ext<2>, 2 k_ksw_grp<2>, 3 k_ksw_grp<4>,
// 4 k_ksw_band (four short jobs per wave), 5 k_ksw_pk, 6 k_ksw (LDS), 7 k_ksw_band (one long job per wave)
#define KSW_N_FAMILIES 8
static __device__ unsigned long long g_dp_family[ 2 * KSW_N_FAMILIES ];
__device__ __forceinline__ void ksw_flush( const KswOut& O, const KswWaveAcc& A, int family )
{
    if( threadIdx.x == 0 && A.njobs )
    {
        atomicAdd( g_dp_family + 2 * family, (unsigned long long)A.cells );
        atomicAdd( g_dp_family + 2 * family + 1, (unsigned long long)A.njobs );
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
        const u32 fetch = ksw_fetch_size( n );
        if( threadIdx.x == 0 )
            *sSlot = atomicAdd( nextSlot, fetch );
        __syncthreads( );
        cur = *sSlot;
        __syncthreads( );
        end = cur + fetch < n ? cur + fetch : n;
        if( cur >= n )
            return false;
    }
    at = cur++;
    return true;
}

} // namespace ma
#include "ksw_grp.h"
#include "ksw_band.h"
namespace ma
{
// MA_KSW_GRP=0 (A/B and test hook) keeps the short extensions on the one-job-per-wavefront kernel.  The switch travels in
// KswScoring (a kernel argument of everything that classifies jobs), so host and device classify alike.
#define ksw_grp_enabled( ) ( SC.grp != 0 )
// pipeline mode: extensions whose callers read only max_q / max_t / cigar go to the extension kernel
MA_HD int ksw_job_class_pipe( const KswScoring& SC, i32 qlen, i32 tlen, i32 w, i32 zdrop, i32 flag )
{
    const int e = ksw_ext_slots( SC, qlen, tlen, w, zdrop, flag );
    if( e && SC.grp >= 1000 && ksw_band_ok( SC, qlen, tlen, w, zdrop, flag, SC.grp - 1000 ) )
        return KSW_CLS_GRP0 + ( ( flag & KSW_EZ_RIGHT ) ? 1 : 0 ); // the proven narrow band: four jobs per wave (ksw_band.h)
    if( e == 1 && ksw_grp_enabled( ) )
        if( const int G = ksw_grp_size( SC, qlen, tlen, w, zdrop, flag ) )
            return KSW_CLS_GRP0 + ( G == 4 ? 4 : ( G == 2 ? 2 : 0 ) ) + ( ( flag & KSW_EZ_RIGHT ) ? 1 : 0 );
    if( !e && SC.band_long && ksw_bandl_ok( SC, qlen, tlen, w, zdrop, flag ) )
        return KSW_CLS_BANDL + ( ( flag & KSW_EZ_RIGHT ) ? 1 : 0 ); // the proven band of 120, one long job per wave (ksw_band.h)
    return e ? 4 + e : ksw_job_class( qlen, tlen, w );
}

// job source of a launch: mode 0 = list[0..n), 1 = every slot 0..n that is valid and of class cls,
// 2 = list[0..*nDev) filtered by class cls (the jobs the extension kernel handed back)
struct KswJobs
{
    const u32* list;
    u32 n;
    const unsigned int* nDev;
    int mode, cls;
    // a class with a few huge jobs is run as two launches: tier 1 takes the jobs with ksw_p_bytes <= pSplit and
    // ksw_q_lds <= qSplit (scratch and LDS of a full set of waves), tier 2 the others on fewer waves; tier 0 takes all
    int tier;
    u64 pSplit;
    i32 qSplit;
};
MA_HD bool ksw_tier_takes( const KswJobs& JB, i32 qlen, i32 tlen, i32 w )
{
    if( JB.tier == 0 )
        return true;
    const bool small = ksw_p_bytes( qlen, tlen, w ) <= JB.pSplit && ksw_q_lds( qlen, tlen, w ) <= (i64)JB.qSplit;
    return JB.tier == 1 ? small : !small;
}

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
                                                  u32* redo, unsigned int* nRedo, const unsigned int* nMore /*jobs k_ksw_band appended to the list, or null*/ )
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) char lds[];
    if( nMore )
        n += *nMore;
    __shared__ u32 sSlot;
    __shared__ unsigned long long sOff;
    __shared__ uint2 sSnap[ 64 * R ]; // the lanes' H of the diagonal that raised ez.max last (ksw_ext.h)
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
                ok = ksw_ext_core<R, false, true>( SC, J, qf, tf, (uint8_t*)lds, ldsBytes, my, cig, ez, nCig, cells, path, sSnap KSW_PROF_ARG );
            else
                ok = ksw_ext_core<R, true, true>( SC, J, qf, tf, (uint8_t*)lds, ldsBytes, my, cig, ez, nCig, cells, path, sSnap KSW_PROF_ARG );
        }
        else if( J.flag & KSW_EZ_RIGHT )
            ok = ksw_ext_core<R, false, false>( SC, J, qf, tf, (uint8_t*)lds, ldsBytes, my, cig, ez, nCig, cells, path, sSnap KSW_PROF_ARG );
        else
            ok = ksw_ext_core<R, true, false>( SC, J, qf, tf, (uint8_t*)lds, ldsBytes, my, cig, ez, nCig, cells, path, sSnap KSW_PROF_ARG );
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
#if defined( MA_KSW_PROF )
        // wave time (fetch .. publish) of the jobs that would fit HALF a wavefront (two jobs per wave: qlen + 2 <= 64 cells),
        // of those that would fit a quarter, and their numbers: the upper bound of what packing several jobs into one
        // wavefront can save (profiles/r03_ext_pairing_bound.txt)
        if( J.qlen + 2 <= 64 || J.tlen <= 64 )
        {
            prof[ 14 ] += t3 - t0;
            prof[ 15 ] += 1;
        }
        if( J.qlen + 2 <= 32 || J.tlen <= 32 )
        {
            prof[ 10 ] += t3 - t0; // (the GLOBAL slots 10 / 11 are re-used: see tools/ksw_prof.py)
            prof[ 11 ] += 1;
        }
        prof[ 9 ] += t3 - t0;
#endif
    }
    ksw_flush( O, acc, R == 1 ? 0 : 1 );
#if defined( MA_KSW_PROF )
    if( threadIdx.x == 0 )
        for( int i = 0; i < 16; i++ )
            atomicAdd( &g_ksw_prof[ i ], prof[ i ] );
#endif
}

#if !defined( MA_PK5_WAVES )
#define MA_PK5_WAVES 4
#endif
template <typename FETCH, int S>
__global__ void __launch_bounds__( 64 ) __attribute__( ( amdgpu_waves_per_eu( S >= 5 ? MA_PK5_WAVES : 3 ) ) ) k_ksw_pk( FETCH F, KswScoring SC, KswJobs JB, unsigned int* nextSlot,
                                                  uint8_t* scratch, u64 stride, u64 p_cap, u32 ldsBytes, KswOut O )
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) char lds[];
    __shared__ u32 sSlot;
    __shared__ unsigned long long sOff;
    __shared__ int2 sSnap[ 64 * S ]; // the lanes' H of the diagonal that raised ez.max last (ksw_pk.h)
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
        if( !ksw_tier_takes( JB, J.qlen, J.tlen, J.w ) )
            continue;
        KswEz ez;
        u32 nCig = 0;
        u64 cells = 0, path = 0;
        auto qf = F.qfetch( slot );
        auto tf = F.tfetch( slot );
        ksw_pk_core<S, FETCH::EARLY>( SC, J, qf, tf, (uint8_t*)lds, P, cig, ez, nCig, cells, path, ldsBytes, sSnap );
        ksw_publish( O, acc, slot, ez, nCig, cells, path, cig, &sOff );
    }
    ksw_flush( O, acc, 5 );
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
    ksw_flush( O, acc, 6 );
}

inline i32 ksw_grp_env( ) // KswScoring::grp (read on every call: the tests switch it inside one process)
{
    const char* e = getenv( "MA_KSW_GRP" );
    // 2 (experiment build -DMA_EXP_GRP_NR2 only): also the jobs of 65..128 query bases, two per wave with four rows per lane -- measured slower than k_ksw_ext<1> (150 bp:
    // DP 19.1 -> 20.0 ms): a register set's recurrence is 76 of the ~91 instructions of a diagonal, so sharing the rest buys 16 % per
    // job at best, and 128 VGPRs leave 4 waves per SIMD where k_ksw_ext runs 8
    // 3 (or 1000 + n): extensions of 65 (n) .. 254 query bases on the proven narrow band, four per wave (ksw_band.h).  Default: 1033
    // (150 bp: DP 19.0 ms with 1, 16.4 with 3, 15.7 with 1033)
    const i32 v = e ? std::max( 0, atoi( e ) ) : 1033;
#if defined( MA_EXP_GRP_NR2 )
    const i32 top = 2;
#else
    const i32 top = 1; // the shipped library has no four-rows-per-lane kernels (measured slower): 2 means 1
#endif
    return v == 3 ? 1065 : ( v >= 1000 ? std::min( v, 1000 + KSW_BAND_QMAX ) : std::min( v, top ) );
}
inline i32 ksw_bandl_env( ) // KswScoring::band_long
{
    const char* e = getenv( "MA_KSW_BANDL" );
    return e ? ( atoi( e ) != 0 ? 1 : 0 ) : 1;
}
inline i32 ksw_band_mis_env( )
{
    const char* e = getenv( "MA_KSW_BAND_MAXMIS" );
    return e ? std::max( 0, std::min( 64, atoi( e ) ) ) : KSW_BAND_MAXMIS;
}
// sizes for a job population (host side)
struct KswSizing
{
    u64 state = 0, h = 0, p = 0, cig = 0;
    u64 qlen = 0; // longest query (LDS bytes of the register kernels)
    u64 cls[ KSW_N_CLASSES ] = { }; // jobs per class
    u64 pc[ KSW_N_CLASSES ] = { }; // largest direction-byte scratch of a job, per class (0: use p)
    u64 cigc[ KSW_N_CLASSES ] = { }; // largest cigar scratch in words, per class (0: use cig)
    u64 pRedo = 0, cigRedo = 0; // the same for jobs the extension kernel hands back to the exact kernels (0: use p / cig)
    u64 bandlN = 0; // largest min(qlen, tlen) of the long jobs on the band of 120 (their scratch: KSW_BANDL_ROWS( N ) direction rows per wave)
};
inline void ksw_size_job( KswSizing& S, i32 qlen, i32 tlen, i32 w )
{
    if( qlen <= 0 || tlen <= 0 )
        return;
    const int k = ksw_job_class( qlen, tlen, w );
    S.cls[ k ]++;
    S.qlen = S.qlen > (u64)qlen ? S.qlen : (u64)qlen;
    const u64 st = ksw_state_bytes( qlen, tlen );
    const u64 L = (u64)( ( tlen + 15 ) / 16 ) * 16;
    const u64 p = ksw_p_bytes( qlen, tlen, w );
    S.pc[ k ] = S.pc[ k ] > p ? S.pc[ k ] : p;
    S.cigc[ k ] = S.cigc[ k ] > (u64)qlen + tlen + 2 ? S.cigc[ k ] : (u64)qlen + tlen + 2;
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

// Launches every class that has jobs.  `next` = 28 zeroed counters (one per launch; [11..16]: the six lists of k_ksw_grp, [17], [18]:
// jobs the narrow band appended to the extension kernels' lists, [19], [20]: the two lists of long jobs on the band of 120, [21..24]:
// the launches that take the long jobs the band handed back when they have a stream of their own, [25]: the number of those jobs),
// `nextBig` = 4 more.  `lists`
// (device, or null) holds the job slots of class k at lists + k * list_stride, SZ.cls[k] entries, and room for the jobs
// the extension kernel hands back at lists + KSW_N_CLASSES * list_stride (counted in *nRedo); without lists every
// launch scans nSlots and there are no extension-kernel classes.
//
// Scratch: every wave owns `stride` bytes (direction bytes + cigar of the job it is working on) of one allocation that
// the launches, which run back to back on one stream, share.  Each launch is sized for ITS class: the classes differ
// by orders of magnitude (gap between two seeds: KBs; end extension of a 50 kb read: 27 MB), and one stride for all
// would cut every launch down to the few hundred waves the largest job allows -- less than one wave per SIMD
// (50 kb reads: DP stage 2.4 s -> see DESIGN.md 3.6).  A class whose own largest job still does not leave room for a
// full set of waves is run as two launches: the jobs that fit a full set, then the few huge ones on fewer waves.
#define KSW_REG_LDS 6144u // per-wave LDS of the exact register kernels: reversed query, later the back-trace staging block
#define KSW_EXT_LDS 4096u // extension kernel: back-trace staging only (8 waves per SIMD fit)
// per-wave scratch of all resident waves of a launch may take this much HBM (MA_KSW_SCRATCH_MB: test hook that makes small
// batches take the paths of the large ones -- fewer waves, classes split in two launches, the side stream).
// B bounds ONE launch.  With the kernel classes on their own streams (the default for batches with job lists) four lanes of
// launches run side by side, each in its own region of B/4, 5B/12, B/3 and B/4 bytes: a batch whose DP stage needs more than
// 2 GB of scratch then holds 1.25 B (30 GB at the default), once per batch in flight -- DESIGN.md section 2 counts it that way.
inline u64 ksw_scratch_budget( )
{
    const char* e = getenv( "MA_KSW_SCRATCH_MB" ); // (read on every call: the tests switch it inside one process)
    return e && atoi( e ) > 0 ? (u64)atoi( e ) << 20 : 24ull << 30;
}
#define KSW_SCRATCH_BUDGET ksw_scratch_budget( )
struct KswLaunchPlan
{
    u64 p_cap = 0, stride = 0;
    u32 waves = 0;
    u32 lds = 0; // dynamic LDS bytes of the launch (register kernels: reversed query of ITS longest job)
    int tier = 0; // KswJobs::tier
    u64 region = 0; // byte offset of the launch's scratch region
};
inline KswLaunchPlan ksw_plan_launch( u64 p, u64 cigWords, u64 jobs, u64 wantWaves, u64 budget )
{
    auto al = []( u64 x ) { return ( x + 255 ) / 256 * 256; };
    KswLaunchPlan L;
    L.p_cap = al( p );
    L.stride = al( L.p_cap + al( cigWords * 4 ) );
    u64 waves = std::max<u64>( 1, std::min<u64>( wantWaves, jobs ) );
    if( L.stride * waves > budget )
        waves = std::max<u64>( 1, budget / L.stride );
    L.waves = (u32)waves;
    return L;
}
// Streams of one DP stage.  The kernel classes of a stage are independent of each other, and each is a persistent launch
// that ends in a tail -- a few waves still working on their last jobs while the rest of the machine idles.  On ONE stream
// the tails add up (50 kb reads: five launches, five tails); on separate streams, each launch with its own scratch region,
// the next class's waves take the SIMDs the moment a class's waves retire, so only the last tail of the stage is paid.
//   lane 0 (the batch's stream): extension kernel classes, then the jobs they handed back, then the LDS kernel
//   lane 1: k_ksw_pk<5>            lane 2: k_ksw_pk<3>, <2>, <1>            lane 3: the huge tiers of split classes
struct KswSide
{
    hipStream_t stream[ 3 ] = { nullptr, nullptr, nullptr };
    hipEvent_t fork = nullptr, join[ 3 ] = { nullptr, nullptr, nullptr };
    hipEvent_t band = nullptr; // the long jobs' band kernels are done: their handed-back jobs start on lane 3 (ksw_run_all)
    bool ready( ) const
    {
        return stream[ 0 ] && stream[ 1 ] && stream[ 2 ];
    }
};
#define KSW_SIDE_WAVES 512u
// LDS bytes of a job's query window (ksw_q_lds) up to which it runs in the class's first tier: the register kernels'
// minimum per-wave LDS, so that 16 waves take 96 KB of a CU's 160 KB and the extension kernels' waves still fit beside them
#define KSW_Q_SPLIT KSW_REG_LDS
template <typename FETCH>
int ksw_run_all( const FETCH& F, const KswScoring& SC, u32 nSlots, const KswSizing& SZ, DevBuf& scratch,
                 unsigned int* next, KswOut O, hipStream_t stream, u32* lists = nullptr, u64 list_stride = 0,
                 unsigned int* nRedo = nullptr, unsigned int* nextBig = nullptr, const KswSide* side = nullptr )
{
    auto al = []( u64 x ) { return ( x + 255 ) / 256 * 256; };
    u64 nJobs = 0;
    for( int k = 0; k < KSW_N_CLASSES; k++ )
        nJobs += SZ.cls[ k ];
    if( nJobs == 0 )
        return 0;
    const u64 nGrp = SZ.cls[ 7 ] + SZ.cls[ 8 ] + SZ.cls[ 9 ] + SZ.cls[ 10 ] + SZ.cls[ 11 ] + SZ.cls[ 12 ];
    // jobs on the proven narrow band (ksw_band.h): those that fail a check are appended to the lists of k_ksw_ext<1> / <2>, which
    // run after them (next[ 17 ], next[ 18 ] count them)
    const u64 nBand = SC.grp >= 1000 ? SZ.cls[ 7 ] + SZ.cls[ 8 ] : 0;
    const u64 nBandL = SZ.cls[ KSW_CLS_BANDL ] + SZ.cls[ KSW_CLS_BANDL + 1 ]; // long jobs on the band of 120; those that fail go to `redo`
    const u64 nExt = SZ.cls[ 5 ] + SZ.cls[ 6 ] + nGrp + nBandL;
    const bool conc = side && side->ready( ) && lists; // classes on their own streams
    u64 perCu = 32; // waves per CU of the persistent ksw launches (MA_KSW_WAVES_PER_CU: tuning hook)
    if( const char* e = getenv( "MA_KSW_WAVES_PER_CU" ) )
        perCu = (u64)std::max( 1, atoi( e ) );
    const u64 wantWaves = 256ull * perCu;
    // Concurrent launches each own a scratch region, so each asks only for the waves that can be RESIDENT (registers:
    // k_ksw_pk<1> 74 VGPRs = 6 waves per SIMD, <2> 99, <3> 122, <5> 128 = 4; the extension kernels 7 / 4): a persistent wave
    // beyond that would only start when another one retires and its scratch would sit idle until then.
    const u64 resident[ 7 ] = { 256 * 24, 256 * 16, 256 * 16, 256 * 16, 0, 256 * 28, 256 * 16 };
    auto wantOf = [ & ]( int k ) { return conc ? std::min<u64>( wantWaves, resident[ k ] ) : wantWaves; };
    // budgets of the scratch regions: lane 0 / 1 / 2 (sequential launches of a lane share its region)
    const u64 B = KSW_SCRATCH_BUDGET; // one reading per call
    const u64 budgetOf[ 7 ] = { conc ? B / 3 : B, conc ? B / 3 : B, conc ? B / 3 : B, conc ? 5 * B / 12 : B, B, conc ? B / 4 : B, conc ? B / 4 : B };
    auto ldsOf = []( u64 qBytes ) { return std::max<u32>( (u32)( ( ( std::min<u64>( qBytes, 150000 + 64 ) + 15 ) / 16 ) * 16 ), KSW_REG_LDS ); };
    // pass 0 launches of the register kernels: classes 0..3 (exact), 5 / 6 (extension); [7..10]: classes 0..3 of the
    // second pass (jobs handed back), [11..14]: the huge tier of classes 0..3
    KswLaunchPlan LP[ 15 ];
    u64 pSplit[ 4 ] = { 0, 0, 0, 0 };
    // k_ksw_grp: fixed scratch per wave (KSW_GRP_ROWS direction rows; the cigars stay in LDS), 5 waves per SIMD
    KswLaunchPlan LG;
    if( nGrp )
    {
        const u64 sets = ( SZ.cls[ 7 ] + 1 ) / ( SC.grp >= 1000 ? 4 : 2 ) + ( SZ.cls[ 8 ] + 1 ) / ( SC.grp >= 1000 ? 4 : 2 ) + ( SZ.cls[ 9 ] + 1 ) / 2 + ( SZ.cls[ 10 ] + 1 ) / 2 + ( SZ.cls[ 11 ] + 3 ) / 4 + ( SZ.cls[ 12 ] + 3 ) / 4;
        // (direction rows of 256 B when the four-rows-per-lane lists have jobs)
        LG = ksw_plan_launch( (u64)KSW_GRP_ROWS * ( SC.grp < 1000 && SZ.cls[ 7 ] + SZ.cls[ 8 ] ? 256 : 128 ), 0, sets, std::min<u64>( wantWaves, 256 * 20 ), conc ? B / 4 : B );
    }
    KswLaunchPlan LBL; // k_ksw_band<.., 1>: KSW_BANDL_ROWS direction rows per wave, 5 waves per SIMD
    // The few long jobs the band hands back (0.1 % of a 10 kb batch, and the long ones among them: queries AND targets of thousands of
    // bases) take tens of ms each in k_ksw_pk<5>.  Behind the extension kernels on the batch's stream they were a tail of their own
    // (10 kb: 38 of 161 ms); with streams they get their own list and start on lane 3 the moment the band kernels are done.
    const bool ownRedo = nBandL && conc && side->band;
    KswLaunchPlan LPR;
    if( nBandL )
    {
        // rows for the largest job of the batch, but no more than leave a full set of waves their scratch: the few larger jobs are handed
        // on by the kernel (a 7 900 x 7 900 job needs 2 MB of direction rows; sized for it, a launch of 10^5 jobs of 2 500 x 1 000 would
        // run on half its waves)
        const u64 wavesL = std::min<u64>( std::min<u64>( wantWaves, 256 * 20 ), nBandL );
        const u64 rowsFull = std::max<u64>( KSW_BANDL_ROWS( 2048 ), ( ( conc ? B / 4 : B ) / std::max<u64>( wavesL, 1 ) ) / 128 );
        const u64 rows = std::min<u64>( KSW_BANDL_ROWS( SZ.bandlN ? std::min<u64>( SZ.bandlN, KSW_BANDL_NMAX ) : KSW_BANDL_NMAX ), rowsFull );
        LBL = ksw_plan_launch( rows * 128, 0, nBandL, wavesL, conc ? B / 4 : B );
    }
    if( ownRedo )
    {
        LPR = ksw_plan_launch( SZ.pRedo ? SZ.pRedo : SZ.p, SZ.cigRedo ? SZ.cigRedo : SZ.cig, std::min<u64>( nBandL / 4 + 64, 256 * 8 ), wantWaves, B / 4 );
        LPR.lds = ldsOf( std::min<u64>( SZ.qlen, SZ.cigRedo ? SZ.cigRedo : SZ.qlen ) + 48 );
    }
    for( int k = 0; k < 7; k++ )
    {
        if( k == 4 || ( SZ.cls[ k ] == 0 && !( k >= 5 && nBand ) ) )
            continue;
        const u64 pk = SZ.pc[ k ] ? SZ.pc[ k ] : ( k >= 5 ? std::max( SZ.p, ksw_ext_p_bytes( 256, 2048, k - 4 ) ) : SZ.p );
        const u64 cg = SZ.cigc[ k ] ? SZ.cigc[ k ] : SZ.cig;
        // (the narrow band hands on a few per cent of its jobs: the extension kernels' waves are planned for their own jobs plus a
        // sixteenth of the band's -- more of them would only wait their turn)
        LP[ k ] = ksw_plan_launch( pk, cg, SZ.cls[ k ] + ( k >= 5 && nBand ? nBand / 16 + 64 : 0 ), wantOf( k ), budgetOf[ k ] );
        if( k >= 5 )
        {
            LP[ k ].lds = KSW_EXT_LDS;
            continue;
        }
        // LDS bytes of the longest query of the class (ksw_q_lds <= round16( qlen ) + 32; qlen + tlen + 2 <= cigc[k])
        const u64 qMaxK = std::min<u64>( SZ.qlen, SZ.cigc[ k ] ? SZ.cigc[ k ] : SZ.qlen ) + 48;
        LP[ k ].lds = ldsOf( qMaxK );
        const u64 full = std::min<u64>( wantOf( k ), SZ.cls[ k ] );
        const bool fewWaves = LP[ k ].waves < full && LP[ k ].waves < 256 * 16;
        // The reversed query of a job lives in LDS, sized for the longest job of the LAUNCH: one 49 kb end extension in a
        // class of 10^5 gap fills would leave 3 waves per CU to all of them (this, not the tails, held the exact kernels of the
        // 50 kb workload at a fifth of the issue rate).  Jobs whose query or direction matrix does not leave room for a full
        // set of waves go to the class's second tier: few waves, own scratch, own stream.
        const bool splitQ = qMaxK > KSW_Q_SPLIT;
        bool splitP = false;
        u64 pSmall = pk;
        if( fewWaves )
        {
            // direction bytes one wave of a full set can have, after its cigar scratch (a job of the small tier has
            // qlen + tlen <= pSmall / 16 + 1: a diagonal stores 16 bytes at least)
            const u64 room = budgetOf[ k ] / std::min<u64>( full, 256 * 16 );
            if( room > 8192 )
            {
                pSmall = std::min<u64>( pk, ( room - room / 5 - 512 ) / 256 * 256 ); // stride = p + 4 * (p / 16 + 3) + padding
                splitP = pSmall < pk;
            }
        }
        if( nextBig && ( splitP || splitQ ) )
        {
            LP[ 11 + k ] = LP[ k ]; // the huge jobs: what the small tier leaves
            LP[ 11 + k ].tier = 2;
            if( conc ) // their region is a fixed quarter of the budget (a region that follows the largest job of every batch
                       // would be re-allocated -- seconds, device-wide -- whenever a later batch brings a larger one)
                LP[ 11 + k ].waves = (u32)std::max<u64>( 1, std::min<u64>( std::min<u32>( LP[ 11 + k ].waves, KSW_SIDE_WAVES ),
                                                                        ( B / 4 ) / std::max<u64>( LP[ 11 + k ].stride, 1 ) ) );
            const u64 cgSmall = splitP ? std::min<u64>( cg, pSmall / 16 + 3 ) : cg;
            LP[ k ] = ksw_plan_launch( pSmall, cgSmall, SZ.cls[ k ], wantOf( k ), budgetOf[ k ] );
            LP[ k ].tier = 1;
            LP[ k ].lds = ldsOf( std::min<u64>( qMaxK, KSW_Q_SPLIT ) );
            pSplit[ k ] = pSmall;
        }
    }
    if( nExt )
        for( int k = 0; k < 4; k++ )
        {
            // (the band of 120 proves nearly all of its jobs: a quarter of them as waves of the second pass is plenty)
            LP[ 7 + k ] = ksw_plan_launch( SZ.pRedo ? SZ.pRedo : SZ.p, SZ.cigRedo ? SZ.cigRedo : SZ.cig,
                                           std::min<u64>( nExt, 256 * 4 ) + std::min<u64>( nBandL / 4, 256 * 12 ), wantWaves, conc ? B / 4 : B );
            LP[ 7 + k ].lds = ldsOf( std::min<u64>( SZ.qlen, SZ.cigRedo ? SZ.cigRedo : SZ.qlen ) + 48 );
        }
    KswPlan plan = ksw_plan( SZ, SZ.cls[ 4 ] ? SZ.cls[ 4 ] : 1, conc ? B / 4 : B );
    // scratch regions.  Sequential mode: one region for all launches (+ one for the huge tiers when they have a stream).
    // Concurrent mode: lane 0 = extension classes, handed-back jobs, LDS kernel; lane 1 = class 3; lane 2 = classes 0..2;
    // lane 3 = huge tiers.
    auto laneOf = [ & ]( int i ) -> int {
        if( !conc )
            return i >= 11 && side && side->stream[ 2 ] ? 3 : 0;
        if( i >= 11 )
            return 3;
        if( i == 3 )
            return 1;
        if( i <= 2 )
            return 2;
        return 0;
    };
    u64 needLane[ 4 ] = { std::max<u64>( std::max<u64>( SZ.cls[ 4 ] ? plan.ws.stride * plan.waves : 0, LG.stride * LG.waves ), LBL.stride * LBL.waves ), 0, 0, 0 };
    for( int i = 0; i < 15; i++ )
        if( LP[ i ].waves )
            needLane[ laneOf( i ) ] = std::max<u64>( needLane[ laneOf( i ) ], LP[ i ].stride * LP[ i ].waves );
    if( ownRedo )
        needLane[ 3 ] = std::max<u64>( needLane[ 3 ], LPR.stride * LPR.waves );
    // the per-wave scratch follows the largest jobs of the batch, which vary a lot from batch to batch for long
    // reads: once it is in the GB range take the whole budget so that later batches never re-allocate mid-step
    if( !conc && needLane[ 0 ] > ( 2ull << 30 ) )
        needLane[ 0 ] = std::max<u64>( needLane[ 0 ], B );
    if( conc && needLane[ 0 ] + needLane[ 1 ] + needLane[ 2 ] + needLane[ 3 ] > ( 2ull << 30 ) )
    {
        needLane[ 0 ] = std::max<u64>( needLane[ 0 ], B / 4 );
        needLane[ 1 ] = std::max<u64>( needLane[ 1 ], 5 * B / 12 );
        needLane[ 2 ] = std::max<u64>( needLane[ 2 ], B / 3 );
        needLane[ 3 ] = std::max<u64>( needLane[ 3 ], B / 4 );
    }
    u64 laneBase[ 4 ], total = 0;
    for( int l = 0; l < 4; l++ )
    {
        laneBase[ l ] = total;
        total += al( needLane[ l ] );
    }
    if( scratch.reserve( total ) )
        return 1;
    u32 ldsMax = LPR.lds;
    for( int i = 0; i < 15; i++ )
        if( i != 5 && i != 6 )
            ldsMax = std::max( ldsMax, LP[ i ].lds );
    if( ldsMax > 48 * 1024 )
    {
        MA_HIP( hipFuncSetAttribute( (const void*)k_ksw_pk<FETCH, KSW_S0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsMax ) );
        MA_HIP( hipFuncSetAttribute( (const void*)k_ksw_pk<FETCH, KSW_S1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsMax ) );
        MA_HIP( hipFuncSetAttribute( (const void*)k_ksw_pk<FETCH, KSW_S2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsMax ) );
        MA_HIP( hipFuncSetAttribute( (const void*)k_ksw_pk<FETCH, KSW_S3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsMax ) );
    }
    uint8_t* base = scratch.as<uint8_t>( );
    u32* redo = lists ? lists + (u64)KSW_N_CLASSES * list_stride : nullptr;
    u32* redoL = lists ? lists + (u64)( KSW_N_CLASSES + 1 ) * list_stride : nullptr; // (ownRedo: the long jobs the band of 120 handed back)
    hipStream_t laneStream[ 4 ] = { stream, conc ? side->stream[ 0 ] : stream, conc ? side->stream[ 1 ] : stream,
                                    side && side->stream[ 2 ] ? side->stream[ 2 ] : stream };
    bool laneUsed[ 4 ] = { true, false, false, false };
    for( int i = 0; i < 15; i++ )
        if( LP[ i ].waves && laneStream[ laneOf( i ) ] != stream )
            laneUsed[ laneOf( i ) ] = true;
    if( ownRedo )
        laneUsed[ 3 ] = true;
    const bool forked = laneUsed[ 1 ] || laneUsed[ 2 ] || laneUsed[ 3 ];
    if( forked )
    {
        MA_HIP( hipEventRecord( side->fork, stream ) );
        for( int l = 1; l < 4; l++ )
            if( laneUsed[ l ] )
                MA_HIP( hipStreamWaitEvent( laneStream[ l ], side->fork, 0 ) );
    }
    // pass 0: the classes' own jobs; pass 1: whatever the extension kernel handed back (usually nothing); pass 2: the huge
    // tier of a class that was split
    auto launchPk = [ & ]( int pass, int k ) {
        const int i = pass == 0 ? k : ( pass == 1 ? 7 + k : 11 + k );
        const KswLaunchPlan& L = LP[ i ];
        if( L.waves == 0 )
            return;
        KswJobs JB;
        JB.list = pass == 1 ? redo : ( lists ? lists + (u64)k * list_stride : nullptr );
        JB.n = lists ? (u32)SZ.cls[ k ] : nSlots;
        JB.nDev = nRedo;
        JB.mode = pass == 1 ? 2 : ( lists ? 0 : 1 );
        JB.cls = k;
        JB.tier = pass == 1 ? 0 : L.tier;
        JB.pSplit = pSplit[ k ];
        JB.qSplit = KSW_Q_SPLIT;
        unsigned int* nx = pass == 0 ? next + k : ( pass == 1 ? next + 7 + k : nextBig + k );
        hipStream_t st = laneStream[ laneOf( i ) ];
        uint8_t* sbase = base + laneBase[ laneOf( i ) ];
        switch( k )
        {
        case 0:
            hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S0> ), dim3( L.waves ), dim3( 64 ), L.lds, st, F, SC, JB, nx, sbase, L.stride, L.p_cap, L.lds, O );
            break;
        case 1:
            hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S1> ), dim3( L.waves ), dim3( 64 ), L.lds, st, F, SC, JB, nx, sbase, L.stride, L.p_cap, L.lds, O );
            break;
        case 2:
            hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S2> ), dim3( L.waves ), dim3( 64 ), L.lds, st, F, SC, JB, nx, sbase, L.stride, L.p_cap, L.lds, O );
            break;
        default:
            hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S3> ), dim3( L.waves ), dim3( 64 ), L.lds, st, F, SC, JB, nx, sbase, L.stride, L.p_cap, L.lds, O );
        }
    };
    // longest jobs first: the huge tiers, then the wide classes; the extension kernels' small jobs fill the tails
    for( int k = 3; k >= 0; k-- )
        launchPk( 2, k );
    if( conc )
        for( int k = 3; k >= 0; k-- )
            launchPk( 0, k );
    if( nBandL ) // the long extensions on the band of 120 first: they are the longest jobs of lane 0
        for( int k = 0; k < 2; k++ )
        {
            const u32 nk = (u32)SZ.cls[ KSW_CLS_BANDL + k ];
            if( nk == 0 )
                continue;
            const u32* lk = lists + (u64)( KSW_CLS_BANDL + k ) * list_stride;
            const u32 waves = (u32)std::max<u64>( 1, std::min<u64>( LBL.waves, nk ) );
            uint8_t* sb = base + laneBase[ 0 ];
            if( k == 0 )
                hipLaunchKernelGGL( ( k_ksw_band<FETCH, true, 1> ), dim3( waves ), dim3( 64 ), 0, stream, F, SC, lk, nk, next + 19 + k, sb, LBL.stride, O,
                                    ownRedo ? redoL : redo, 0u, (u32*)nullptr, 0u, ownRedo ? next + 25 : nRedo );
            else
                hipLaunchKernelGGL( ( k_ksw_band<FETCH, false, 1> ), dim3( waves ), dim3( 64 ), 0, stream, F, SC, lk, nk, next + 19 + k, sb, LBL.stride, O,
                                    ownRedo ? redoL : redo, 0u, (u32*)nullptr, 0u, ownRedo ? next + 25 : nRedo );
        }
    if( ownRedo )
    {
        MA_HIP( hipEventRecord( side->band, stream ) );
        MA_HIP( hipStreamWaitEvent( laneStream[ 3 ], side->band, 0 ) );
        for( int k = 3; k >= 0; k-- )
        {
            KswJobs JB;
            JB.list = redoL;
            JB.n = 0;
            JB.nDev = next + 25;
            JB.mode = 2;
            JB.cls = k;
            JB.tier = 0;
            JB.pSplit = 0;
            JB.qSplit = KSW_Q_SPLIT;
            uint8_t* sb = base + laneBase[ 3 ];
            switch( k )
            {
            case 0:
                hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S0> ), dim3( LPR.waves ), dim3( 64 ), LPR.lds, laneStream[ 3 ], F, SC, JB, next + 21 + k, sb, LPR.stride, LPR.p_cap, LPR.lds, O );
                break;
            case 1:
                hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S1> ), dim3( LPR.waves ), dim3( 64 ), LPR.lds, laneStream[ 3 ], F, SC, JB, next + 21 + k, sb, LPR.stride, LPR.p_cap, LPR.lds, O );
                break;
            case 2:
                hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S2> ), dim3( LPR.waves ), dim3( 64 ), LPR.lds, laneStream[ 3 ], F, SC, JB, next + 21 + k, sb, LPR.stride, LPR.p_cap, LPR.lds, O );
                break;
            default:
                hipLaunchKernelGGL( ( k_ksw_pk<FETCH, KSW_S3> ), dim3( LPR.waves ), dim3( 64 ), LPR.lds, laneStream[ 3 ], F, SC, JB, next + 21 + k, sb, LPR.stride, LPR.p_cap, LPR.lds, O );
            }
        }
    }
    if( nGrp ) // the short extensions, several per wavefront (lane 0 of the streams, like the other extension kernels); longest first
        for( int k = 0; k < KSW_GRP_LISTS; k++ )
        {
            const u32 nk = (u32)SZ.cls[ KSW_CLS_GRP0 + k ];
            if( nk == 0 )
                continue;
            const u32* lk = lists + (u64)( KSW_CLS_GRP0 + k ) * list_stride;
            const u32 Gk = k < 2 && SC.grp >= 1000 ? 4u : ( k < 4 ? 2u : 4u );
            const u32 waves = (u32)std::max<u64>( 1, std::min<u64>( LG.waves, ( nk + Gk - 1 ) / Gk ) );
            uint8_t* sb = base + laneBase[ 0 ];
            switch( k )
            {
            case 0:
                if( SC.grp >= 1000 )
                {
                    hipLaunchKernelGGL( ( k_ksw_band<FETCH, true> ), dim3( waves ), dim3( 64 ), 0, stream, F, SC, lk, nk, next + 11 + k, sb, LG.stride, O,
                                        lists + 5 * list_stride, (u32)SZ.cls[ 5 ], lists + 6 * list_stride, (u32)SZ.cls[ 6 ], next + 17 );
                    break;
                }
#if defined( MA_EXP_GRP_NR2 ) // experiment build only (make expx X=-DMA_EXP_GRP_NR2): two jobs of 65..128 bases per wave, four rows per lane
                hipLaunchKernelGGL( ( k_ksw_grp<FETCH, 2, 2, true> ), dim3( waves ), dim3( 64 ), 0, stream, F, SC, lk, nk, next + 11 + k, sb, LG.stride, O, redo, nRedo );
#endif
                break;
            case 1:
                if( SC.grp >= 1000 )
                {
                    hipLaunchKernelGGL( ( k_ksw_band<FETCH, false> ), dim3( waves ), dim3( 64 ), 0, stream, F, SC, lk, nk, next + 11 + k, sb, LG.stride, O,
                                        lists + 5 * list_stride, (u32)SZ.cls[ 5 ], lists + 6 * list_stride, (u32)SZ.cls[ 6 ], next + 17 );
                    break;
                }
#if defined( MA_EXP_GRP_NR2 )
                hipLaunchKernelGGL( ( k_ksw_grp<FETCH, 2, 2, false> ), dim3( waves ), dim3( 64 ), 0, stream, F, SC, lk, nk, next + 11 + k, sb, LG.stride, O, redo, nRedo );
#endif
                break;
            case 2:
                hipLaunchKernelGGL( ( k_ksw_grp<FETCH, 2, 1, true> ), dim3( waves ), dim3( 64 ), 0, stream, F, SC, lk, nk, next + 11 + k, sb, LG.stride, O, redo, nRedo );
                break;
            case 3:
                hipLaunchKernelGGL( ( k_ksw_grp<FETCH, 2, 1, false> ), dim3( waves ), dim3( 64 ), 0, stream, F, SC, lk, nk, next + 11 + k, sb, LG.stride, O, redo, nRedo );
                break;
            case 4:
                hipLaunchKernelGGL( ( k_ksw_grp<FETCH, 4, 1, true> ), dim3( waves ), dim3( 64 ), 0, stream, F, SC, lk, nk, next + 11 + k, sb, LG.stride, O, redo, nRedo );
                break;
            default:
                hipLaunchKernelGGL( ( k_ksw_grp<FETCH, 4, 1, false> ), dim3( waves ), dim3( 64 ), 0, stream, F, SC, lk, nk, next + 11 + k, sb, LG.stride, O, redo, nRedo );
            }
        }
    if( SZ.cls[ 5 ] || nBand )
        hipLaunchKernelGGL( ( k_ksw_ext<FETCH, 1> ), dim3( LP[ 5 ].waves ), dim3( 64 ), KSW_EXT_LDS, stream, F, SC,
                            lists + 5 * list_stride, (u32)SZ.cls[ 5 ], next + 5, base + laneBase[ 0 ], LP[ 5 ].stride, LP[ 5 ].p_cap, KSW_EXT_LDS,
                            O, redo, nRedo, nBand ? next + 17 : nullptr );
    if( SZ.cls[ 6 ] || nBand )
        hipLaunchKernelGGL( ( k_ksw_ext<FETCH, 2> ), dim3( LP[ 6 ].waves ), dim3( 64 ), KSW_EXT_LDS, stream, F, SC,
                            lists + 6 * list_stride, (u32)SZ.cls[ 6 ], next + 6, base + laneBase[ 0 ], LP[ 6 ].stride, LP[ 6 ].p_cap, KSW_EXT_LDS,
                            O, redo, nRedo, nBand ? next + 18 : nullptr );
    if( !conc )
        for( int k = 0; k < 4; k++ )
            launchPk( 0, k );
    if( SZ.cls[ 4 ] ) // a handed-back job always fits a register kernel
    {
        KswJobs JB;
        JB.list = lists ? lists + (u64)4 * list_stride : nullptr;
        JB.n = lists ? (u32)SZ.cls[ 4 ] : nSlots;
        JB.nDev = nRedo;
        JB.mode = lists ? 0 : 1;
        JB.cls = 4;
        JB.tier = 0;
        JB.pSplit = 0;
        JB.qSplit = 0;
        plan.ws.base = base + laneBase[ 0 ];
        if( plan.lds_bytes > 48 * 1024 )
            MA_HIP( hipFuncSetAttribute( (const void*)k_ksw<FETCH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)plan.lds_bytes ) );
        hipLaunchKernelGGL( k_ksw<FETCH>, dim3( plan.waves ), dim3( 64 ), plan.lds_bytes, stream, F, SC, JB, next + 4,
                            plan.ws, plan.lds_bytes, O );
    }
    if( nExt )
        for( int k = 0; k < 4; k++ )
            launchPk( 1, k ); // on the batch's stream, behind the extension kernels that fill the list
    if( forked )
        for( int l = 1; l < 4; l++ )
            if( laneUsed[ l ] )
            {
                MA_HIP( hipEventRecord( side->join[ l - 1 ], laneStream[ l ] ) );
                MA_HIP( hipStreamWaitEvent( stream, side->join[ l - 1 ], 0 ) );
            }
    MA_HIP( hipGetLastError( ) );
    return 0;
}
} // namespace ma
