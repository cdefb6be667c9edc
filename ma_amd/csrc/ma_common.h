// ma_common.h -- shared host/device definitions of the MI355X seed-and-extend engine.
// All stage logic lives in MA_HD functions so that the very same code can be exercised on the CPU
// by tests/host_emul (logic tests, "-m 'not gpu'") while the shipped path runs it inside HIP kernels.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "../../include/ma_amd.h"

#if defined( __HIPCC__ )
#include <hip/hip_runtime.h>
#define MA_HD __host__ __device__ __forceinline__
#define MA_HD_NOINLINE __host__ __device__
#define MA_HD_OUTLINE inline __host__ __device__ __attribute__( ( noinline ) ) // one copy of a big helper per kernel
#else
#define MA_HD inline
#define MA_HD_NOINLINE
#define MA_HD_OUTLINE inline
#endif

typedef int64_t i64;
typedef uint64_t u64;
typedef uint32_t u32;
typedef int32_t i32;

namespace ma
{
#if defined( __HIP_DEVICE_COMPILE__ )
MA_HD int popc32( u32 x )
{
    return __popc( x );
}
MA_HD int popc64( u64 x )
{
    return __popcll( x );
}
#else
MA_HD int popc32( u32 x )
{
    return __builtin_popcount( x );
}
MA_HD int popc64( u64 x )
{
    return __builtin_popcountll( x );
}
#endif

template <typename T> MA_HD T mmin( T a, T b )
{
    return a < b ? a : b;
}
template <typename T> MA_HD T mmax( T a, T b )
{
    return a > b ? a : b;
}
template <typename T> MA_HD void mswap( T& a, T& b )
{
    T t = a;
    a = b;
    b = t;
}

// Device-resident index view (FMIndex fMIndex.h:195-230 + Pack pack.h:39-176).
// HBM layout: bwt = the reference's occ-injected word array, one 64-byte block per 128 nt:
// 4 x u64 cumulative A/C/G/T counts followed by 8 x u32 words of 16 nt (2 bit each, MSB first).
struct IndexView
{
    const u32* bwt; // 64-B aligned
    const i64* sa; // every 32nd SA row, sa[0] = -1 (the reference's .sa file)
    // A denser sample of the same suffix array for the hot path (null: use sa): the LF walk of bwt_sa ends at the first
    // sampled row it hits, a geometric wait -- mean 31 steps at interval 32, the longest of 2 M rows ~450 dependent loads,
    // which is what k_lf_walk's run time was.  Every 8th row costs 6.2 GB for GRCh38 (of 288 GB) and divides both by 4.
    const i64* sa_dense = nullptr;
    u32 sa_shift = 5; // log2 of the interval of the sample bwt_sa uses
    // bi-interval of every K-mer (null: none): entry[key] = init_interval( b0 ) extended by b1 .. b(K-1), key = b0 b1 .. in
    // base 4, packed like the SMEM list entries (two u64: 35-bit starts, 35-bit size).  An extension run that starts from a
    // single base (maxSpan: the first right run and the second left run of a centre) takes its first K-1 steps from here.
    const u64* kmer_tab = nullptr;
    u32 kmer_k = 0;
    const uint8_t* pac; // forward strand, 2 bit/base, MSB first
    const u64* cstart; // contig start offsets (forward strand)
    const u64* clen;
    u64 n; // forward + reverse length
    u64 F; // forward length
    i64 primary;
    u64 L2[ 5 ];
    i32 n_contigs;
};

// error / overflow flags raised by kernels (checked in ma_batch_sync)
enum : u32
{
    MA_ERR_SEG_OVERFLOW = 1u,
    MA_ERR_SEED_OVERFLOW = 2u,
    MA_ERR_STACK_OVERFLOW = 4u,
    MA_ERR_CIGAR_OVERFLOW = 8u,
    MA_ERR_OPS_OVERFLOW = 16u,
    MA_ERR_SCRATCH_OVERFLOW = 32u,
    MA_ERR_SMEM_OVERFLOW = 64u,
};
} // namespace ma
