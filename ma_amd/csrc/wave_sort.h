// wave_sort.h -- libstdc++'s std::sort (GCC 11: introsort, bits/stl_algo.h) run by ONE WAVEFRONT on an array in LDS, with the
// permutation std::sort itself produces.  The chaining stage sorts a long read's seeds twice (by delta, by reference
// position: stripOfConsideration.cpp:33, soc.h:213); ties are common, std::sort is not stable, and what follows depends on
// the order of equal elements -- so the SAME algorithm has to run, only faster than one lane chasing its own array
// through global memory (stdsort.h: 55 % of k_chain for 50 kb reads, latency-bound).
//
// What is parallel here:
//   __unguarded_partition (Hoare): the serial loop swaps the k-th element from the left that is not < pivot ("left stopper")
//     with the k-th element from the right that is not > pivot ("right stopper") for as long as the former lies left of the
//     latter.  Both stopper sequences can be read off the UNPARTITIONED range (a swap only touches positions both scans have
//     passed), so: list the stoppers of both sides with wave ballots, count the pairs K that still cross, swap them all at
//     once; the cut is the (K+1)-th left stopper if that lies below the K-th right one, else the K-th right stopper (where
//     the serial left scan runs into the element the last swap put there).  tools-checked against the serial loop on 2*10^5
//     random ranges with heavy ties, and against stdsort.h / libstdc++ by tests/test_host_logic.py + the GPU parity tests.
//   __final_insertion_sort: after the introsort loop the array consists of blocks of <= 16 elements, every element of a
//     block <= every element of the next; the one insertion sort over everything is a stable sort of each block in place
//     (strict comparisons never move an element across a block boundary) -- so it can be done range by range.
//   ranges of <= WS_SERIAL elements: the rest of the loop and the insertion sort of 64 such ranges at a time, one per lane.
// What stays serial (wave-uniform, all lanes redundantly): the loop over the large ranges, median-of-three, the recursion
// stack; the heap sort of a range whose depth budget ran out (lane 0; practically never).
// Elements are u64 with the order LESS( a, b ); the caller packs (key << 20 | index).
#pragma once
#include "internal.h"
#include "stdsort.h"

#if defined( __HIPCC__ )
namespace ma
{
namespace ws
{
struct Scratch // all in LDS
{
    u64* a; // n elements
    uint16_t* sl; // n + 1 left stoppers
    uint16_t* sr; // n + 1 right stoppers
    u32* blockBits; // the list of small ranges: 2 words per item
    i32 itemCap;
    i32* stack; // 3 * 64 ints
};
MA_HD u32 item_cap( u32 n ) // ranges of <= WS_SERIAL elements the loop can leave: two per wave-wide partition
{
    return n / 8 + 128;
}
MA_HD u64 scratch_bytes( u32 n )
{
    return (u64)n * 8 + 2 * ( (u64)n + 8 ) * 2 + (u64)item_cap( n ) * 8 + 3 * 64 * 4 + 64;
}
__device__ __forceinline__ Scratch carve( uint8_t* lds, u32 n )
{
    Scratch S;
    S.a = (u64*)lds;
    uint8_t* p = lds + (u64)n * 8;
    S.sl = (uint16_t*)p;
    p += ( ( (u64)n + 2 ) * 2 + 7 ) / 8 * 8;
    S.sr = (uint16_t*)p;
    p += ( ( (u64)n + 2 ) * 2 + 7 ) / 8 * 8; // (scratch_bytes leaves 6 elements of slack per list for the rounding)
    S.blockBits = (u32*)p;
    S.itemCap = (i32)item_cap( n );
    p += (u64)item_cap( n ) * 8;
    S.stack = (i32*)( ( (uintptr_t)p + 7 ) & ~(uintptr_t)7 );
    return S;
}

// Hoare partition of [lo, hi) around the pivot a[first] (first = lo - 1), all 64 lanes; returns the cut (wave-uniform)
template <typename LESS> __device__ i32 wave_partition( const Scratch& S, i32 first, i32 hi, LESS less )
{
    const int lane = threadIdx.x & 63;
    const unsigned long long below = ( 1ull << lane ) - 1;
    u64* a = S.a;
    const i32 lo = first + 1;
    const u64 pv = a[ first ];
    i32 nL = 0, nR = 0;
    for( i32 base = lo; base < hi; base += 64 )
    {
        const i32 i = base + lane;
        const bool f = i < hi && !less( a[ i ], pv );
        const unsigned long long m = __ballot( f );
        if( f )
            S.sl[ nL + __popcll( m & below ) ] = (uint16_t)i;
        nL += __popcll( m );
    }
    for( i32 top = hi - 1; top >= first; top -= 64 ) // descending, down to the pivot's own place (the scan's last resort)
    {
        const i32 j = top - lane;
        const bool f = j >= first && !less( pv, a[ j ] );
        const unsigned long long m = __ballot( f );
        if( f )
            S.sr[ nR + __popcll( m & below ) ] = (uint16_t)j;
        nR += __popcll( m );
    }
    __syncthreads( );
    const i32 nPairs = nL < nR ? nL : nR;
    i32 K = 0;
    for( i32 base = 0; base < nPairs; base += 64 )
    {
        const i32 k = base + lane;
        const bool crossing = k < nPairs && S.sl[ k ] < S.sr[ k ];
        const unsigned long long m = __ballot( crossing );
        K += __popcll( m );
        if( m != ~0ull )
            break; // the sequences cross once
    }
    for( i32 k = lane; k < K; k += 64 )
    {
        const i32 i = S.sl[ k ], j = S.sr[ k ];
        const u64 t = a[ i ];
        a[ i ] = a[ j ];
        a[ j ] = t;
    }
    const i32 prevR = K > 0 ? (i32)S.sr[ K - 1 ] : hi;
    const i32 cut = K < nL && (i32)S.sl[ K ] < prevR ? (i32)S.sl[ K ] : prevR;
    __syncthreads( );
    return cut;
}

// std::sort( a, a + n, less ) by one wavefront (blockDim = 64); n <= 65535.
// Ranges of more than WS_SERIAL elements are partitioned by the whole wave; what the loop leaves of them -- ranges of
// <= WS_SERIAL elements with the depth budget they were reached with -- goes on a list and is finished one range per lane
// (stdsort.h: finish_range): a wave-wide partition of 20 elements costs as much as one of 64, and its latency (LDS round
// trips, ballots, barriers) is paid serially, whereas 64 lanes finish 64 small ranges side by side.
#define WS_SERIAL 128
template <typename LESS> __device__ void wave_std_sort( const Scratch& S, i32 n, LESS less )
{
    const int lane = threadIdx.x & 63;
    u64* a = S.a;
    if( n <= 1 )
        return;
    // the list of small ranges lives where the stopper lists of the LAST partitions no longer reach: behind them.  Items:
    // first | last << 16 (u32), depth (u32)
    u32* items = S.blockBits;
    const i32 itemCap = S.itemCap;
    i32 nItems = 0;
    auto defer = [ & ]( i32 first, i32 last, i32 depth ) {
        if( last - first <= 1 )
            return;
        if( nItems < itemCap )
        {
            if( lane == 0 )
            {
                items[ 2 * nItems ] = (u32)first | (u32)last << 16;
                items[ 2 * nItems + 1 ] = (u32)depth;
            }
            nItems++;
        }
        else
        {
            if( lane == 0 ) // list full (cannot happen with the capacity carve() gives it): finish it right here
                ss::finish_range( a, (i64)first, (i64)last, (i64)depth, less );
            __syncthreads( );
        }
    };
    // ---- __introsort_loop on the large ranges: the right-hand part goes on a stack, the loop continues on the left-hand one
    i32 sp = 0, first = 0, last = n, depth = 0;
    for( u32 m = (u32)n; m >>= 1; )
        depth += 2; // 2 * __lg( n )
    while( true )
    {
        while( last - first > WS_SERIAL )
        {
            if( depth == 0 )
            {
                if( lane == 0 )
                    ss::heap_sort_range( a, (i64)first, (i64)last, less );
                __syncthreads( );
                first = last; // done (a sorted range needs no final insertion)
                break;
            }
            --depth;
            const i32 mid = first + ( last - first ) / 2;
            if( lane == 0 )
                ss::move_median_to_first( a, (i64)first, (i64)first + 1, (i64)mid, (i64)last - 1, less );
            __syncthreads( );
            const i32 cut = wave_partition( S, first, last, less );
            if( lane == 0 )
            {
                S.stack[ 3 * sp ] = cut;
                S.stack[ 3 * sp + 1 ] = last;
                S.stack[ 3 * sp + 2 ] = depth;
            }
            sp++;
            last = cut;
        }
        defer( first, last, depth );
        if( sp == 0 )
            break;
        sp--;
        __syncthreads( );
        first = S.stack[ 3 * sp ];
        last = S.stack[ 3 * sp + 1 ];
        depth = S.stack[ 3 * sp + 2 ];
    }
    __syncthreads( );
    // ---- the small ranges, one per lane
    for( i32 k = lane; k < nItems; k += 64 )
    {
        const u32 fl = items[ 2 * k ];
        ss::finish_range( a, (i64)( fl & 0xffffu ), (i64)( fl >> 16 ), (i64)items[ 2 * k + 1 ], less );
    }
    __syncthreads( );
}
} // namespace ws
} // namespace ma
#endif
