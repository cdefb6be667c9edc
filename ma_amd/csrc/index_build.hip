// index_build.hip -- FMD-index + pack construction on the GPU; replaces FMIndex::build_FMIndex
// (fMIndex.cpp:152-391: BWT via is_bwt / bwtLarge, occ injection 204-264, SA sampling 266-314) and
// Pack::vAppendSequence (pack.h:586-698).  The BWT, its occ blocks and the sampled SA are canonical
// functions of the text T.revcomp(T), so any correct suffix sort reproduces the reference's bytes.
//
// Suffix sorting is an MSD multi-round radix sort sized for GRCh38 (n = 6.2e9 suffixes) in 288 GB:
//   round 0: suffixes are bucketed by their first K bases (K = 0 for small genomes, 2 for large) and
//            every bucket is sorted by a 63-bit key = next 29 bases (58 bit) + 5-bit "remaining
//            length" field that orders suffixes running into '$' before their padded twins;
//   round r: only suffixes still tied (repeats) are re-keyed 29 bases further down and sorted by
//            (group, key) with two stable radix passes, until no ties remain.
#include <memory>
#include "internal.h"
#include "fm_device.h"
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstdlib>
#include <vector>

using namespace ma;

namespace
{
struct TextView // T.revcomp(T) over the 2-bit forward pack
{
    const uint8_t* pac;
    u64 F, n;
    __device__ __forceinline__ u32 at( u64 p ) const
    {
        const u64 q = p < F ? p : n - 1 - p;
        const u32 b = ( pac[ q >> 2 ] >> ( ( ~q & 3 ) << 1 ) ) & 3;
        return p < F ? b : 3 - b;
    }
};

// 63-bit sort key of suffix p at depth d: 29 bases (zero padded past the end) + clamp(n-(p+d)+1, 0, 30)
__device__ __forceinline__ u64 suffix_key( const TextView& T, u64 p, u64 d )
{
    const u64 s = p + d;
    u64 k = 0;
    for( u32 i = 0; i < 29; i++ )
    {
        const u64 q = s + i;
        k = ( k << 2 ) | ( q < T.n ? (u64)T.at( q ) : 0ull );
    }
    i64 rem = (i64)T.n - (i64)s + 1;
    rem = rem < 0 ? 0 : ( rem > 30 ? 30 : rem );
    return ( k << 5 ) | (u64)rem;
}

__global__ void k_pack( const uint8_t* codes, u64 F, uint8_t* pac, unsigned long long* hist )
{
    const u64 byte = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 c[ 4 ] = { 0, 0, 0, 0 };
    if( byte < ( F + 3 ) / 4 )
    {
        u32 v = 0;
        for( u32 j = 0; j < 4; j++ )
        {
            const u64 p = byte * 4 + j;
            const u32 b = p < F ? ( codes[ p ] & 3 ) : 0;
            if( p < F )
                c[ b ]++;
            v |= b << ( ( 3 - j ) << 1 );
        }
        pac[ byte ] = (uint8_t)v;
    }
    __shared__ unsigned int sh[ 4 ];
    if( threadIdx.x < 4 )
        sh[ threadIdx.x ] = 0;
    __syncthreads( );
    for( int b = 0; b < 4; b++ )
    {
        u32 x = c[ b ];
        for( int m = 32; m >= 1; m >>= 1 )
            x += __shfl_xor( x, m, 64 );
        if( ( threadIdx.x & 63 ) == 0 && x )
            atomicAdd( &sh[ b ], x );
    }
    __syncthreads( );
    if( threadIdx.x < 4 && sh[ threadIdx.x ] )
        atomicAdd( &hist[ threadIdx.x ], (unsigned long long)sh[ threadIdx.x ] );
}

struct InBucket // suffixes whose first K bases spell `code` (bases past the end read as A)
{
    TextView T;
    u32 K, code;
    __device__ bool operator( )( const u64& p ) const
    {
        u32 c = 0;
        for( u32 i = 0; i < K; i++ )
        {
            const u64 q = p + i;
            c = ( c << 2 ) | ( q < T.n ? T.at( q ) : 0u );
        }
        return c == code;
    }
};

// one pass: how many suffixes fall into each first-K-bases bucket
__global__ void k_bucket_hist( TextView T, u32 K, unsigned long long* hist /* 4^K */ )
{
    __shared__ unsigned int sh[ 256 ];
    sh[ threadIdx.x ] = 0;
    __syncthreads( );
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for( u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p < T.n; p += stride )
    {
        u32 c = 0;
        for( u32 i = 0; i < K; i++ )
        {
            const u64 q = p + i;
            c = ( c << 2 ) | ( q < T.n ? T.at( q ) : 0u );
        }
        atomicAdd( &sh[ c ], 1u );
    }
    __syncthreads( );
    if( threadIdx.x < ( 1u << ( 2 * K ) ) && sh[ threadIdx.x ] )
        atomicAdd( &hist[ threadIdx.x ], (unsigned long long)sh[ threadIdx.x ] );
}

__global__ void k_keys( TextView T, const u64* pos, u64 m, u64 depth, u64* keys )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i < m )
        keys[ i ] = suffix_key( T, pos[ i ], depth );
}

// after sorting by (gid,key): mark run heads (new gid) and sub-group heads (new (gid,key))
__global__ void k_heads( const u64* gid, const u64* key, u64 m, u64* runHeadIdx, u64* subHeadIdx )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= m )
        return;
    const bool rh = i == 0 || gid[ i ] != gid[ i - 1 ];
    const bool sh = rh || key[ i ] != key[ i - 1 ];
    runHeadIdx[ i ] = rh ? i : 0;
    subHeadIdx[ i ] = sh ? i : 0;
}

// place suffixes, derive the refined group ids and tie flags
__global__ void k_place( const u64* gid, const u64* pos, const u64* runHead, const u64* subHead, u64 m, i64* SA,
                         u64* newGid, uint8_t* tied )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= m )
        return;
    const u64 g = gid[ i ];
    SA[ g + ( i - runHead[ i ] ) ] = (i64)pos[ i ];
    const u64 sh = subHead[ i ];
    newGid[ i ] = g + ( sh - runHead[ i ] );
    const bool head = sh == i;
    const bool nextHead = i + 1 == m || subHead[ i + 1 ] == i + 1;
    tied[ i ] = ( head && nextHead ) ? 0 : 1;
}

__global__ void k_iota( u64* a, u64 m )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i < m )
        a[ i ] = i;
}

__global__ void k_fill_u64( u64* a, u64 m, u64 v )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i < m )
        a[ i ] = v;
}

__global__ void k_find_primary( const i64* SA, u64 n, unsigned long long* primary )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i < n && SA[ i ] == 0 )
        *primary = i + 1; // row of the suffix that is the whole text
}

// '$'-removed BWT symbol i (fMIndex.cpp:186-201 contract of is_bwt): rows skip `primary`
__device__ __forceinline__ u32 bwt_symbol( const TextView& T, const i64* SA, u64 primary, u64 i )
{
    const u64 row = i + ( i >= primary ? 1 : 0 ); // row index in [0,n], row != primary
    if( row == 0 )
        return T.at( T.n - 1 );
    const u64 s = (u64)SA[ row - 1 ];
    return T.at( s - 1 );
}

// per 128-symbol block: 8 packed words + symbol histogram
__global__ void k_bwt_blocks( TextView T, const i64* SA, u64 primary, u64 nblk, u32* words /*nblk*8*/,
                              u64* hA, u64* hC, u64* hG, u64* hT )
{
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; // one thread per 16-symbol word
    const u64 blk = w >> 3;
    u32 c[ 4 ] = { 0, 0, 0, 0 };
    if( blk < nblk )
    {
        u32 v = 0;
        for( u32 j = 0; j < 16; j++ )
        {
            const u64 i = w * 16 + j;
            if( i < T.n )
            {
                const u32 b = bwt_symbol( T, SA, primary, i );
                c[ b ]++;
                v |= b << ( ( 15 - j ) << 1 );
            }
        }
        words[ w ] = v;
    }
    // 8 consecutive lanes form one block
    for( int b = 0; b < 4; b++ )
    {
        u32 x = c[ b ];
        x += __shfl_xor( x, 1, 64 );
        x += __shfl_xor( x, 2, 64 );
        x += __shfl_xor( x, 4, 64 );
        c[ b ] = x;
    }
    if( blk < nblk && ( threadIdx.x & 7 ) == 0 )
    {
        hA[ blk ] = c[ 0 ];
        hC[ blk ] = c[ 1 ];
        hG[ blk ] = c[ 2 ];
        hT[ blk ] = c[ 3 ];
    }
}

// interleave counters and words into the reference layout (fMIndex.cpp:204-264)
__global__ void k_bwt_assemble( const u32* words, const u64* cA, const u64* cC, const u64* cG, const u64* cT, u64 n,
                                u64 nblkFull /* blocks holding symbols */, u32* out )
{
    const u64 blk = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( blk > nblkFull )
        return;
    // block `blk` starts at word blk*16 as long as all previous blocks are full (128 symbols); the final
    // counter block follows the (possibly partial) last symbol block directly
    u64 at = blk * 16;
    if( blk == nblkFull && ( n & 127 ) != 0 )
        at = ( nblkFull - 1 ) * 16 + 8 + ( ( n & 127 ) + 15 ) / 16;
    u32* o = out + at;
    const u64 cnt[ 4 ] = { cA[ blk ], cC[ blk ], cG[ blk ], cT[ blk ] }; // exclusive prefix = counts before the block
    for( int b = 0; b < 4; b++ )
    {
        o[ 2 * b ] = (u32)cnt[ b ];
        o[ 2 * b + 1 ] = (u32)( cnt[ b ] >> 32 );
    }
    if( blk == nblkFull )
        return; // final counter block only
    const u64 symBegin = blk * 128;
    const u64 nw = ( ( n - symBegin < 128 ? n - symBegin : 128 ) + 15 ) / 16;
    for( u64 k = 0; k < nw; k++ )
        o[ 8 + k ] = words[ blk * 8 + k ];
}

__global__ void k_sa_samples( const i64* SA, u64 n, u64 nsa, i64* out, u32 shift = 5 )
{
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( j >= nsa )
        return;
    out[ j ] = j == 0 ? (i64)-1 : SA[ ( j << shift ) - 1 ];
}

struct MaxOp
{
    __device__ u64 operator( )( const u64& a, const u64& b ) const
    {
        return a > b ? a : b;
    }
};
struct IsTied
{
    __device__ bool operator( )( const uint8_t& t ) const
    {
        return t != 0;
    }
};

struct Builder
{
    DevBuf tmp;
    int ensure_tmp( size_t b )
    {
        return tmp.reserve( b + 256 );
    }
};

#define GRID( n ) dim3( (unsigned)( ( ( n ) + 255 ) / 256 ) ), dim3( 256 )

// sort (key,pos[,gid]) by (gid,key) and refine groups; appends the still-tied elements to (tPos,tGid)
static int refine( Builder& B, const TextView& T, DevBuf& dPos, DevBuf& dGid, u64 m, u64 depth, bool singleGroup,
                   i64* SA, DevBuf& outPos, DevBuf& outGid, u64& outCount, u64 outOffset )
{
    if( m == 0 )
    {
        outCount = 0;
        return 0;
    }
    DevBuf key, key2, pos2, gid2, idx, idx2, rh, sh, tied, cnt;
    if( key.reserve( m * 8 ) || key2.reserve( m * 8 ) || pos2.reserve( m * 8 ) || rh.reserve( m * 8 ) ||
        sh.reserve( m * 8 ) || tied.reserve( m ) || cnt.reserve( 16 ) )
        return 1;
    hipLaunchKernelGGL( k_keys, GRID( m ), 0, 0, T, dPos.as<u64>( ), m, depth, key.as<u64>( ) );
    size_t tb = 0;
    // pass 1: by key (63 bits)
    MA_HIP( rocprim::radix_sort_pairs( nullptr, tb, key.as<u64>( ), key2.as<u64>( ), dPos.as<u64>( ), pos2.as<u64>( ), m,
                                       0, 63 ) );
    if( B.ensure_tmp( tb ) )
        return 1;
    MA_HIP( rocprim::radix_sort_pairs( B.tmp.p, tb, key.as<u64>( ), key2.as<u64>( ), dPos.as<u64>( ), pos2.as<u64>( ), m,
                                       0, 63 ) );
    u64 *sKey = key2.as<u64>( ), *sPos = pos2.as<u64>( ), *sGid = dGid.as<u64>( );
    if( !singleGroup )
    {
        // the first pass permuted pos/key; gid must follow: sort (key,gid) with the same keys -> same stable
        // permutation, then pass 2: stable sort of everything by gid
        if( gid2.reserve( m * 8 ) || idx.reserve( m * 8 ) || idx2.reserve( m * 8 ) )
            return 1;
        MA_HIP( rocprim::radix_sort_pairs( B.tmp.p, tb, key.as<u64>( ), key2.as<u64>( ), dGid.as<u64>( ),
                                           gid2.as<u64>( ), m, 0, 63 ) );
        // pass 2 carries pos and key via two stable sorts with identical keys (gid)
        size_t tb2 = 0;
        MA_HIP( rocprim::radix_sort_pairs( nullptr, tb2, gid2.as<u64>( ), idx.as<u64>( ), pos2.as<u64>( ),
                                           dPos.as<u64>( ), m, 0, 40 ) );
        if( B.ensure_tmp( tb2 ) )
            return 1;
        MA_HIP( rocprim::radix_sort_pairs( B.tmp.p, tb2, gid2.as<u64>( ), idx.as<u64>( ), pos2.as<u64>( ),
                                           dPos.as<u64>( ), m, 0, 40 ) );
        MA_HIP( rocprim::radix_sort_pairs( B.tmp.p, tb2, gid2.as<u64>( ), idx2.as<u64>( ), key2.as<u64>( ),
                                           key.as<u64>( ), m, 0, 40 ) );
        sGid = idx.as<u64>( );
        sPos = dPos.as<u64>( );
        sKey = key.as<u64>( );
    }
    hipLaunchKernelGGL( k_heads, GRID( m ), 0, 0, sGid, sKey, m, rh.as<u64>( ), sh.as<u64>( ) );
    size_t tb3 = 0;
    MA_HIP( rocprim::inclusive_scan( nullptr, tb3, rh.as<u64>( ), rh.as<u64>( ), m, MaxOp( ) ) );
    if( B.ensure_tmp( tb3 ) )
        return 1;
    MA_HIP( rocprim::inclusive_scan( B.tmp.p, tb3, rh.as<u64>( ), rh.as<u64>( ), m, MaxOp( ) ) );
    MA_HIP( rocprim::inclusive_scan( B.tmp.p, tb3, sh.as<u64>( ), sh.as<u64>( ), m, MaxOp( ) ) );
    DevBuf newGid;
    if( newGid.reserve( m * 8 ) )
        return 1;
    hipLaunchKernelGGL( k_place, GRID( m ), 0, 0, sGid, sPos, rh.as<u64>( ), sh.as<u64>( ), m, SA, newGid.as<u64>( ),
                        tied.as<uint8_t>( ) );
    // count + compact the tied elements
    size_t tb4 = 0;
    MA_HIP( rocprim::select( nullptr, tb4, sPos, tied.as<uint8_t>( ), (u64*)nullptr, cnt.as<u64>( ), m ) );
    if( B.ensure_tmp( tb4 ) )
        return 1;
    // worst case all tied: make room
    if( outPos.cap < ( outOffset + m ) * 8 )
    {
        DevBuf np, ng;
        const size_t want = std::max<size_t>( ( outOffset + m ) * 8, outPos.cap * 2 );
        if( np.reserve( want ) || ng.reserve( want ) )
            return 1;
        if( outOffset )
        {
            MA_HIP( hipMemcpy( np.p, outPos.p, outOffset * 8, hipMemcpyDeviceToDevice ) );
            MA_HIP( hipMemcpy( ng.p, outGid.p, outOffset * 8, hipMemcpyDeviceToDevice ) );
        }
        outPos = std::move( np );
        outGid = std::move( ng );
    }
    MA_HIP( rocprim::select( B.tmp.p, tb4, sPos, tied.as<uint8_t>( ), outPos.as<u64>( ) + outOffset, cnt.as<u64>( ), m ) );
    MA_HIP( rocprim::select( B.tmp.p, tb4, newGid.as<u64>( ), tied.as<uint8_t>( ), outGid.as<u64>( ) + outOffset,
                             cnt.as<u64>( ), m ) );
    u64 c = 0;
    MA_HIP( hipMemcpy( &c, cnt.p, 8, hipMemcpyDeviceToHost ) );
    outCount = c;
    for( DevBuf* d : { &key, &key2, &pos2, &gid2, &idx, &idx2, &rh, &sh, &tied, &cnt, &newGid } )
        d->release( );
    return 0;
}

static int build_impl( int32_t n_contigs, const uint64_t* contig_lens, const uint8_t* d_codes, ma_index** out )
{
    u64 F = 0;
    std::vector<u64> cs( n_contigs ), cl( n_contigs );
    for( int i = 0; i < n_contigs; i++ )
    {
        cs[ i ] = F;
        cl[ i ] = contig_lens[ i ];
        F += contig_lens[ i ];
    }
    if( F == 0 )
        return fail( "ma_index_build: empty genome" );
    const u64 n = 2 * F;
    std::unique_ptr<ma_index> x( new ma_index( ) ); // freed with its buffers on any early error return
    MA_HIP( hipGetDevice( &x->device ) );
    DevBuf hist;
    if( x->pac.reserve( ( F + 3 ) / 4 + 16 ) || hist.reserve( 64 ) )
        return 1;
    MA_HIP( hipMemset( hist.p, 0, 64 ) );
    MA_HIP( hipMemset( x->pac.p, 0, ( F + 3 ) / 4 + 16 ) );
    hipLaunchKernelGGL( k_pack, GRID( ( F + 3 ) / 4 ), 0, 0, d_codes, F, x->pac.as<uint8_t>( ),
                        hist.as<unsigned long long>( ) );
    unsigned long long h[ 8 ];
    MA_HIP( hipMemcpy( h, hist.p, 64, hipMemcpyDeviceToHost ) );
    // L2 of T.revcomp(T) (fMIndex.cpp:171-183)
    u64 L2[ 5 ];
    L2[ 0 ] = 0;
    for( int c = 0; c < 4; c++ )
        L2[ c + 1 ] = L2[ c ] + h[ c ] + h[ 3 - c ];
    TextView T{ x->pac.as<uint8_t>( ), F, n };
    DevBuf SA;
    if( SA.reserve( n * 8 ) )
        return 1;
    Builder B;
    // ---- round 0: bucketed sort
    u32 K = n > ( 400ull << 20 ) ? 2 : 0;
    if( const char* e = getenv( "MA_INDEX_BUCKET_K" ) ) // test hook: force the bucketed round 0 on small genomes
        K = (u32)atoi( e ) & 3;
    const u32 nb = 1u << ( 2 * K );
    DevBuf tPos, tGid, bPos, bGid, cnt;
    if( cnt.reserve( 16 ) || tPos.reserve( 1 << 20 ) || tGid.reserve( 1 << 20 ) )
        return 1;
    u64 nTied = 0, base = 0;
    std::vector<unsigned long long> bucketCount( nb, n );
    if( K > 0 )
    {
        DevBuf bh;
        if( bh.reserve( nb * 8 ) )
            return 1;
        MA_HIP( hipMemset( bh.p, 0, nb * 8 ) );
        hipLaunchKernelGGL( k_bucket_hist, dim3( 256 * 8 ), dim3( 256 ), 0, 0, T, K, bh.as<unsigned long long>( ) );
        MA_HIP( hipMemcpy( bucketCount.data( ), bh.p, nb * 8, hipMemcpyDeviceToHost ) );
        bh.release( );
    }
    for( u32 b = 0; b < nb; b++ )
    {
        u64 m = bucketCount[ b ];
        if( K == 0 )
        {
            if( bPos.reserve( n * 8 ) )
                return 1;
            hipLaunchKernelGGL( k_iota, GRID( n ), 0, 0, bPos.as<u64>( ), n );
        }
        else
        {
            rocprim::counting_iterator<u64> it( 0 );
            InBucket pred{ T, K, b };
            if( m == 0 )
                continue;
            if( bPos.reserve( ( m + 1 ) * 8 ) )
                return 1;
            size_t tb = 0;
            MA_HIP( rocprim::select( nullptr, tb, it, bPos.as<u64>( ), cnt.as<u64>( ), n, pred ) );
            if( B.ensure_tmp( tb ) )
                return 1;
            MA_HIP( rocprim::select( B.tmp.p, tb, it, bPos.as<u64>( ), cnt.as<u64>( ), n, pred ) );
            u64 got = 0;
            MA_HIP( hipMemcpy( &got, cnt.p, 8, hipMemcpyDeviceToHost ) );
            if( got != m )
                return fail( "ma_index_build: bucket histogram / selection mismatch" );
        }
        if( m == 0 )
            continue;
        if( bGid.reserve( m * 8 ) )
            return 1;
        hipLaunchKernelGGL( k_fill_u64, GRID( m ), 0, 0, bGid.as<u64>( ), m, base );
        u64 c = 0;
        if( refine( B, T, bPos, bGid, m, K, true, SA.as<i64>( ), tPos, tGid, c, nTied ) )
            return 1;
        nTied += c;
        base += m;
    }
    bPos.release( );
    bGid.release( );
    // ---- rounds r >= 1 over the tied suffixes only
    u64 depth = K + 29;
    int rounds = 0;
    while( nTied > 0 )
    {
        DevBuf nPos, nGid;
        if( nPos.reserve( 1 << 20 ) || nGid.reserve( 1 << 20 ) )
            return 1;
        u64 c = 0;
        if( refine( B, T, tPos, tGid, nTied, depth, false, SA.as<i64>( ), nPos, nGid, c, 0 ) )
            return 1;
        tPos = std::move( nPos );
        tGid = std::move( nGid );
        nTied = c;
        depth += 29;
        if( ++rounds > 100000 )
            return fail( "ma_index_build: suffix sort did not converge" );
    }
    tPos.release( );
    tGid.release( );
    // ---- BWT, occ blocks, SA samples
    DevBuf prim;
    if( prim.reserve( 16 ) )
        return 1;
    MA_HIP( hipMemset( prim.p, 0, 16 ) );
    hipLaunchKernelGGL( k_find_primary, GRID( n ), 0, 0, SA.as<i64>( ), n, prim.as<unsigned long long>( ) );
    unsigned long long primary = 0;
    MA_HIP( hipMemcpy( &primary, prim.p, 8, hipMemcpyDeviceToHost ) );
    const u64 nblk = ( n + 127 ) / 128; // blocks that hold symbols
    DevBuf words, hA, hC, hG, hT;
    if( words.reserve( nblk * 8 * 4 ) || hA.reserve( ( nblk + 1 ) * 8 ) || hC.reserve( ( nblk + 1 ) * 8 ) ||
        hG.reserve( ( nblk + 1 ) * 8 ) || hT.reserve( ( nblk + 1 ) * 8 ) )
        return 1;
    for( DevBuf* d : { &hA, &hC, &hG, &hT } )
        MA_HIP( hipMemset( d->p, 0, ( nblk + 1 ) * 8 ) );
    hipLaunchKernelGGL( k_bwt_blocks, GRID( nblk * 8 ), 0, 0, T, SA.as<i64>( ), (u64)primary, nblk, words.as<u32>( ),
                        hA.as<u64>( ), hC.as<u64>( ), hG.as<u64>( ), hT.as<u64>( ) );
    {
        size_t tb = 0;
        MA_HIP( rocprim::exclusive_scan( nullptr, tb, hA.as<u64>( ), hA.as<u64>( ), (u64)0, nblk + 1,
                                         rocprim::plus<u64>( ) ) );
        if( B.ensure_tmp( tb ) )
            return 1;
        for( DevBuf* d : { &hA, &hC, &hG, &hT } )
            MA_HIP( rocprim::exclusive_scan( B.tmp.p, tb, d->as<u64>( ), d->as<u64>( ), (u64)0, nblk + 1,
                                             rocprim::plus<u64>( ) ) );
    }
    const u64 nOcc = nblk + 1;
    x->n_words = ( n + 15 ) / 16 + nOcc * 8;
    x->n_sa = ( n + 32 ) / 32;
    if( x->bwt.reserve( x->n_words * 4 + 128 ) || x->sa.reserve( x->n_sa * 8 ) || x->cstart.reserve( n_contigs * 8 ) ||
        x->clen.reserve( n_contigs * 8 ) )
        return 1;
    MA_HIP( hipMemset( x->bwt.p, 0, x->n_words * 4 + 128 ) );
    hipLaunchKernelGGL( k_bwt_assemble, GRID( nblk + 1 ), 0, 0, words.as<u32>( ), hA.as<u64>( ), hC.as<u64>( ),
                        hG.as<u64>( ), hT.as<u64>( ), n, nblk, x->bwt.as<u32>( ) );
    hipLaunchKernelGGL( k_sa_samples, GRID( x->n_sa ), 0, 0, SA.as<i64>( ), n, x->n_sa, x->sa.as<i64>( ), 5u );
    const u32 dShift = ma::sa_dense_shift( ); // the hot path's denser sample (ma_common.h), straight from the full suffix array
    const u64 nDense = dShift ? ( n + ( 1ull << dShift ) ) >> dShift : 0;
    const bool dense = dShift != 0;
    if( dense )
    {
        if( x->saDense.reserve( nDense * 8 ) )
            return 1;
        hipLaunchKernelGGL( k_sa_samples, GRID( nDense ), 0, 0, SA.as<i64>( ), n, nDense, x->saDense.as<i64>( ), dShift );
    }
    MA_HIP( hipMemcpy( x->cstart.p, cs.data( ), n_contigs * 8, hipMemcpyHostToDevice ) );
    MA_HIP( hipMemcpy( x->clen.p, cl.data( ), n_contigs * 8, hipMemcpyHostToDevice ) );
    MA_HIP( hipDeviceSynchronize( ) );
    for( DevBuf* d : { &SA, &words, &hA, &hC, &hG, &hT, &prim, &hist, &cnt, &B.tmp } )
        d->release( );
    x->h_cstart = cs;
    x->h_clen = cl;
    x->v.bwt = x->bwt.as<u32>( );
    x->v.sa = x->sa.as<i64>( );
    if( dense )
    {
        x->v.sa_dense = x->saDense.as<i64>( );
        x->v.sa_shift = dShift;
    }
    x->v.pac = x->pac.as<uint8_t>( );
    x->v.cstart = x->cstart.as<u64>( );
    x->v.clen = x->clen.as<u64>( );
    x->v.n = n;
    x->v.F = F;
    x->v.primary = (i64)primary;
    for( int i = 0; i < 5; i++ )
        x->v.L2[ i ] = L2[ i ];
    x->v.n_contigs = n_contigs;
    if( ma::index_kmer_table( x.get( ) ) )
        return 1;
    *out = x.release( );
    return 0;
}
} // namespace

extern "C" int ma_index_build_device( int32_t n_contigs, const uint64_t* contig_lens, const void* d_codes, ma_index** out )
{
    if( !contig_lens || !d_codes || !out || n_contigs <= 0 )
        return fail( "ma_index_build_device: null argument" );
    return build_impl( n_contigs, contig_lens, (const uint8_t*)d_codes, out );
}

extern "C" int ma_index_build( int32_t n_contigs, const uint64_t* contig_lens, const uint8_t* codes, ma_index** out )
{
    if( !contig_lens || !codes || !out || n_contigs <= 0 )
        return fail( "ma_index_build: null argument" );
    u64 F = 0;
    for( int i = 0; i < n_contigs; i++ )
        F += contig_lens[ i ];
    DevBuf d;
    if( d.reserve( F + 16 ) )
        return 1;
    MA_HIP( hipMemcpy( d.p, codes, F, hipMemcpyHostToDevice ) );
    const int rc = build_impl( n_contigs, contig_lens, d.as<uint8_t>( ), out );
    d.release( );
    return rc;
}
