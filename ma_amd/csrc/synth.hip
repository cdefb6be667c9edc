// synth.hip -- deterministic counter-based generators for the BASELINE.json synthetic workloads:
// a GRCh38-like genome (i.i.d. ACGT with two planted repeat families) and reads sampled from the
// indexed genome with substitution / insertion / deletion errors.  Everything is a pure function of
// (seed, position) so CPU and GPU sides can regenerate the same data.
#include "internal.h"
#include "fm_device.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

using namespace ma;

namespace
{
__host__ __device__ __forceinline__ u64 mix64( u64 x )
{
    x += 0x9E3779B97F4A7C15ull;
    x = ( x ^ ( x >> 30 ) ) * 0xBF58476D1CE4E5B9ull;
    x = ( x ^ ( x >> 27 ) ) * 0x94D049BB133111EBull;
    return x ^ ( x >> 31 );
}
__host__ __device__ __forceinline__ u64 h2( u64 seed, u64 a )
{
    return mix64( mix64( seed ) ^ a );
}
__host__ __device__ __forceinline__ u64 h3( u64 seed, u64 a, u64 b )
{
    return mix64( h2( seed, a ) ^ ( b * 0xD6E8FEB86659FD93ull ) );
}

__global__ void k_genome( u64 seed, u64 total, int repeats, uint8_t* out )
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= total )
        return;
    u32 b = (u32)( h2( seed, i ) & 3 );
    if( repeats )
    {
        const u64 blk6 = i / 6000, blk3 = i / 300;
        if( h2( seed + 1, blk6 ) % 103 == 0 )
        { // ~5 k copies of a 6 kb element, 5 % divergence
            const u32 cb = (u32)( h2( seed + 2, i % 6000 ) & 3 );
            const u64 hm = h2( seed + 3, i );
            b = ( hm % 100 ) < 5 ? ( cb + 1 + (u32)( ( hm >> 8 ) % 3 ) ) & 3 : cb;
        }
        else if( h2( seed + 4, blk3 ) % 10 == 0 )
        { // ~1 M copies of a 300 nt element, 12 % divergence
            const u32 cb = (u32)( h2( seed + 5, i % 300 ) & 3 );
            const u64 hm = h2( seed + 6, i );
            b = ( hm % 100 ) < 12 ? ( cb + 1 + (u32)( ( hm >> 8 ) % 3 ) ) & 3 : cb;
        }
    }
    out[ i ] = (uint8_t)b;
}

struct ReadGen
{
    IndexView X;
    u64 seed;
    u32 L;
    u32 thrSub, thrIns, thrDel; // thresholds on a 2^24 scale
    __device__ void window( u64 g, u64& start, u32& len ) const
    {
        const u64 F = X.F;
        len = L < F ? L : (u32)F;
        start = h2( seed, g ) % ( F - len + 1 );
        // keep the read inside one contig
        const i64 c = seq_id_for_position( X, start );
        const u64 cb = X.cstart[ c ], ce = cb + X.clen[ c ];
        if( ce - cb < len )
        {
            start = cb;
            len = (u32)( ce - cb );
        }
        else if( start + len > ce )
            start = ce - len;
    }
    // emits the read (or only counts when out == nullptr); returns its length
    __device__ u32 emit( u64 g, uint8_t* out ) const
    {
        u64 start;
        u32 len;
        window( g, start, len );
        const bool rev = g & 1;
        u32 n = 0;
        for( u32 j = 0; j < len; j++ )
        {
            // walk the source in read orientation
            const u64 p = rev ? start + len - 1 - j : start + j;
            u32 b = fwd_base( X, p );
            if( rev )
                b = 3 - b;
            const u64 h = h3( seed + 7, g, j );
            const u32 r0 = (u32)( h & 0xffffff ), r1 = (u32)( ( h >> 24 ) & 0xffffff );
            if( r0 < thrDel )
                continue;
            if( r0 < thrDel + thrIns )
            {
                if( out )
                    out[ n ] = (uint8_t)( ( h >> 48 ) & 3 );
                n++;
            }
            if( r1 < thrSub )
                b = ( b + 1 + (u32)( ( h >> 50 ) % 3 ) ) & 3;
            if( out )
                out[ n ] = (uint8_t)b;
            n++;
        }
        return n;
    }
};

__global__ void k_read_lens( ReadGen G, u64 first, u64 n, u64* lens )
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( t < n )
        lens[ t ] = G.emit( first + t, nullptr );
}
__global__ void k_read_fill( ReadGen G, u64 first, u64 n, const u64* off, uint8_t* codes, u64 cap )
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( t < n && off[ t + 1 ] <= cap )
        G.emit( first + t, codes + off[ t ] );
}
} // namespace

extern "C" int ma_synth_genome_device( uint64_t seed, uint64_t total_len, int32_t with_repeats, void* d_codes )
{
    if( !d_codes )
        return fail( "ma_synth_genome_device: null buffer" );
    if( total_len == 0 )
        return 0;
    hipLaunchKernelGGL( k_genome, dim3( (unsigned)( ( total_len + 255 ) / 256 ) ), dim3( 256 ), 0, 0, seed, total_len,
                        with_repeats, (uint8_t*)d_codes );
    MA_HIP( hipGetLastError( ) );
    MA_HIP( hipDeviceSynchronize( ) );
    return 0;
}

extern "C" int ma_synth_reads_device( const ma_index* x, uint64_t seed, uint64_t n_reads, uint32_t read_len,
                                      double sub_rate, double ins_rate, double del_rate, uint64_t first_read_index,
                                      void* d_codes, void* d_offsets, uint64_t codes_cap, uint64_t* n_bases )
{
    if( !x || !d_codes || !d_offsets )
        return fail( "ma_synth_reads_device: null argument" );
    MA_BIND_DEVICE( x->device );
    ReadGen G;
    G.X = x->v;
    G.seed = seed;
    G.L = read_len;
    G.thrSub = (u32)( sub_rate * 16777216.0 );
    G.thrIns = (u32)( ins_rate * 16777216.0 );
    G.thrDel = (u32)( del_rate * 16777216.0 );
    if( n_reads == 0 )
    {
        if( n_bases )
            *n_bases = 0;
        return 0;
    }
    DevBuf lens, tmp;
    if( lens.reserve( ( n_reads + 1 ) * 8 ) )
        return 1;
    MA_HIP( hipMemset( lens.p, 0, ( n_reads + 1 ) * 8 ) );
    hipLaunchKernelGGL( k_read_lens, dim3( (unsigned)( ( n_reads + 255 ) / 256 ) ), dim3( 256 ), 0, 0, G, first_read_index,
                        n_reads, lens.as<u64>( ) );
    size_t tb = 0;
    MA_HIP( rocprim::exclusive_scan( nullptr, tb, lens.as<u64>( ), (u64*)d_offsets, (u64)0, n_reads + 1,
                                     rocprim::plus<u64>( ) ) );
    if( tmp.reserve( tb + 256 ) )
        return 1;
    MA_HIP( rocprim::exclusive_scan( tmp.p, tb, lens.as<u64>( ), (u64*)d_offsets, (u64)0, n_reads + 1,
                                     rocprim::plus<u64>( ) ) );
    u64 total = 0;
    MA_HIP( hipMemcpy( &total, (u64*)d_offsets + n_reads, 8, hipMemcpyDeviceToHost ) );
    if( n_bases )
        *n_bases = total;
    int rc = 0;
    if( total > codes_cap )
        rc = fail( "ma_synth_reads_device: codes buffer too small" );
    else
    {
        hipLaunchKernelGGL( k_read_fill, dim3( (unsigned)( ( n_reads + 255 ) / 256 ) ), dim3( 256 ), 0, 0, G,
                            first_read_index, n_reads, (const u64*)d_offsets, (uint8_t*)d_codes, codes_cap );
        MA_HIP( hipGetLastError( ) );
        MA_HIP( hipDeviceSynchronize( ) );
    }
    lens.release( );
    tmp.release( );
    return rc;
}
