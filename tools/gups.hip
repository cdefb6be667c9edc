// Random 64-byte gather ceiling of the device: what an FM-index occ lookup stream can reach at best, and the
// calibration of rocprofv3's FETCH_SIZE for exactly this access pattern (MI355X_MICROARCH.md: the x2 correction is
// documented for wide coalesced streaming reads only).
//   tools/_prof/gups                      sweep: 32 / 64 / 128 bytes per random block, independent and dependent
//   tools/_prof/gups one <bytes> <dep> <wavesPerCu> <iters>    ONE launch of a known byte count (run it under
//                                          rocprofv3 --pmc FETCH_SIZE: counter / printed bytes = the correction)
//   tools/_prof/gups stream               ONE coalesced 16 B/lane pass over the table (the documented x2 case)
// hipcc --offload-arch=gfx950 -O3 tools/gups.hip -o tools/_prof/gups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
// BYTES per random block: 32 (2 x 16 B of a 64-B block), 64 (the occ block), 128 (a whole 128-B line)
template <int BYTES, bool DEP>
__global__ void k_gather( const uint4* tab, uint64_t nblk, uint64_t iters, uint64_t* out )
{
    uint64_t x = ( blockIdx.x * (uint64_t)blockDim.x + threadIdx.x ) * 0x9E3779B97F4A7C15ull + 1;
    uint64_t acc = 0;
    constexpr int LOADS = BYTES / 16, STRIDE = BYTES == 128 ? 8 : 4;
    for( uint64_t i = 0; i < iters; i++ )
    {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint64_t b = ( x + ( DEP ? acc : 0 ) ) % nblk; // DEP: the next address needs the loaded data (LF mapping)
        uint4 v[ LOADS ];
#pragma unroll
        for( int k = 0; k < LOADS; k++ )
            v[ k ] = tab[ b * STRIDE + ( BYTES == 32 ? 3 * k : k ) ];
#pragma unroll
        for( int k = 0; k < LOADS; k++ )
            acc += v[ k ].x + v[ k ].w;
    }
    out[ blockIdx.x * (uint64_t)blockDim.x + threadIdx.x ] = acc;
}
__global__ void k_stream( const uint4* tab, uint64_t n16, uint64_t* out )
{
    uint64_t acc = 0;
    for( uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x )
    {
        const uint4 v = tab[ i ];
        acc += v.x + v.w;
    }
    out[ blockIdx.x * (uint64_t)blockDim.x + threadIdx.x ] = acc;
}
template <int BYTES, bool DEP> float launch( const uint4* tab, uint64_t nblk, uint64_t* out, int wavesPerCu, uint64_t iters )
{
    const int blocks = 256 * wavesPerCu / 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate( &e0 ); (void)hipEventCreate( &e1 );
    (void)hipEventRecord( e0 );
    hipLaunchKernelGGL( ( k_gather<BYTES, DEP> ), dim3( blocks ), dim3( 256 ), 0, 0, tab, nblk, iters, out );
    (void)hipEventRecord( e1 ); (void)hipEventSynchronize( e1 );
    float ms = 0; (void)hipEventElapsedTime( &ms, e0, e1 );
    return ms;
}
template <int BYTES, bool DEP> void run( const uint4* tab, uint64_t nblk, uint64_t* out )
{
    for( int wavesPerCu : { 2, 4, 8, 16, 32 } )
    {
        const uint64_t iters = 1000;
        launch<BYTES, DEP>( tab, nblk, out, wavesPerCu, 50 );
        const float ms = launch<BYTES, DEP>( tab, nblk, out, wavesPerCu, iters );
        const double n = (double)( 256 * wavesPerCu / 4 ) * 256 * iters;
        printf( "bytes/block=%3d dependent=%d waves/CU=%2d: %6.2f G blocks/s (%.2f TB/s useful), %.2f us per dependent step\n", BYTES,
                (int)DEP, wavesPerCu, n / ms / 1e6, n * BYTES / ms / 1e9, ms * 1e3 / iters );
    }
}
int main( int argc, char** argv )
{
    const uint64_t bytes = 3200ull << 20;
    uint4* tab; uint64_t* out;
    if( hipMalloc( &tab, bytes ) != hipSuccess || hipMemset( tab, 1, bytes ) != hipSuccess || hipMalloc( &out, 8ull << 20 ) != hipSuccess )
        return 1;
    (void)hipDeviceSynchronize( );
    if( argc >= 2 && !strcmp( argv[ 1 ], "stream" ) )
    {
        hipLaunchKernelGGL( k_stream, dim3( 256 * 8 ), dim3( 256 ), 0, 0, tab, bytes / 16, out );
        (void)hipDeviceSynchronize( );
        printf( "k_stream: known_bytes=%llu\n", (unsigned long long)bytes );
        return 0;
    }
    if( argc >= 6 && !strcmp( argv[ 1 ], "one" ) )
    {
        const int B = atoi( argv[ 2 ] ), dep = atoi( argv[ 3 ] ), w = atoi( argv[ 4 ] );
        const uint64_t iters = strtoull( argv[ 5 ], nullptr, 10 );
        float ms = 0;
        if( B == 32 ) ms = dep ? launch<32, true>( tab, bytes / 64, out, w, iters ) : launch<32, false>( tab, bytes / 64, out, w, iters );
        if( B == 64 ) ms = dep ? launch<64, true>( tab, bytes / 64, out, w, iters ) : launch<64, false>( tab, bytes / 64, out, w, iters );
        if( B == 128 ) ms = dep ? launch<128, true>( tab, bytes / 128, out, w, iters ) : launch<128, false>( tab, bytes / 128, out, w, iters );
        const double n = (double)( 256 * w / 4 ) * 256 * iters;
        printf( "k_gather<%d,%d>: blocks=%.0f known_bytes=%.0f lines64=%.0f ms=%.3f\n", B, dep, n, n * B, n * ( B == 128 ? 2 : 1 ), ms );
        return 0;
    }
    run<32, false>( tab, bytes / 64, out );
    run<64, false>( tab, bytes / 64, out );
    run<64, true>( tab, bytes / 64, out );
    run<128, false>( tab, bytes / 128, out );
    run<128, true>( tab, bytes / 128, out );
    return 0;
}
