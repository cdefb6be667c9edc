// Random 64-byte gather ceiling of the device: what an FM-index occ lookup stream can reach at best.
// hipcc --offload-arch=gfx950 -O3 tools/gups.hip -o tools/_prof/gups && tools/_prof/gups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int LOADS, bool DEP>
__global__ void k_gather( const uint4* tab, uint64_t nblk, uint64_t iters, uint64_t* out )
{
    uint64_t x = ( blockIdx.x * (uint64_t)blockDim.x + threadIdx.x ) * 0x9E3779B97F4A7C15ull + 1;
    uint64_t acc = 0;
    for( uint64_t i = 0; i < iters; i++ )
    {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint64_t b = ( x + ( DEP ? acc : 0 ) ) % nblk; // DEP: the next address needs the loaded data (LF mapping)
        uint4 v[ 4 ];
#pragma unroll
        for( int k = 0; k < LOADS; k++ )
            v[ k ] = tab[ b * 4 + ( LOADS == 4 ? k : 3 * k ) ];
#pragma unroll
        for( int k = 0; k < LOADS; k++ )
            acc += v[ k ].x + v[ k ].w;
    }
    out[ blockIdx.x * (uint64_t)blockDim.x + threadIdx.x ] = acc;
}
template <int LOADS, bool DEP> void run( const uint4* tab, uint64_t nblk, uint64_t* out )
{
    for( int wavesPerCu : { 2, 4, 8, 16, 32 } )
    {
        const int blocks = 256 * wavesPerCu / 4;
        const uint64_t iters = 1000;
        hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
        hipLaunchKernelGGL( ( k_gather<LOADS, DEP> ), dim3( blocks ), dim3( 256 ), 0, 0, tab, nblk, 50, out );
        hipEventRecord( e0 );
        hipLaunchKernelGGL( ( k_gather<LOADS, DEP> ), dim3( blocks ), dim3( 256 ), 0, 0, tab, nblk, iters, out );
        hipEventRecord( e1 ); hipEventSynchronize( e1 );
        float ms; hipEventElapsedTime( &ms, e0, e1 );
        const double n = (double)blocks * 256 * iters;
        printf( "loads/block=%d dependent=%d waves/CU=%2d: %6.2f G blocks/s (%.2f TB/s), %.2f us per dependent step\n", LOADS, (int)DEP,
                wavesPerCu, n / ms / 1e6, n * 64 / ms / 1e9, ms * 1e3 / iters );
    }
}
int main( )
{
    const uint64_t bytes = 3200ull << 20, nblk = bytes / 64;
    uint4* tab; uint64_t* out;
    if( hipMalloc( &tab, bytes ) != hipSuccess || hipMemset( tab, 1, bytes ) != hipSuccess || hipMalloc( &out, 8ull << 20 ) != hipSuccess )
        return 1;
    run<2, false>( tab, nblk, out );
    run<4, false>( tab, nblk, out );
    run<4, true>( tab, nblk, out );
    return 0;
}
