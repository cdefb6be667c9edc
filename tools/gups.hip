// Random 64-byte gather ceiling of the device: what an FM-index occ lookup stream can reach at best.
// hipcc --offload-arch=gfx950 -O3 tools/gups.hip -o tools/_prof/gups && tools/_prof/gups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_gather( const uint4* tab, uint64_t nblk, uint64_t iters, int dep, uint64_t* out )
{
    uint64_t x = ( blockIdx.x * (uint64_t)blockDim.x + threadIdx.x ) * 0x9E3779B97F4A7C15ull + 1;
    uint64_t acc = 0;
    for( uint64_t i = 0; i < iters; i++ )
    {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint64_t b = ( x + ( dep ? acc & 1 : 0 ) ) % nblk;
        const uint4 a = tab[ b * 4 ], c = tab[ b * 4 + 3 ]; // first and last 16 B of a 64-B block
        acc += a.x + c.w;
    }
    out[ blockIdx.x * (uint64_t)blockDim.x + threadIdx.x ] = acc;
}
int main( )
{
    const uint64_t bytes = 3200ull << 20, nblk = bytes / 64;
    uint4* tab; uint64_t* out;
    hipMalloc( &tab, bytes ); hipMemset( tab, 1, bytes );
    hipMalloc( &out, 8ull << 20 );
    for( int dep = 0; dep < 2; dep++ )
        for( int wavesPerCu : { 8, 16, 32 } )
        {
            const int blocks = 256 * wavesPerCu / 4;
            const uint64_t iters = 2000;
            hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
            hipLaunchKernelGGL( k_gather, dim3( blocks ), dim3( 256 ), 0, 0, tab, nblk, 100, dep, out );
            hipEventRecord( e0 );
            hipLaunchKernelGGL( k_gather, dim3( blocks ), dim3( 256 ), 0, 0, tab, nblk, iters, dep, out );
            hipEventRecord( e1 ); hipEventSynchronize( e1 );
            float ms; hipEventElapsedTime( &ms, e0, e1 );
            const double n = (double)blocks * 256 * iters;
            printf( "dependent=%d waves/CU=%d: %.2f G blocks/s, %.2f TB/s of 64-B blocks\n", dep, wavesPerCu, n / ms / 1e6, n * 64 / ms / 1e9 );
        }
    return 0;
}
