// Issue rate of the VALU instructions the DP kernels are made of (ksw_ext.h / ksw_pk.h): wave64 instructions per
// SIMD cycle for each opcode on its own and for the mix of one DP diagonal.  The "peak" of the roofline's valu_issue
// block comes from this measurement, not from an assumed cycles-per-instruction.
// hipcc --offload-arch=gfx950 -O3 tools/valu_mix.hip -o tools/_prof/valu_mix && tools/_prof/valu_mix
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define R8( X ) X( 0 ) X( 1 ) X( 2 ) X( 3 ) X( 4 ) X( 5 ) X( 6 ) X( 7 )
// one asm statement = 8 instructions on 8 independent accumulators; the body repeats it 8 times = 64 instructions
#define OPS                                                                                                            \
    OP( pk_add_u16, "v_pk_add_u16 %0, %0, %1\n" )                                                                      \
    OP( pk_sub_u16, "v_pk_sub_u16 %0, %0, %1\n" )                                                                      \
    OP( pk_max_i16, "v_pk_max_i16 %0, %0, %1\n" )                                                                      \
    OP( pk_min_i16, "v_pk_min_i16 %0, %0, %1\n" )                                                                      \
    OP( pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2\n" )                                                                  \
    OP( pk_ashr_i16, "v_pk_ashrrev_i16 %0, 1, %0\n" )                                                                  \
    OP( bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca\n" )                                                          \
    OP( bfi, "v_bfi_b32 %0, %1, %0, %2\n" )                                                                            \
    OP( mov_dpp_wave_ror, "v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n" )                             \
    OP( mov_dpp_row_ror, "v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n" )                               \
    OP( max_dpp_row_ror, "v_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n" )                           \
    OP( perm, "v_perm_b32 %0, %0, %1, %2\n" )                                                                          \
    OP( alignbit, "v_alignbit_b32 %0, %0, %1, 16\n" )                                                                  \
    OP( and_b32, "v_and_b32 %0, %0, %1\n" )                                                                            \
    OP( and_or_b32, "v_and_or_b32 %0, %0, %1, %2\n" )                                                                  \
    OP( lshrrev_b32, "v_lshrrev_b32 %0, 1, %0\n" )                                                                     \
    OP( add_u32, "v_add_u32 %0, %0, %1\n" )                                                                            \
    OP( max_i32, "v_max_i32 %0, %0, %1\n" )                                                                            \
    OP( fma_f32, "v_fma_f32 %0, %0, %1, %2\n" )                                                                        \
    OP( add_u16_sdwa, "v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\n" ) \
    OP( mbcnt, "v_mbcnt_lo_u32_b32 %0, %0, %1\n" )

enum Kind
{
#define OP( name, txt ) K_##name,
    OPS
#undef OP
        K_dp_mix,
    K_COUNT
};
static const char* kNames[] = {
#define OP( name, txt ) #name,
    OPS
#undef OP
    "dp_mix(ksw_ext diagonal)" };

template <int KIND> __device__ __forceinline__ void body( uint32_t ( &a )[ 8 ], uint32_t c, uint32_t d, uint64_t& w )
{
#define E8( t ) t t t t t t t t
    switch( KIND )
    {
#define OP( name, txt )                                                                                                \
    case K_##name:                                                                                                     \
        _Pragma( "unroll" ) for( int k = 0; k < 8; k++ ) _Pragma( "unroll" ) for( int i = 0; i < 8; i++ )              \
            asm volatile( txt : "+v"( a[ i ] ) : "v"( c ), "v"( d ), "v"( w ) );                                      \
        break;
        OPS
#undef OP
    default:
        break;
    }
}

// The instruction mix of one diagonal of ksw_ext_core<1, LEFT, false> (two cells per lane): same opcodes in the same
// proportions, each on independent registers so that only the issue rate is measured: 64 instructions.
__device__ __forceinline__ void dp_mix( uint32_t ( &a )[ 8 ], uint32_t c, uint32_t d )
{
#pragma unroll
    for( int k = 0; k < 1; k++ )
        asm volatile(
            // 5 DPP moves + 5 alignbit (neighbour shift of X, V, X2, H, Q)
            "v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b32_dpp %1, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b32_dpp %2, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b32_dpp %3, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b32_dpp %4, %4 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
            "v_alignbit_b32 %5, %5, %0, 16\n"
            "v_alignbit_b32 %6, %6, %1, 16\n"
            "v_alignbit_b32 %7, %7, %2, 16\n"
            "v_alignbit_b32 %0, %0, %3, 16\n"
            "v_alignbit_b32 %1, %1, %4, 16\n"
            // live mask: sub, subsat, add, ashr
            "v_pk_sub_u16 %2, %2, %8\n"
            "v_pk_sub_u16 %3, %8, %2 clamp\n"
            "v_pk_add_u16 %4, %3, %9\n"
            "v_pk_ashrrev_i16 %4, 15, %4\n"
            // score: xor, min, lshl_or, perm
            "v_xor_b32 %5, %5, %6\n"
            "v_pk_min_u16 %5, %5, %8\n"
            "v_lshl_or_b32 %5, %5, 8, %9\n"
            "v_perm_b32 %5, %8, %9, %5\n"
            // a, b, a2, b2
            "v_pk_add_u16 %6, %6, %7\n"
            "v_pk_add_u16 %7, %7, %0\n"
            "v_pk_add_u16 %0, %0, %1\n"
            "v_pk_add_u16 %1, %1, %2\n"
            // max chain (4), d (and, sub), clip (min, and)
            "v_pk_max_i16 %5, %5, %6\n"
            "v_pk_max_i16 %2, %7, %0\n"
            "v_pk_max_i16 %2, %2, %1\n"
            "v_pk_max_i16 %5, %5, %2\n"
            "v_and_b32 %3, %5, %8\n"
            "v_pk_sub_u16 %3, %9, %3\n"
            "v_pk_min_i16 %5, %5, %8\n"
            "v_and_b32 %5, %5, %9\n"
            // nu, nv, tmp, a, b, tmp2, a2, b2 (8 subs)
            "v_pk_sub_u16 %2, %5, %6\n"
            "v_pk_sub_u16 %4, %5, %7\n"
            "v_pk_sub_u16 %3, %5, %8\n"
            "v_pk_sub_u16 %6, %6, %3\n"
            "v_pk_sub_u16 %7, %7, %3\n"
            "v_pk_sub_u16 %3, %5, %9\n"
            "v_pk_sub_u16 %0, %0, %3\n"
            "v_pk_sub_u16 %1, %1, %3\n"
            // nx, ny, nx2, ny2: 4 max + 4 sub
            "v_pk_max_i16 %2, %6, %8\n"
            "v_pk_sub_u16 %2, %2, %9\n"
            "v_pk_max_i16 %3, %7, %8\n"
            "v_pk_sub_u16 %3, %3, %9\n"
            "v_pk_max_i16 %4, %0, %8\n"
            "v_pk_sub_u16 %4, %4, %9\n"
            "v_pk_max_i16 %5, %1, %8\n"
            "v_pk_sub_u16 %5, %5, %9\n"
            // flags: 4 sub + 4 (shift-and-or)
            "v_pk_sub_u16 %6, %8, %6\n"
            "v_pk_sub_u16 %7, %8, %7\n"
            "v_pk_sub_u16 %0, %8, %0\n"
            "v_pk_sub_u16 %1, %8, %1\n"
            "v_lshrrev_b32 %6, 12, %6\n"
            "v_and_or_b32 %3, %6, %8, %3\n"
            "v_lshrrev_b32 %7, 11, %7\n"
            "v_and_or_b32 %3, %7, %8, %3\n"
            "v_lshrrev_b32 %0, 10, %0\n"
            "v_and_or_b32 %3, %0, %8, %3\n"
            "v_lshrrev_b32 %1, 9, %1\n"
            "v_and_or_b32 %3, %1, %8, %3\n"
            // commits: 3 bitop3 + H: ashr, add, bitop3 + max test: max, cmp
            "v_bitop3_b32 %2, %2, %8, %9 bitop3:0xca\n"
            "v_bitop3_b32 %4, %4, %8, %9 bitop3:0xca\n"
            "v_bitop3_b32 %5, %5, %8, %9 bitop3:0xca\n"
            "v_pk_ashrrev_i16 %6, 8, %2\n"
            "v_pk_add_u16 %6, %6, %7\n"
            "v_bitop3_b32 %6, %6, %8, %9 bitop3:0xca\n"
            : "+v"( a[ 0 ] ), "+v"( a[ 1 ] ), "+v"( a[ 2 ] ), "+v"( a[ 3 ] ), "+v"( a[ 4 ] ), "+v"( a[ 5 ] ), "+v"( a[ 6 ] ),
              "+v"( a[ 7 ] )
            : "v"( c ), "v"( d ) );
}

template <int KIND> __global__ void __launch_bounds__( 256 ) k_valu( uint32_t iters, uint32_t* out, unsigned long long* cyc )
{
    uint32_t a[ 8 ];
#pragma unroll
    for( int i = 0; i < 8; i++ )
        a[ i ] = threadIdx.x * 2654435761u + i;
    uint32_t c = 0x00030003u + threadIdx.x, d = 0x01010101u;
    uint64_t w = ( (uint64_t)c << 32 ) | d;
    const unsigned long long t0 = __builtin_readcyclecounter( );
    for( uint32_t it = 0; it < iters; it++ )
    {
        if( KIND == K_dp_mix )
            dp_mix( a, c, d );
        else
            body<KIND>( a, c, d, w );
    }
    const unsigned long long t1 = __builtin_readcyclecounter( );
    uint32_t s = 0;
#pragma unroll
    for( int i = 0; i < 8; i++ )
        s ^= a[ i ];
    out[ blockIdx.x * blockDim.x + threadIdx.x ] = s;
    if( ( threadIdx.x & 63 ) == 0 )
        atomicMax( cyc, t1 - t0 );
}

template <int KIND> static void run( uint32_t* out, unsigned long long* dcyc )
{
    const uint32_t iters = 20000;
    const int instrPerIter = 64;
    for( int wavesPerSimd : { 1, 2, 4, 8 } )
    {
        const int blocks = 256 * wavesPerSimd; // 256 threads = 4 waves = one per SIMD of a CU
        hipEvent_t e0, e1;
        (void)hipEventCreate( &e0 );
        (void)hipEventCreate( &e1 );
        hipLaunchKernelGGL( k_valu<KIND>, dim3( blocks ), dim3( 256 ), 0, 0, 200u, out, dcyc );
        (void)hipMemset( dcyc, 0, 8 );
        (void)hipEventRecord( e0 );
        hipLaunchKernelGGL( k_valu<KIND>, dim3( blocks ), dim3( 256 ), 0, 0, iters, out, dcyc );
        (void)hipEventRecord( e1 );
        (void)hipEventSynchronize( e1 );
        float ms = 0;
        (void)hipEventElapsedTime( &ms, e0, e1 );
        unsigned long long cyc = 0;
        (void)hipMemcpy( &cyc, dcyc, 8, hipMemcpyDeviceToHost );
        const double waveInstr = (double)blocks * 4 * iters * instrPerIter;
        const double perSimd = (double)wavesPerSimd * iters * instrPerIter; // instructions one SIMD issued
        // readcyclecounter = s_memtime: a constant 100 MHz-class counter on some parts; report both views
        printf( "%-26s waves/SIMD=%d: %8.1f G wave-instr/s  %6.3f ns per instr per SIMD  (s_memtime ticks/instr/SIMD %.3f)\n",
                kNames[ KIND ], wavesPerSimd, waveInstr / ms / 1e6, ms * 1e6 / perSimd, (double)cyc / perSimd );
        (void)hipEventDestroy( e0 );
        (void)hipEventDestroy( e1 );
    }
}

template <int K> struct RunAll
{
    static void go( uint32_t* out, unsigned long long* c )
    {
        RunAll<K - 1>::go( out, c );
        run<K>( out, c );
    }
};
template <> struct RunAll<-1>
{
    static void go( uint32_t*, unsigned long long* ) {}
};

int main( )
{
    uint32_t* out;
    unsigned long long* c;
    if( hipMalloc( &out, 256 * 8 * 256 * 4 ) != hipSuccess || hipMalloc( &c, 8 ) != hipSuccess )
        return 1;
    int clk = 0;
    (void)hipDeviceGetAttribute( &clk, hipDeviceAttributeClockRate, 0 );
    printf( "device clock attribute: %d kHz; 1024 SIMDs\n", clk );
    RunAll<K_COUNT - 1>::go( out, c );
    return 0;
}
