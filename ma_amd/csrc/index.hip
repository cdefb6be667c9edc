// index.hip -- index handle: upload / download of the FMD-index + pack (FMIndex::vLoadFMIndex
// fMIndex.h:555-663, Pack::vLoadCollection pack.h:271-470 replaced by ma_index_create), runtime helpers.
#include "internal.h"
#include "fm_device.h"
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <sched.h>
#include <unistd.h>

namespace ma
{
static thread_local std::string g_err;
void set_error( const std::string& s )
{
    g_err = s;
}
int fail( const std::string& s )
{
    g_err = s;
    return 1;
}
int DevBuf::reserve( size_t bytes )
{
    if( bytes <= cap && p )
        return 0;
    // leave headroom: batch sizes of successive steps differ by a few per cent and a multi-GB hipFree + hipMalloc in
    // the middle of a step costs seconds (an eighth, at most 4 GiB; index arrays are allocated once and pay it too)
    if( bytes > ( 1u << 20 ) )
        bytes += std::min<size_t>( bytes / 8, (size_t)4 << 30 );
    if( p )
    {
        (void)hipFree( p );
        p = nullptr;
        cap = 0;
    }
    if( bytes == 0 )
        bytes = 16;
    hipError_t e = hipMalloc( &p, bytes );
    if( e != hipSuccess )
    {
        p = nullptr;
        return fail( std::string( "hipMalloc(" ) + std::to_string( bytes ) + "): " + hipGetErrorString( e ) );
    }
    cap = bytes;
    return 0;
}
void DevBuf::release( )
{
    if( p )
        (void)hipFree( p );
    p = nullptr;
    cap = 0;
}
} // namespace ma

using namespace ma;

extern "C" {

const char* ma_last_error( void )
{
    return g_err.c_str( );
}
int ma_abi_version( void )
{
    return MA_AMD_ABI_VERSION;
}
int ma_device_count( int* n )
{
    MA_HIP( hipGetDeviceCount( n ) );
    return 0;
}
int ma_set_device( int device )
{
    MA_HIP( hipSetDevice( device ) );
    return 0;
}

int ma_host_alloc( uint64_t bytes, void** out )
{
    if( !out )
        return fail( "ma_host_alloc: null argument" );
    *out = nullptr;
    MA_HIP( hipHostMalloc( out, bytes ? bytes : 1, hipHostMallocDefault ) );
    return 0;
}
int ma_host_free( void* p )
{
    if( p )
        MA_HIP( hipHostFree( p ) );
    return 0;
}

// The CPUs next to a GPU: /sys/bus/pci/devices/<domain:bus:device.function>/local_cpulist, e.g. "64-127,192-255".
int ma_host_bind_thread( int device, int mode, int* n_cpus )
{
    if( n_cpus )
        *n_cpus = 0;
    if( mode < -1 || mode > 1 )
        return fail( "ma_host_bind_thread: mode must be -1, 0 or 1" );
    const long nConf = sysconf( _SC_NPROCESSORS_CONF );
    if( nConf <= 0 || nConf > CPU_SETSIZE )
        return 0;
    // The mask the thread had when it first came here is the caller's (taskset, numactl, a SLURM task's affinity): every mode
    // stays inside it, and mode -1 gives exactly that mask back (the kernel itself intersects with the cgroup's cpuset only).
    static thread_local cpu_set_t tOrig;
    static thread_local bool tHaveOrig = false;
    if( !tHaveOrig )
    {
        CPU_ZERO( &tOrig );
        if( sched_getaffinity( 0, sizeof( tOrig ), &tOrig ) != 0 )
            return 0;
        tHaveOrig = true;
    }
    cpu_set_t local, want;
    CPU_ZERO( &local );
    CPU_ZERO( &want );
    if( mode >= 0 )
    {
        char bdf[ 64 ] = { 0 };
        MA_HIP( hipDeviceGetPCIBusId( bdf, (int)sizeof( bdf ) - 1, device ) );
        for( char* c = bdf; *c; ++c )
            *c = (char)tolower( (unsigned char)*c );
        char path[ 160 ];
        snprintf( path, sizeof( path ), "/sys/bus/pci/devices/%s/local_cpulist", bdf );
        FILE* f = fopen( path, "r" );
        if( !f )
            return 0; // no topology information: leave the thread where it is
        char line[ 1024 ] = { 0 };
        const bool got = fgets( line, (int)sizeof( line ) - 1, f ) != nullptr;
        fclose( f );
        if( !got )
            return 0;
        for( const char* c = line; *c; )
        {
            if( *c < '0' || *c > '9' )
            {
                ++c;
                continue;
            }
            char* e = nullptr;
            long a = strtol( c, &e, 10 ), z = a;
            if( *e == '-' )
                z = strtol( e + 1, &e, 10 );
            for( long i = a; i <= z && i < nConf; i++ )
                CPU_SET( (int)i, &local );
            c = e;
        }
        if( CPU_COUNT( &local ) == 0 || CPU_COUNT( &local ) >= nConf )
            return 0; // one node: nothing to choose
    }
    for( long i = 0; i < nConf; i++ )
        if( CPU_ISSET( (int)i, &tOrig ) && ( mode < 0 || ( CPU_ISSET( (int)i, &local ) != 0 ) == ( mode == 0 ) ) )
            CPU_SET( (int)i, &want );
    if( CPU_COUNT( &want ) == 0 )
        return 0; // none of the wanted CPUs is in the caller's mask: the thread stays where it is (n_cpus = 0)
    if( sched_setaffinity( 0, sizeof( want ), &want ) != 0 )
        return 0;
    cpu_set_t have;
    CPU_ZERO( &have );
    if( n_cpus && sched_getaffinity( 0, sizeof( have ), &have ) == 0 )
        *n_cpus = CPU_COUNT( &have );
    return 0;
}

void ma_params_default( ma_params* p )
{
    memset( p, 0, sizeof( *p ) );
    p->seeding_technique = 0;
    p->min_seed_len = 16;
    p->min_ambiguity = 0;
    p->max_ambiguity = 100;
    p->min_seed_size_drop = 15;
    p->max_num_soc = 30;
    p->min_num_soc = 1;
    p->harm_score_min = 18;
    p->max_score_lookahead = 3;
    p->switch_qlen = 800;
    p->min_delta_dist = 16;
    p->max_gap_area = 20;
    p->padding = 1000;
    p->bandwidth_ext = 512;
    p->min_bandwidth_gap = 20;
    p->zdrop = 200;
    p->sv_penalty = 100;
    p->match = 2;
    p->mismatch = 4;
    p->gap = 4;
    p->extend = 2;
    p->gap2 = 24;
    p->extend2 = 1;
    p->disable_heuristics = 0;
    p->soc_width = 0;
    p->srand_seed = 1;
    p->genome_size_disable = 10000000;
    p->rel_min_seed_size_amount = 0.005;
    p->harm_score_min_rel = 0.002;
    p->soc_score_decrease_tol = 0.1;
    p->score_diff_tol = 0.0001;
    p->max_delta_dist = 0.1;
    p->min_alignment_score = 75;
    p->report_n_best = 0;
    p->max_supplementary = 1;
    p->max_overlap_supplementary = 0.1;
    p->search_inversions = 0;
    p->zdrop_inversion = 100;
    p->use_paired_reads = 0;
    p->libm_probe = 0;
    p->mean_paired_dist = 400;
    p->std_paired_dist = 150;
    p->paired_bonus = 1.25;
}
void ma_params_illumina( ma_params* p )
{
    ma_params_default( p );
    p->seeding_technique = 1;
    p->max_ambiguity = 500;
    p->min_num_soc = 10;
    p->max_num_soc = 20;
}
void ma_params_illuminapaired( ma_params* p )
{
    ma_params_illumina( p ); // parameter.h:1089-1094
    p->use_paired_reads = 1;
}
void ma_params_pacbio( ma_params* p )
{
    ma_params_default( p ); // parameter.h:1096-1098
    p->max_supplementary = 100;
    p->min_num_soc = 5;
}
void ma_params_nanopore( ma_params* p )
{
    ma_params_pacbio( p ); // parameter.h:1101-1104
    p->seeding_technique = 1;
}
int ma_params_preset( const char* key, ma_params* p )
{
    if( !key || !p )
        return ma::fail( "ma_params_preset: null argument" );
    std::string k;
    for( const char* c = key; *c; ++c )
        k += (char)tolower( (unsigned char)*c );
    if( k == "default" )
        ma_params_default( p );
    else if( k == "illumina" )
        ma_params_illumina( p );
    else if( k == "illuminapaired" )
        ma_params_illuminapaired( p );
    else if( k == "pacbio" )
        ma_params_pacbio( p );
    else if( k == "nanopore" )
        ma_params_nanopore( p );
    else if( k == "sv-illumina" || k == "sv-pacbio" )
        return ma::fail( "ma_params_preset: the presetting '" + k + "' is not implemented on the device ('Rectangular SoC' = false)" );
    else
        return ma::fail( "The presetting '" + std::string( key ) + "' can not be found." );
    return 0;
}

extern "C++" {
// Dense SA sample (every 2^shift-th row) out of the reference's every-32nd one: each dense row that is not in
// the sparse sample walks LF steps until it hits one (one-off, ~1 s for GRCh38).  MA_SA_DENSE=0 keeps the sparse sample.
__global__ void k_sa_densify( ma::IndexView X, u64 n_dense, i64* dense, u32 shift )
{
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( j >= n_dense )
        return;
    i64 k = (i64)( j << shift ), s = 0;
    while( k & 31 )
    {
        k = ma::inv_psi( X, k );
        ++s;
    }
    dense[ j ] = j == 0 ? (i64)-1 : s + X.sa[ k >> 5 ];
}
namespace ma
{
// log2 of the dense sample's interval: 3 (every 8th row) by default; MA_SA_DENSE=0 keeps the sparse sample only,
// MA_SA_DENSE=1..4 choose another interval (tuning hook: 2 = 12.4 GB for GRCh38)
u32 sa_dense_shift( )
{
    const char* e = getenv( "MA_SA_DENSE" );
    if( !e )
        return 3;
    const int v = atoi( e );
    return v <= 0 ? 0u : (u32)( v > 4 ? 4 : v );
}
// x->v must be complete (sparse sample, bwt, L2, primary) and x->v.sa_dense null
int index_densify( ma_index* x )
{
    const u32 shift = sa_dense_shift( );
    if( shift == 0 )
        return 0;
    const u64 nd = ( x->v.n + ( 1ull << shift ) ) >> shift;
    if( x->saDense.reserve( nd * 8 ) )
        return 1;
    hipLaunchKernelGGL( k_sa_densify, dim3( (unsigned)( ( nd + 255 ) / 256 ) ), dim3( 256 ), 0, 0, x->v, nd, x->saDense.as<i64>( ), shift );
    MA_HIP( hipGetLastError( ) );
    MA_HIP( hipDeviceSynchronize( ) );
    x->v.sa_dense = x->saDense.as<i64>( );
    x->v.sa_shift = shift;
    return 0;
}
} // namespace ma

// K-mer table of the index (ma_common.h): one thread per K-mer runs the extension chain the seeding kernels would
__global__ void k_kmer_table( ma::IndexView X, u32 K, u64 n_keys, u64* out )
{
    const u64 key = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if( key >= n_keys )
        return;
    i64 ik[ 3 ];
    ma::init_interval( X, (u32)( key >> ( 2 * ( K - 1 ) ) ) & 3u, ik );
    for( u32 j = 1; j < K && ik[ 2 ] > 0; j++ )
    {
        i64 ok[ 3 ];
        u32 nb;
        ma::extend_backward( X, ik, (u32)( key >> ( 2 * ( K - 1 - j ) ) ) & 3u, ok, nb );
        ik[ 0 ] = ok[ 0 ], ik[ 1 ] = ok[ 1 ], ik[ 2 ] = ok[ 2 ];
    }
    if( ik[ 2 ] <= 0 )
        ik[ 0 ] = ik[ 1 ] = ik[ 2 ] = 0;
    out[ 2 * key ] = (u64)ik[ 0 ] | ( ( (u64)ik[ 2 ] & 0x1fffffffull ) << 35 );
    out[ 2 * key + 1 ] = (u64)ik[ 1 ] | ( ( (u64)ik[ 2 ] >> 29 ) << 35 );
}
namespace ma
{
// MA_KMER_K: 0 = no table, default 12 (268 MB), at most 14 (4.3 GB)
int index_kmer_table( ma_index* x )
{
    int K = 12;
    if( const char* e = getenv( "MA_KMER_K" ) )
        K = atoi( e );
    if( K < 2 || x->v.n >= ( 1ull << 35 ) )
        return 0;
    K = K > 14 ? 14 : K;
    const u64 keys = 1ull << ( 2 * K );
    if( x->kmerTab.reserve( keys * 16 ) )
        return 1;
    hipLaunchKernelGGL( k_kmer_table, dim3( (unsigned)( ( keys + 255 ) / 256 ) ), dim3( 256 ), 0, 0, x->v, (u32)K, keys, x->kmerTab.as<u64>( ) );
    MA_HIP( hipGetLastError( ) );
    MA_HIP( hipDeviceSynchronize( ) );
    x->v.kmer_tab = x->kmerTab.as<u64>( );
    x->v.kmer_k = (u32)K;
    return 0;
}
} // namespace ma
} // extern "C++"

int ma_index_create( const uint32_t* bwt_words, uint64_t n_words, const int64_t* sa, uint64_t n_sa,
                     const uint64_t L2[ 5 ], int64_t primary, uint64_t ref_len, const uint8_t* pac, int32_t n_contigs,
                     const uint64_t* contig_starts, const uint64_t* contig_lens, ma_index** out )
{
    if( !bwt_words || !sa || !pac || !out || n_contigs <= 0 )
        return fail( "ma_index_create: null argument" );
    std::unique_ptr<ma_index> x( new ma_index( ) ); // freed with its buffers on any early error return
    MA_HIP( hipGetDevice( &x->device ) );
    x->n_words = n_words;
    x->n_sa = n_sa;
    const uint64_t F = ref_len / 2;
    if( x->bwt.reserve( n_words * 4 + 64 ) || x->sa.reserve( n_sa * 8 ) || x->pac.reserve( ( F + 3 ) / 4 + 16 ) ||
        x->cstart.reserve( n_contigs * 8 ) || x->clen.reserve( n_contigs * 8 ) )
        return 1;
    MA_HIP( hipMemcpy( x->bwt.p, bwt_words, n_words * 4, hipMemcpyHostToDevice ) );
    MA_HIP( hipMemcpy( x->sa.p, sa, n_sa * 8, hipMemcpyHostToDevice ) );
    MA_HIP( hipMemcpy( x->pac.p, pac, ( F + 3 ) / 4, hipMemcpyHostToDevice ) );
    MA_HIP( hipMemcpy( x->cstart.p, contig_starts, n_contigs * 8, hipMemcpyHostToDevice ) );
    MA_HIP( hipMemcpy( x->clen.p, contig_lens, n_contigs * 8, hipMemcpyHostToDevice ) );
    x->h_cstart.assign( contig_starts, contig_starts + n_contigs );
    x->h_clen.assign( contig_lens, contig_lens + n_contigs );
    x->v.bwt = x->bwt.as<u32>( );
    x->v.sa = x->sa.as<i64>( );
    x->v.pac = x->pac.as<uint8_t>( );
    x->v.cstart = x->cstart.as<u64>( );
    x->v.clen = x->clen.as<u64>( );
    x->v.n = ref_len;
    x->v.F = F;
    x->v.primary = primary;
    for( int i = 0; i < 5; i++ )
        x->v.L2[ i ] = L2[ i ];
    x->v.n_contigs = n_contigs;
    if( ma::index_densify( x.get( ) ) || ma::index_kmer_table( x.get( ) ) )
        return 1;
    *out = x.release( );
    return 0;
}

int ma_index_destroy( ma_index* x )
{
    if( !x )
        return 0;
    MA_BIND_DEVICE( x->device );
    x->bwt.release( );
    x->sa.release( );
    x->saDense.release( );
    x->kmerTab.release( );
    x->pac.release( );
    x->cstart.release( );
    x->clen.release( );
    delete x;
    return 0;
}

int ma_index_device( const ma_index* x, int* device )
{
    if( !x || !device )
        return fail( "ma_index_device: null argument" );
    *device = x->device;
    return 0;
}

int ma_index_sizes( const ma_index* x, uint64_t* n_words, uint64_t* n_sa, uint64_t* ref_len, int32_t* n_contigs )
{
    if( !x )
        return fail( "ma_index_sizes: null index" );
    if( n_words )
        *n_words = x->n_words;
    if( n_sa )
        *n_sa = x->n_sa;
    if( ref_len )
        *ref_len = x->v.n;
    if( n_contigs )
        *n_contigs = x->v.n_contigs;
    return 0;
}

int ma_index_download( const ma_index* x, uint32_t* bwt_words, int64_t* sa, uint64_t L2[ 5 ], int64_t* primary,
                       uint8_t* pac, uint64_t* contig_starts, uint64_t* contig_lens )
{
    if( !x )
        return fail( "ma_index_download: null index" );
    MA_BIND_DEVICE( x->device );
    if( bwt_words )
        MA_HIP( hipMemcpy( bwt_words, x->bwt.p, x->n_words * 4, hipMemcpyDeviceToHost ) );
    if( sa )
        MA_HIP( hipMemcpy( sa, x->sa.p, x->n_sa * 8, hipMemcpyDeviceToHost ) );
    if( L2 )
        for( int i = 0; i < 5; i++ )
            L2[ i ] = x->v.L2[ i ];
    if( primary )
        *primary = x->v.primary;
    if( pac )
        MA_HIP( hipMemcpy( pac, x->pac.p, ( x->v.F + 3 ) / 4, hipMemcpyDeviceToHost ) );
    if( contig_starts )
        memcpy( contig_starts, x->h_cstart.data( ), x->h_cstart.size( ) * 8 );
    if( contig_lens )
        memcpy( contig_lens, x->h_clen.data( ), x->h_clen.size( ) * 8 );
    return 0;
}

// Pack::vExtract / vExtractSubsection (pack.h:1147-1236, 1440-1450) for n ranges [begin[i], end[i]) of the doubled text
// (reverse strand = complement of the mirrored forward base); out receives the ranges back to back.
int ma_pack_extract( const ma_index* x, const uint64_t* begin, const uint64_t* end, uint64_t n, uint8_t* out )
{
    if( !x || ( n && ( !begin || !end || !out ) ) )
        return fail( "ma_pack_extract: null argument" );
    MA_BIND_DEVICE( x->device );
    std::vector<uint8_t> bytes;
    for( uint64_t i = 0; i < n; i++ )
    {
        const uint64_t b = begin[ i ], e = end[ i ];
        if( b > e )
            return fail( "(vExtractSubsection) Try to extract with begin greater than end." );
        if( e > x->v.n || b >= x->v.n )
            return fail( "(vExtractSubsection) range check failed" );
        if( b < e && ( b >= x->v.F ) != ( e - 1 >= x->v.F ) )
            return fail( "(vExtractSubsection) Try to extract bridging sequence. This is impossible." );
        if( b == e )
            continue;
        const bool bRev = b >= x->v.F;
        // forward-strand positions covered: [lo, hi]
        const uint64_t lo = bRev ? x->v.n - e : b, hi = bRev ? x->v.n - 1 - b : e - 1;
        bytes.resize( ( hi >> 2 ) - ( lo >> 2 ) + 1 );
        MA_HIP( hipMemcpy( bytes.data( ), x->pac.as<uint8_t>( ) + ( lo >> 2 ), bytes.size( ), hipMemcpyDeviceToHost ) );
        auto base = [ & ]( uint64_t p ) { return (uint8_t)( ( bytes[ ( p >> 2 ) - ( lo >> 2 ) ] >> ( ( ~p & 3 ) << 1 ) ) & 3 ); };
        if( !bRev )
            for( uint64_t p = b; p < e; p++ )
                *out++ = base( p );
        else
            for( uint64_t p = b; p < e; p++ )
                *out++ = (uint8_t)( 3 - base( x->v.n - 1 - p ) );
    }
    return 0;
}

} // extern "C"
