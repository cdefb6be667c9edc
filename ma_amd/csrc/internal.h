// internal.h -- host-side internals of libma_amd.so (not part of the ABI)
#pragma once
#include "ma_common.h"
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

namespace ma
{
void set_error( const std::string& s );
int fail( const std::string& s );
int band_stats_of_prims( unsigned long long out[ 8 ] ); // prims.hip: its copy of ksw_band.h's statistics
int band_long_stats_of_prims( unsigned long long out[ 8 ] ); // ... of the long jobs on the band of 120
int dp_family_stats_of_prims( unsigned long long out[ 16 ] ); // prims.hip: its copy of ksw_launch.h's per-family cell counters
u32 sa_dense_shift( ); // index.hip: log2 of the dense SA sample's interval (0: none)
}
struct ma_index;
namespace ma
{
int index_kmer_table( ma_index* x ); // index.hip: builds IndexView::kmer_tab (x->v otherwise complete)

#define MA_HIP( call )                                                                                                 \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t _e = ( call );                                                                                      \
        if( _e != hipSuccess )                                                                                         \
            return ma::fail( std::string( #call ) + ": " + hipGetErrorString( _e ) );                                  \
    } while( 0 )

// grow-only device buffer; owns its allocation (freed on destruction, so an early error return leaks nothing)
struct DevBuf
{
    void* p = nullptr;
    size_t cap = 0;
    DevBuf( ) = default;
    DevBuf( const DevBuf& ) = delete;
    DevBuf& operator=( const DevBuf& ) = delete;
    DevBuf( DevBuf&& o ) noexcept : p( o.p ), cap( o.cap )
    {
        o.p = nullptr;
        o.cap = 0;
    }
    DevBuf& operator=( DevBuf&& o ) noexcept
    {
        if( this != &o )
        {
            release( );
            p = o.p;
            cap = o.cap;
            o.p = nullptr;
            o.cap = 0;
        }
        return *this;
    }
    ~DevBuf( )
    {
        release( );
    }
    int reserve( size_t bytes );
    void release( );
    template <typename T> T* as( ) const
    {
        return reinterpret_cast<T*>( p );
    }
};

// HIP's current device is a per-host-thread setting that defaults to 0: every entry point that takes an index or
// a batch binds the calling thread to the device that object lives on (and restores the previous one on return), so a
// worker thread spawned by the host graph never allocates or launches on the wrong GPU.
struct DeviceGuard
{
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard( int dev )
    {
        err = hipGetDevice( &prev );
        if( err == hipSuccess && prev != dev )
        {
            err = hipSetDevice( dev );
            switched = err == hipSuccess;
        }
    }
    DeviceGuard( const DeviceGuard& ) = delete;
    DeviceGuard& operator=( const DeviceGuard& ) = delete;
    ~DeviceGuard( )
    {
        if( switched )
            (void)hipSetDevice( prev );
    }
};
#define MA_BIND_DEVICE( dev )                                                                                          \
    ma::DeviceGuard _ma_guard( dev );                                                                                  \
    if( _ma_guard.err != hipSuccess )                                                                                  \
    return ma::fail( std::string( "hipSetDevice(" ) + std::to_string( dev ) + "): " + hipGetErrorString( _ma_guard.err ) )
} // namespace ma

struct ma_index
{
    ma::IndexView v; // device pointers
    ma::DevBuf bwt, sa, saDense, kmerTab, pac, cstart, clen;
    uint64_t n_words = 0, n_sa = 0;
    std::vector<uint64_t> h_cstart, h_clen;
    int device = 0;
};
