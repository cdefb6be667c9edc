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

#define MA_HIP( call )                                                                                                 \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t _e = ( call );                                                                                      \
        if( _e != hipSuccess )                                                                                         \
            return ma::fail( std::string( #call ) + ": " + hipGetErrorString( _e ) );                                  \
    } while( 0 )

// grow-only device buffer
struct DevBuf
{
    void* p = nullptr;
    size_t cap = 0;
    int reserve( size_t bytes );
    void release( );
    template <typename T> T* as( ) const
    {
        return reinterpret_cast<T*>( p );
    }
};
} // namespace ma

struct ma_index
{
    ma::IndexView v; // device pointers
    ma::DevBuf bwt, sa, pac, cstart, clen;
    uint64_t n_words = 0, n_sa = 0;
    std::vector<uint64_t> h_cstart, h_clen;
    int device = 0;
};
