// TEST INFRASTRUCTURE ONLY -- binary case-file readers shared by ref_dump (real reference) and
// oracle_dump (our CPU restatement).  Writers live in tests/ma_testlib.py.
//
// MACASE01: magic[8] | u32 n_contigs | { u32 name_len, name, u64 len, codes[len] (0..3, >=4 = N) }*
//           | u32 n_reads | { u32 len, codes[len] (0..4) }*
// KSWCAS01: magic[8] | u32 n | { i32 qlen, i32 tlen, i32 w, i32 zdrop, i32 flag, q[qlen], t[tlen] }*
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

struct CaseFile
{
    std::vector<std::string> names;
    std::vector<std::vector<uint8_t>> contigs;
    std::vector<std::vector<uint8_t>> reads;
};

struct KswCase
{
    int w, zdrop, flag;
    std::vector<uint8_t> q, t;
};

namespace dumpfmt
{
inline void rd( FILE* f, void* p, size_t n )
{
    if( n && fread( p, 1, n, f ) != n )
        throw std::runtime_error( "short read in case file" );
}
template <typename T> inline T rdv( FILE* f )
{
    T x;
    rd( f, &x, sizeof( T ) );
    return x;
}
} // namespace dumpfmt

inline CaseFile readCase( const char* sPath )
{
    FILE* f = fopen( sPath, "rb" );
    if( !f )
        throw std::runtime_error( std::string( "cannot open " ) + sPath );
    char magic[ 8 ];
    dumpfmt::rd( f, magic, 8 );
    if( memcmp( magic, "MACASE01", 8 ) )
        throw std::runtime_error( "bad case magic" );
    CaseFile c;
    uint32_t nC = dumpfmt::rdv<uint32_t>( f );
    for( uint32_t i = 0; i < nC; i++ )
    {
        uint32_t nl = dumpfmt::rdv<uint32_t>( f );
        std::string s( nl, ' ' );
        dumpfmt::rd( f, &s[ 0 ], nl );
        uint64_t len = dumpfmt::rdv<uint64_t>( f );
        std::vector<uint8_t> v( len );
        dumpfmt::rd( f, v.data( ), len );
        c.names.push_back( s );
        c.contigs.push_back( std::move( v ) );
    }
    uint32_t nR = dumpfmt::rdv<uint32_t>( f );
    for( uint32_t i = 0; i < nR; i++ )
    {
        uint32_t len = dumpfmt::rdv<uint32_t>( f );
        std::vector<uint8_t> v( len );
        dumpfmt::rd( f, v.data( ), len );
        c.reads.push_back( std::move( v ) );
    }
    fclose( f );
    return c;
}

inline std::vector<KswCase> readKswCases( const char* sPath )
{
    FILE* f = fopen( sPath, "rb" );
    if( !f )
        throw std::runtime_error( std::string( "cannot open " ) + sPath );
    char magic[ 8 ];
    dumpfmt::rd( f, magic, 8 );
    if( memcmp( magic, "KSWCAS01", 8 ) )
        throw std::runtime_error( "bad ksw case magic" );
    uint32_t n = dumpfmt::rdv<uint32_t>( f );
    std::vector<KswCase> v( n );
    for( auto& k : v )
    {
        int32_t ql = dumpfmt::rdv<int32_t>( f ), tl = dumpfmt::rdv<int32_t>( f );
        k.w = dumpfmt::rdv<int32_t>( f );
        k.zdrop = dumpfmt::rdv<int32_t>( f );
        k.flag = dumpfmt::rdv<int32_t>( f );
        k.q.resize( ql );
        k.t.resize( tl );
        dumpfmt::rd( f, k.q.data( ), ql );
        dumpfmt::rd( f, k.t.data( ), tl );
    }
    fclose( f );
    return v;
}
