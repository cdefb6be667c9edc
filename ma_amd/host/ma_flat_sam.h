// ma_flat_sam.h -- SAM text straight from the FLAT result of a device batch (ma_engine.h: BatchResult -- one header array +
// one ops array per batch, as ma_batch_get_mapq_alignments downloads them), without building an Alignment object per
// alignment.  Reference-free (C ABI records + plain views of the reads and of the contig table), so the mirror modules
// (ma_modules.h / ma_sam.h) and the binding on the reference's real types (ma_ref_binding.h) share it.
//
// The bytes are those of the reference's FileWriter::execute (libs/ma/src/module/fileWriter.cpp:11-158) with
// Alignment::cigarString / getSamFlag / getSamPosition / getQuerySequence (libs/ma/inc/ma/container/alignment.h:367-467,
// 576-623) and Pack's position arithmetic (pack.h:900-997,1063-1067) for the options a flat view can serve: soft / hard
// clipping, M or =/X cigars, the CG tag of over-long cigars, omitting secondary / supplementary records.  "Emulate NGMLR's
// tag output" needs reference bases and Alignment objects: callers fall back to the per-read writer for it.
// Pinned by tests/test_sam_writer.py against the SAM goldens the compiled reference wrote.
#pragma once
#include "ma_amd.h"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace ma_amd
{
namespace flat
{
struct SamFormat
{
    bool bNoSecondary = false, bNoSupplementary = false, bOutputMCigar = true, bCGTag = true, bSoftClip = false;
};
// contig table of the pack (forward strand): names, start offsets, lengths
struct Contigs
{
    std::vector<std::string> vNames;
    std::vector<uint64_t> vStarts, vLengths;
    uint64_t forwardSize( ) const
    {
        return vStarts.empty( ) ? 0 : vStarts.back( ) + vLengths.back( );
    }
    // Pack::uiSequenceIdForPosition (pack.h:933-990) of a position on the forward strand
    size_t idOfForward( uint64_t uiPos ) const
    {
        size_t lo = 0, hi = vStarts.size( );
        while( hi - lo > 1 )
        {
            const size_t mid = ( lo + hi ) / 2;
            if( uiPos >= vStarts[ mid ] )
                lo = mid;
            else
                hi = mid;
        }
        return lo;
    }
};
struct ReadView
{
    const char* sName = nullptr;
    size_t uiNameLen = 0;
    const uint8_t* pCodes = nullptr; // A0 C1 G2 T3, else N
    const uint8_t* pQuality = nullptr; // FASTQ quality characters or null
    size_t uiLength = 0;
};

// An append-only byte arena: one per formatting thread and batch, written to the stream with one call.
class Arena
{
    std::vector<char> v;
    size_t n = 0;

  public:
    void clear( )
    {
        n = 0;
    }
    size_t size( ) const
    {
        return n;
    }
    const char* data( ) const
    {
        return v.data( );
    }
    char* grow( size_t k ) // k more bytes, returns where they start
    {
        if( n + k > v.size( ) )
            v.resize( ( n + k ) * 2 + 4096 );
        char* p = v.data( ) + n;
        n += k;
        return p;
    }
    void put( char c )
    {
        *grow( 1 ) = c;
    }
    void put( const char* s, size_t k )
    {
        memcpy( grow( k ), s, k );
    }
    void lit( const char* s )
    {
        put( s, strlen( s ) );
    }
    void number( uint64_t x )
    {
        char a[ 24 ];
        int k = 0;
        do
        {
            a[ k++ ] = (char)( '0' + x % 10 );
            x /= 10;
        } while( x != 0 );
        char* p = grow( (size_t)k );
        while( k > 0 )
            *p++ = a[ --k ];
    }
    void numberSigned( int64_t x )
    {
        if( x < 0 )
        {
            put( '-' );
            number( (uint64_t)( -x ) );
        }
        else
            number( (uint64_t)x );
    }
};

namespace detail
{
inline const char* baseChars( )
{
    return "ACGTN";
}
inline void putBases( Arena& rOut, const ReadView& rQ, uint64_t uiFrom, uint64_t uiTo )
{
    if( uiTo > rQ.uiLength )
        uiTo = rQ.uiLength;
    if( uiFrom >= uiTo )
        return;
    char* p = rOut.grow( uiTo - uiFrom );
    for( uint64_t i = uiFrom; i < uiTo; i++ )
        *p++ = baseChars( )[ rQ.pCodes[ i ] < 4 ? rQ.pCodes[ i ] : 4 ];
}
inline void putBasesReverseComplement( Arena& rOut, const ReadView& rQ, uint64_t uiFrom, uint64_t uiTo )
{
    if( uiTo > rQ.uiLength )
        throw std::runtime_error( "Index out of range (compCharAt)" );
    if( uiFrom >= uiTo )
        return;
    char* p = rOut.grow( uiTo - uiFrom );
    for( uint64_t i = uiTo; i > uiFrom; i-- )
    {
        const uint8_t c = rQ.pCodes[ i - 1 ];
        *p++ = baseChars( )[ c < 4 ? 3 - c : 4 ];
    }
}
inline void putQuality( Arena& rOut, const ReadView& rQ, uint64_t uiFrom, uint64_t uiTo ) // nucSeq.h:697-709; never reversed
{
    if( rQ.pQuality == nullptr )
    {
        rOut.put( '*' );
        return;
    }
    if( uiTo > rQ.uiLength )
        uiTo = rQ.uiLength;
    if( uiFrom < uiTo )
        rOut.put( (const char*)rQ.pQuality + uiFrom, uiTo - uiFrom );
}
inline void putUnmapped( Arena& rOut, const ReadView& rQ, const char* sMapQ ) // fileWriter.cpp:126-140
{
    rOut.put( rQ.sName, rQ.uiNameLen );
    rOut.lit( "\t4\t*\t0\t" );
    rOut.lit( sMapQ );
    rOut.lit( "\t*\t*\t0\t0\t" );
    putBases( rOut, rQ, 0, rQ.uiLength );
    rOut.put( '\t' );
    putQuality( rOut, rQ, 0, rQ.uiLength );
    rOut.put( '\n' );
}
} // namespace detail

// The SAM records of ONE read: its alignments pAlns[0 .. uiAlns) (MappingQuality order) with their (type, length) pairs in
// pOps (pAlns[k].ops_off counts pairs).
inline void formatRead( Arena& rOut, const SamFormat& rF, const Contigs& rContigs, const ReadView& rQ, const ma_alignment* pAlns, size_t uiAlns,
                        const uint64_t* pOps )
{
    const size_t uiStart = rOut.size( );
    const uint64_t uiFwd = rContigs.forwardSize( );
    for( size_t k = 0; k < uiAlns; k++ )
    {
        const ma_alignment& rA = pAlns[ k ];
        const uint64_t* pPairs = pOps + 2 * rA.ops_off;
        uint64_t uiLength = 0;
        for( uint32_t j = 0; j < rA.n_ops; j++ )
            uiLength += pPairs[ 2 * j + 1 ];
        if( uiLength == 0 )
            continue;
        if( ( rF.bNoSecondary && rA.secondary ) || ( rF.bNoSupplementary && rA.supplementary ) )
            continue;
        const uint64_t uiBeginRef = (uint64_t)rA.begin_ref, uiEndRef = (uint64_t)rA.end_ref;
        const uint64_t uiBeginQ = (uint64_t)rA.begin_q, uiEndQ = (uint64_t)rA.end_q;
        const bool bRev = uiBeginRef >= uiFwd;
        const bool bLong = rF.bCGTag && rA.n_ops >= 0x10000;
        // QNAME FLAG RNAME POS MAPQ
        rOut.put( rQ.sName, rQ.uiNameLen );
        rOut.put( '\t' );
        rOut.number( ( bRev ? 0x10u : 0u ) | ( rA.secondary ? 0x100u : 0u ) | ( rA.supplementary ? 0x800u : 0u ) );
        rOut.put( '\t' );
        // contig of the begin (pack.h:1063-1067); position of the alignment's forward-strand start, 1-based, (sic) one further
        // for reverse-strand alignments (alignment.h:596-603)
        const uint64_t uiAbsBegin = bRev ? 2 * uiFwd - ( uiBeginRef + 1 ) : uiBeginRef;
        const std::string& rName = rContigs.vNames[ rContigs.idOfForward( uiAbsBegin ) ];
        rOut.put( rName.data( ), rName.size( ) );
        rOut.put( '\t' );
        const uint64_t uiAbs = uiEndRef >= uiFwd ? 2 * uiFwd - ( uiEndRef + 1 ) : uiBeginRef;
        rOut.number( uiAbs - rContigs.vStarts[ rContigs.idOfForward( uiAbs ) ] + ( bRev ? 1 : 0 ) + 1 );
        rOut.put( '\t' );
        if( std::isnan( rA.mapq ) )
            rOut.lit( "255" );
        else
            rOut.numberSigned( (int64_t) static_cast<int>( std::ceil( rA.mapq * 254 ) ) );
        rOut.put( '\t' );
        // CIGAR (alignment.h:367-467): clip, the sections in forward-strand direction, clip
        if( bLong )
        {
            rOut.number( uiEndQ - uiBeginQ );
            rOut.put( 'S' );
        }
        else
        {
            const uint64_t uiLeftOver = uiEndQ < rQ.uiLength ? rQ.uiLength - uiEndQ : 0;
            const uint64_t uiHead = bRev ? uiLeftOver : uiBeginQ, uiTail = bRev ? uiBeginQ : uiLeftOver;
            const char cClip = rF.bSoftClip ? 'S' : 'H';
            if( uiHead > 0 )
            {
                rOut.number( uiHead );
                rOut.put( cClip );
            }
            uint64_t uiRunM = 0;
            for( uint32_t j = 0; j < rA.n_ops; j++ )
            {
                const uint64_t* pPair = pPairs + 2 * ( bRev ? rA.n_ops - 1 - j : j );
                const uint64_t uiType = pPair[ 0 ], uiLen = pPair[ 1 ];
                if( uiType <= 2 ) // seed, match, missmatch
                {
                    if( rF.bOutputMCigar )
                        uiRunM += uiLen;
                    else
                    {
                        rOut.number( uiLen );
                        rOut.put( uiType == 2 ? 'X' : '=' );
                    }
                }
                else
                {
                    if( rF.bOutputMCigar && uiRunM > 0 )
                    {
                        rOut.number( uiRunM );
                        rOut.put( 'M' );
                        uiRunM = 0;
                    }
                    rOut.number( uiLen );
                    rOut.put( uiType == 3 ? 'I' : 'D' );
                }
            }
            if( rF.bOutputMCigar && uiRunM > 0 )
            {
                rOut.number( uiRunM );
                rOut.put( 'M' );
            }
            if( uiTail > 0 )
            {
                rOut.number( uiTail );
                rOut.put( cClip );
            }
        }
        rOut.lit( "\t*\t0\t0\t" );
        // SEQ: the whole read when soft clipping, else the aligned part; reverse-complemented on the reverse strand
        const uint64_t uiFrom = rF.bSoftClip ? 0 : uiBeginQ, uiTo = rF.bSoftClip ? rQ.uiLength : uiEndQ;
        if( !rF.bSoftClip && uiTo > rQ.uiLength && !bRev )
            throw std::runtime_error( "Query length is off by " + std::to_string( (int64_t)rQ.uiLength - (int64_t)uiTo ) + "." );
        if( bRev )
            detail::putBasesReverseComplement( rOut, rQ, uiFrom, uiTo );
        else
            detail::putBases( rOut, rQ, uiFrom, uiTo );
        rOut.put( '\t' );
        detail::putQuality( rOut, rQ, uiBeginQ, uiEndQ ); // (sic) the aligned part, not reversed (alignment.h:611-614)
        if( bLong ) // TagGenerator::computeTag (fileWriter.h:327-357): the real cigar as CG:B:I
        {
            rOut.lit( "\tCG:B:I" );
            for( uint32_t j = 0; j < rA.n_ops; j++ )
            {
                static const uint32_t aOp[ 5 ] = { 7, 7, 8, 1, 2 };
                rOut.put( ',' );
                rOut.number( (uint32_t)( pPairs[ 2 * j + 1 ] << 4 ) | aOp[ pPairs[ 2 * j ] < 5 ? pPairs[ 2 * j ] : 0 ] );
            }
        }
        rOut.put( '\n' );
    }
    if( uiAlns == 0 )
        detail::putUnmapped( rOut, rQ, "255" );
    else if( rOut.size( ) == uiStart )
        detail::putUnmapped( rOut, rQ, "0" );
}
} // namespace flat
} // namespace ma_amd
