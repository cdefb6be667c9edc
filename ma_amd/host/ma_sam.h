// ma_sam.h -- SAM emission and FASTA/FASTQ reading of the drop-in host layer: FileWriter /
// FileReader with the reference's name,
// Module signature, constructors, options and output bytes (libs/ma/inc/ma/module/fileWriter.h:21-78,364-440,
// libs/ma/src/module/fileWriter.cpp:11-158), plus the Alignment / Pack / NucSeq string helpers it calls
// (alignment.h:367-467,576-623; pack.h:900-997,1063-1067; nucSeq.h:558-713).  Pure host code: SAM text is
// formatting of what the device path produced, there is nothing here to put on the GPU.
// Not implemented: the NGMLR tag emulation ("Emulate NGMLR's tag output", off by default) -- requesting it throws.
#pragma once
#include "ma_modules.h"

#ifdef MA_WITH_ZLIB
#include <zlib.h>
#endif
#include <algorithm>
#include <cctype>
#include <iostream>
#include <mutex>
#include <sstream>

namespace libMA
{
#define MA_SAM_MULTIPLE_SEGMENTS_IN_TEMPLATE 0x001 // alignment.h:14-25
#define MA_SAM_SEGMENT_PROPERLY_ALIGNED 0x002
#define MA_SAM_SEGMENT_UNMAPPED 0x004
#define MA_SAM_NEXT_SEGMENT_UNMAPPED 0x008
#define MA_SAM_REVERSE_COMPLEMENTED 0x010
#define MA_SAM_NEXT_REVERSE_COMPLEMENTED 0x020
#define MA_SAM_FIRST_IN_TEMPLATE 0x040
#define MA_SAM_LAST_IN_TEMPLATE 0x080
#define MA_SAM_SECONDARY_ALIGNMENT 0x100
#define MA_SAM_SUPPLEMENTARY_ALIGNMENT 0x800

namespace sam
{
// ---- Pack (pack.h:900-997,1063-1067) on the host-side contig table
inline uint64_t fwdSize( const Pack& rPack )
{
    return rPack.vStarts.empty( ) ? 0 : rPack.vStarts.back( ) + rPack.vLengths.back( );
}
inline bool bPositionIsOnReversStrand( const Pack& rPack, uint64_t uiPosition )
{
    return uiPosition >= fwdSize( rPack );
}
inline int64_t iAbsolutePosition( const Pack& rPack, uint64_t uiPosition )
{
    return bPositionIsOnReversStrand( rPack, uiPosition ) ? (int64_t)( 2 * fwdSize( rPack ) - ( uiPosition + 1 ) )
                                                          : (int64_t)uiPosition;
}
inline int64_t iAbsolutePosition( const Pack& rPack, uint64_t uiBegin, uint64_t uiEnd )
{
    return bPositionIsOnReversStrand( rPack, uiEnd ) ? (int64_t)( 2 * fwdSize( rPack ) - ( uiEnd + 1 ) ) : (int64_t)uiBegin;
}
inline int64_t uiSequenceIdForAbsolute( const Pack& rPack, int64_t iAbsPosition ) // pack.h:945-990: the contig that starts last at or before it
{
    const auto xNext = std::upper_bound( rPack.vStarts.begin( ), rPack.vStarts.end( ), (uint64_t)std::max<int64_t>( iAbsPosition, 0 ) );
    return xNext == rPack.vStarts.begin( ) ? 0 : (int64_t)( xNext - rPack.vStarts.begin( ) ) - 1;
}
inline int64_t uiSequenceIdForPosition( const Pack& rPack, uint64_t uiPosition )
{
    return uiSequenceIdForAbsolute( rPack, iAbsolutePosition( rPack, uiPosition ) );
}
inline std::string nameOfSequenceForPosition( const Pack& rPack, uint64_t uiPosition )
{
    return rPack.vNames[ (size_t)uiSequenceIdForPosition( rPack, uiPosition ) ];
}
inline uint64_t posInSequence( const Pack& rPack, uint64_t uiBegin, uint64_t uiEnd )
{
    // (sic) the relative position is looked up with the ABSOLUTE position as the strand-aware one (pack.h:1065-1066)
    const int64_t uiPosition = iAbsolutePosition( rPack, uiBegin, uiEnd );
    return (uint64_t)uiPosition - rPack.vStarts[ (size_t)uiSequenceIdForPosition( rPack, (uint64_t)uiPosition ) ];
}

// ---- text assembly: a SAM record is appended to ONE buffer, number by number and base by base (no temporaries)
inline void appendNumber( std::string& rOut, uint64_t uiValue )
{
    char aDigits[ 24 ];
    int n = 0;
    do
    {
        aDigits[ n++ ] = (char)( '0' + uiValue % 10 );
        uiValue /= 10;
    } while( uiValue != 0 );
    while( n > 0 )
        rOut.push_back( aDigits[ --n ] );
}

// One SAM line: columns appended to a shared buffer in order, a tab between two of them, a line feed at the end.
class Columns
{
    std::string& rOut;
    bool bFirst = true;
    void separator( )
    {
        if( !bFirst )
            rOut.push_back( '\t' );
        bFirst = false;
    }

  public:
    explicit Columns( std::string& rOut ) : rOut( rOut )
    {}
    Columns& text( const std::string& sText )
    {
        separator( );
        rOut += sText;
        return *this;
    }
    Columns& number( uint64_t uiValue );
    template <typename FILL> Columns& column( FILL&& fFill ) // the column's text is appended by fFill( buffer )
    {
        separator( );
        fFill( rOut );
        return *this;
    }
    void end( )
    {
        rOut.push_back( '\n' );
    }
};

inline Columns& Columns::number( uint64_t uiValue )
{
    separator( );
    appendNumber( rOut, uiValue );
    return *this;
}

// ---- NucSeq (nucSeq.h:558-569,605-626,667-713)
inline char charOf( uint8_t c )
{
    return "ACGTN"[ c < 4 ? c : 4 ];
}
inline void appendFromTo( std::string& rOut, const NucSeq& rQ, nucSeqIndex uiStart, nucSeqIndex uiEnd )
{
    for( nucSeqIndex i = uiStart; i < uiEnd && i < rQ.length( ); i++ )
        rOut.push_back( charOf( rQ.xCodes[ i ] ) );
}
inline void appendFromToComplement( std::string& rOut, const NucSeq& rQ, nucSeqIndex uiStart, nucSeqIndex uiEnd )
{
    if( uiEnd > uiStart && uiEnd > rQ.length( ) )
        throw std::runtime_error( "Index out of range (compCharAt)" );
    for( nucSeqIndex k = 0; k < uiEnd - std::min( uiStart, uiEnd ); k++ )
    {
        const uint8_t c = rQ.xCodes[ uiEnd - 1 - k ];
        rOut.push_back( "TGCAN"[ c < 4 ? c : 4 ] ); // nucleotideComplement nucSeq.h:524-532: anything but ACGT stays N
    }
}
inline void appendFromToQual( std::string& rOut, const NucSeq& rQ, nucSeqIndex uiStart, nucSeqIndex uiEnd ) // nucSeq.h:697-709
{
    if( rQ.xQuality.empty( ) )
    {
        rOut.push_back( '*' );
        return;
    }
    for( nucSeqIndex i = uiStart; i < uiEnd && i < rQ.length( ); i++ )
        rOut.push_back( (char)rQ.xQuality[ i ] );
}
inline std::string fromTo( const NucSeq& rQ, nucSeqIndex uiStart, nucSeqIndex uiEnd )
{
    std::string ret;
    appendFromTo( ret, rQ, uiStart, uiEnd );
    return ret;
}
inline std::string fromToComplement( const NucSeq& rQ, nucSeqIndex uiStart, nucSeqIndex uiEnd )
{
    std::string ret;
    appendFromToComplement( ret, rQ, uiStart, uiEnd );
    return ret;
}
inline std::string toString( const NucSeq& rQ )
{
    return fromTo( rQ, 0, rQ.length( ) );
}
inline std::string fromToQual( const NucSeq& rQ, nucSeqIndex uiStart, nucSeqIndex uiEnd )
{
    std::string ret;
    appendFromToQual( ret, rQ, uiStart, uiEnd );
    return ret;
}

// ---- Alignment (alignment.h:367-467,576-623)
inline nucSeqIndex length( const Alignment& rA )
{
    nucSeqIndex n = 0;
    for( auto& x : rA.data )
        n += x.second;
    return n;
}
inline void appendClip( std::string& rOut, nucSeqIndex n, bool bSoftClip )
{
    appendNumber( rOut, n );
    rOut.push_back( bSoftClip ? 'S' : 'H' );
}
// Alignment::cigarString (alignment.h:367-467): clipping, then the sections in reference direction; with bM runs of
// matches, mismatches and seeds merge into one M
inline void appendCigar( std::string& rOut, const Alignment& rA, const Pack& rPack, size_t uiQuerySize, bool bSoftClip, bool bM )
{
    const bool bRev = bPositionIsOnReversStrand( rPack, rA.uiBeginOnRef );
    const nucSeqIndex uiHead = bRev ? ( rA.uiEndOnQuery < uiQuerySize ? uiQuerySize - rA.uiEndOnQuery : 0 ) : rA.uiBeginOnQuery;
    const nucSeqIndex uiTail = bRev ? rA.uiBeginOnQuery : ( rA.uiEndOnQuery < uiQuerySize ? uiQuerySize - rA.uiEndOnQuery : 0 );
    if( uiHead > 0 )
        appendClip( rOut, uiHead, bSoftClip );
    // one symbol per match type (seed, match, missmatch, insertion, deletion); with bM a run of the first three is one M
    static const char aSymbol[ 5 ] = { '=', '=', 'X', 'I', 'D' };
    nucSeqIndex uiRunOfM = 0;
    auto flushRun = [ & ]( ) {
        if( uiRunOfM > 0 )
        {
            appendNumber( rOut, uiRunOfM );
            rOut.push_back( 'M' );
        }
        uiRunOfM = 0;
    };
    const size_t uiSections = rA.data.size( );
    for( size_t k = 0; k < uiSections; k++ )
    {
        const auto& rSection = rA.data[ bRev ? uiSections - 1 - k : k ];
        const unsigned uiType = (unsigned)rSection.first;
        if( uiType > (unsigned)MatchType::deletion )
        {
            std::cerr << "WARNING invalid cigar symbol" << std::endl;
            continue;
        }
        const bool bIndel = uiType >= (unsigned)MatchType::insertion;
        if( bM && !bIndel )
        {
            uiRunOfM += rSection.second;
            continue;
        }
        if( bM )
            flushRun( );
        appendNumber( rOut, rSection.second );
        rOut.push_back( aSymbol[ uiType ] );
    }
    flushRun( );
    if( uiTail > 0 )
        appendClip( rOut, uiTail, bSoftClip );
}
inline std::string cigarString( const Alignment& rA, const Pack& rPack, size_t uiQuerySize, bool bSoftClip, bool bM )
{
    std::string sCigar;
    appendCigar( sCigar, rA, rPack, uiQuerySize, bSoftClip, bM );
    return sCigar;
}
inline uint32_t getSamFlag( const Alignment& rA, const Pack& rPack )
{
    uint32_t uiRet = 0;
    if( bPositionIsOnReversStrand( rPack, rA.uiBeginOnRef ) )
        uiRet |= MA_SAM_REVERSE_COMPLEMENTED;
    if( rA.bSecondary )
        uiRet |= MA_SAM_SECONDARY_ALIGNMENT;
    if( rA.bSupplementary )
        uiRet |= MA_SAM_SUPPLEMENTARY_ALIGNMENT;
    return uiRet;
}
inline nucSeqIndex getSamPosition( const Alignment& rA, const Pack& rPack )
{
    uint64_t uiRet = posInSequence( rPack, rA.uiBeginOnRef, rA.uiEndOnRef );
    if( bPositionIsOnReversStrand( rPack, rA.uiBeginOnRef ) )
        uiRet += 1;
    return uiRet + 1;
}
inline std::string getQuerySequence( const Alignment& rA, const NucSeq& rQuery, const Pack& rPack )
{
    std::string sRet = bPositionIsOnReversStrand( rPack, rA.uiBeginOnRef )
                           ? fromToComplement( rQuery, rA.uiBeginOnQuery, rA.uiEndOnQuery )
                           : fromTo( rQuery, rA.uiBeginOnQuery, rA.uiEndOnQuery );
    const int64_t iOff = (int64_t)sRet.length( ) - (int64_t)( rA.uiEndOnQuery - rA.uiBeginOnQuery );
    if( iOff != 0 )
        throw std::runtime_error( "Query length is off by " + std::to_string( iOff ) + "." );
    return sRet;
}
// ---- reference bases for the tags (Pack::vExtract pack.h:1440-1450, vExtractSubsectionN 1238-1330): codes of the doubled
// text [uiBegin, uiEnd) on one strand; with bMarkHoles the positions inside recorded runs of N read 4
inline std::vector<uint8_t> referenceCodes( const Pack& rPack, uint64_t uiBegin, uint64_t uiEnd, bool bMarkHoles )
{
    std::vector<uint8_t> vCodes( uiEnd > uiBegin ? uiEnd - uiBegin : 0 );
    if( vCodes.empty( ) )
        return vCodes;
    const uint64_t uiF = fwdSize( rPack );
    if( !rPack.vPacHost.empty( ) )
    {
        if( uiEnd > 2 * uiF || ( uiBegin >= uiF ) != ( uiEnd - 1 >= uiF ) )
            throw std::runtime_error( "(vExtractSubsection) Try to extract bridging sequence. This is impossible." );
        for( uint64_t p = uiBegin; p < uiEnd; p++ )
        {
            const uint64_t a = p < uiF ? p : 2 * uiF - 1 - p;
            const uint8_t b = ( rPack.vPacHost[ a >> 2 ] >> ( ( ~a & 3 ) << 1 ) ) & 3;
            vCodes[ p - uiBegin ] = p < uiF ? b : (uint8_t)( 3 - b );
        }
    }
    else
    {
        if( rPack.pDev == nullptr )
            throw std::runtime_error( "Pack: no reference bases available on the host or on the device" );
        maCheck( ma_pack_extract( rPack.pDev->p, &uiBegin, &uiEnd, 1, vCodes.data( ) ) );
    }
    if( bMarkHoles )
        for( const auto& rHole : rPack.vHoles )
            for( uint64_t p = uiBegin; p < uiEnd; p++ )
            {
                const uint64_t a = p < uiF ? p : 2 * uiF - 1 - p;
                if( rHole.first <= a && a < rHole.first + rHole.second )
                    vCodes[ p - uiBegin ] = 4;
            }
    return vCodes;
}
inline double amountOfRegionCoveredByHole( const Pack& rPack, uint64_t uiStart, uint64_t uiEnd ) // pack.h:551-566 (sic: raw positions)
{
    uint64_t uiCovered = 0;
    for( const auto& rHole : rPack.vHoles )
        if( rHole.first <= uiEnd && rHole.first + rHole.second > uiStart )
            uiCovered += std::min( uiEnd, rHole.first + rHole.second ) - std::max( uiStart, rHole.first );
    return (double)uiCovered / (double)( uiEnd - uiStart );
}
// Alignment::getNumDifferences (alignment.h:287-319): mismatches + inserted + deleted bases + reference Ns under matches
inline size_t getNumDifferences( const Alignment& rA, const Pack& rPack )
{
    const std::vector<uint8_t> vRef = referenceCodes( rPack, rA.uiBeginOnRef, rA.uiEndOnRef, true );
    size_t uiDiff = 0, uiRPos = 0;
    for( const auto& rSection : rA.data )
    {
        if( rSection.first == MatchType::seed || rSection.first == MatchType::match )
        {
            for( size_t i = 0; i < rSection.second; i++ )
                if( vRef[ i + uiRPos ] >= 4 )
                    uiDiff++;
        }
        else
            uiDiff += rSection.second;
        if( rSection.first != MatchType::insertion )
            uiRPos += rSection.second;
    }
    return uiDiff;
}
// the tags NGMLR writes, in its order (TagGenerator::computeTag with "Emulate NGMLR's tag output", fileWriter.h:120-326)
inline std::string ngmlrTags( const NucSeq& rQuery, const std::shared_ptr<Alignment>& pAlignment, const Pack& rPack,
                              const libMS::ContainerVector<std::shared_ptr<Alignment>>& rAll, bool bSoftClip )
{
    const Alignment& rA = *pAlignment;
    std::string sTag = "\tMD:Z:";
    auto appendTag = [ &sTag ]( const char* sKey, const std::string& sValue ) {
        sTag += sKey;
        sTag += sValue;
    };
    {
        const std::vector<uint8_t> vRef = referenceCodes( rPack, rA.uiBeginOnRef, rA.uiEndOnRef, false );
        size_t uiRPos = 0;
        nucSeqIndex uiPending = 0; // matches and seeds not yet written
        bool bAfterDeletion = false; // a mismatch right behind deleted bases is separated from them by a "0"
        for( const auto& rSection : rA.data )
        {
            const MatchType eType = rSection.first;
            if( eType == MatchType::insertion ) // invisible in MD (but it ends an "after deletion")
            {
                bAfterDeletion = false;
                continue;
            }
            if( eType == MatchType::match || eType == MatchType::seed )
                uiPending += rSection.second;
            else if( eType == MatchType::missmatch || eType == MatchType::deletion )
            {
                if( uiPending > 0 )
                    appendNumber( sTag, uiPending );
                uiPending = 0;
                if( eType == MatchType::deletion )
                    sTag.push_back( '^' );
                for( nucSeqIndex i = 0; i < rSection.second; i++ )
                {
                    // "0" = no matching base between two mismatching ones (or between a deletion and a mismatch)
                    if( eType == MatchType::missmatch && ( i > 0 || bAfterDeletion ) )
                        sTag.push_back( '0' );
                    sTag.push_back( charOf( vRef[ uiRPos + i ] ) );
                }
            }
            else
                throw std::runtime_error( "Invalid symbol in cigar!" );
            uiRPos += rSection.second;
            bAfterDeletion = eType == MatchType::deletion;
        }
        if( uiPending > 0 )
            appendNumber( sTag, uiPending );
    }
    {
        size_t uiSv = 0;
        if( amountOfRegionCoveredByHole( rPack, rA.uiBeginOnRef - 100, rA.uiBeginOnRef ) > .8 ||
            amountOfRegionCoveredByHole( rPack, rA.uiEndOnRef, rA.uiEndOnRef + 100 ) > .8 )
            uiSv += 1;
        if( rA.uiEndOnQuery - rA.uiBeginOnQuery >= rQuery.length( ) * 0.95 || bSoftClip )
            uiSv += 2;
        sTag.append( "\tSV:i:" ).append( std::to_string( uiSv ) );
    }
    const size_t uiNm = getNumDifferences( rA, rPack );
    sTag.append( "\tAS:i:" ).append( std::to_string( rA.score( ) ) );
    sTag.append( "\tNM:i:" ).append( std::to_string( uiNm ) );
    {
        size_t uiMatches = 0;
        for( const auto& rSection : rA.data )
            if( rSection.first == MatchType::seed || rSection.first == MatchType::match )
                uiMatches += rSection.second;
        appendTag( "\tXI:f:", std::to_string( uiMatches / (float)std::min( rA.uiEndOnQuery - rA.uiBeginOnQuery, rA.uiEndOnRef - rA.uiBeginOnRef ) ) );
    }
    sTag.append( "\tXE:i:" ).append( std::to_string( rA.score( ) ) ); // (sic) NGMLR puts the score here
    sTag.append( "\tXR:i:" ).append( std::to_string( rA.uiEndOnQuery - rA.uiBeginOnQuery ) );
    appendTag( "\tCV:f:", std::to_string( 100.0f * ( rA.uiEndOnQuery - rA.uiBeginOnQuery ) / (float)rQuery.length( ) ) );
    if( rAll.size( ) > 1 )
    {
        std::string sSisters;
        for( const auto& pOther : rAll )
        {
            if( pOther == pAlignment || pOther->bSecondary || pOther->xStats.bFirst != rA.xStats.bFirst )
                continue;
            sSisters += nameOfSequenceForPosition( rPack, pOther->uiBeginOnRef );
            sSisters.push_back( ',' );
            appendNumber( sSisters, getSamPosition( *pOther, rPack ) );
            sSisters += bPositionIsOnReversStrand( rPack, pOther->uiBeginOnRef ) ? ",-," : ",+,";
            appendCigar( sSisters, *pOther, rPack, rQuery.length( ), bSoftClip, true );
            sSisters.push_back( ',' );
            sSisters += std::isnan( pOther->fMappingQuality ) ? std::string( "255" )
                                                              : std::to_string( static_cast<int>( std::ceil( pOther->fMappingQuality * 254 ) ) );
            sSisters.push_back( ',' );
            appendNumber( sSisters, uiNm ); // (sic) the differences of THIS alignment, not of the sister
            sSisters.push_back( ';' );
        }
        if( !sSisters.empty( ) )
            sTag.append( "\tSA:Z:" ).append( sSisters );
    }
    sTag.append( "\tQS:i:" ).append( std::to_string( rA.uiBeginOnQuery ) ).append( "\tQE:i:" ).append( std::to_string( rA.uiEndOnQuery ) );
    return sTag;
}
// Alignment::invertSuccessiveInserionAndDeletion (alignment.h:328-345): an insertion directly followed by a deletion (or the
// other way round) swaps places; applied to reverse-strand alignments so that the order is the same for every read
inline void invertSuccessiveInsertionAndDeletion( Alignment& rA )
{
    for( size_t i = 1; i < rA.data.size( ); i++ )
    {
        const MatchType a = rA.data[ i - 1 ].first, b = rA.data[ i ].first;
        if( ( a == MatchType::insertion && b == MatchType::deletion ) || ( a == MatchType::deletion && b == MatchType::insertion ) )
        {
            std::swap( rA.data[ i - 1 ], rA.data[ i ] );
            i++;
        }
    }
}
// TagGenerator::computeTag (fileWriter.h:215-357) for the default options: only the CG tag of over-long cigars
inline std::string computeTag( const Alignment& rA, bool bLong )
{
    std::string sTag;
    if( bLong ) // fileWriter.h:327-357
    {
        sTag.append( "\tCG:B:I" );
        static const uint32_t aBamOp[ 5 ] = { 7, 7, 8, 1, 2 }; // seed and match "=", missmatch "X", insertion "I", deletion "D"
        for( const auto& rSection : rA.data )
        {
            sTag.push_back( ',' );
            appendNumber( sTag, (uint32_t)( rSection.second << 4 ) | ( (unsigned)rSection.first < 5 ? aBamOp[ (unsigned)rSection.first ] : 0 ) );
        }
    }
    return sTag;
}
} // namespace sam

// Output sinks of the writers (interface of fileWriter.h:21-78: `*pOut << text`).  Here every sink implements ONE
// primitive, put(); operator<< is the reference's spelling of it and stays virtual so that client subclasses written
// against the reference's OutStream keep working.
class OutStream
{
  protected:
    virtual void put( const char*, size_t )
    {} // the base class swallows its input (fileWriter.h:24-27)
  public:
    virtual OutStream& operator<<( std::string sText )
    {
        put( sText.data( ), sText.size( ) );
        return *this;
    }
    // a block of text (a whole batch of records) without a temporary string; a sink that only overrides operator<< -- client
    // code written against the reference's OutStream -- still receives it through that operator
    virtual void write( const char* pText, size_t uiLength )
    {
        *this << std::string( pText, uiLength );
    }
    virtual ~OutStream( )
    {}
};
class StdOutStream : public OutStream
{
  protected:
    void put( const char* p, size_t n ) override
    {
        fwrite( p, 1, n, stdout );
        fflush( stdout ); // every record is visible at once, like the reference's std::flush
    }

  public:
    void write( const char* p, size_t n ) override
    {
        put( p, n );
    }
};
class FileOutStream : public OutStream
{
    FILE* pHandle;

  protected:
    void put( const char* p, size_t n ) override
    {
        if( n != 0 && fwrite( p, 1, n, pHandle ) != n )
            throw std::runtime_error( "Unable to write to output file" );
        fflush( pHandle );
    }

  public:
    void write( const char* p, size_t n ) override
    {
        put( p, n );
    }
    explicit FileOutStream( std::string sFileName ) : pHandle( fopen( sFileName.c_str( ), "w" ) ) // truncates
    {
        if( pHandle == nullptr )
            throw std::runtime_error( "Unable to open file" + sFileName ); // (sic) no blank: the reference's text
    }
    FileOutStream( const FileOutStream& ) = delete;
    FileOutStream& operator=( const FileOutStream& ) = delete;
    ~FileOutStream( ) override
    {
        fclose( pHandle );
    }
};
class StringOutStream : public OutStream // convenience for tests and in-memory pipelines
{
  public:
    std::string sText;
    StringOutStream& operator<<( std::string s ) override
    {
        sText += s;
        return *this;
    }
    void write( const char* p, size_t n ) override
    {
        sText.append( p, n );
    }
};

namespace sam
{
// "stdout" or a file (fileWriter.h:385-400)
inline std::shared_ptr<OutStream> openSink( const std::string& sFileName )
{
    if( sFileName == "stdout" )
        return std::make_shared<StdOutStream>( );
    return std::make_shared<FileOutStream>( sFileName );
}
// @SQ line per contig, then @PG.  (sic) the file-name constructors separate SN and LN by a tab, the stream constructors
// by a blank (fileWriter.h:385-422,474-511)
inline void writeHeader( OutStream& rOut, const Pack& rPack, bool bTabBeforeLength )
{
    std::string sHeader;
    for( size_t i = 0; i < rPack.vNames.size( ); i++ )
    {
        sHeader += "@SQ\tSN:" + rPack.vNames[ i ];
        sHeader += bTabBeforeLength ? "\tLN:" : " LN:";
        appendNumber( sHeader, rPack.vLengths[ i ] );
        sHeader.push_back( '\n' );
    }
    sHeader += "@PG\tID:ma\tPN:ma\tVN:0.1.0\tCL:na\n";
    rOut << sHeader;
}
} // namespace sam

// What the two writers share: the sink, the lock that serialises the records of different graph threads on it (several
// writers may share both: fileWriter.h:430-436,520-541) and the SAM options of the parameter set.
class SamSink
{
  public:
    std::shared_ptr<OutStream> pOut;
    std::shared_ptr<std::mutex> pLock;
    const SamOptions xOptions;
    static const size_t uiMaxCigarLen = 0x10000; // cigars with more operations go to the CG tag (fileWriter.h:364)
    // 0 (default): every read's records are written under the lock as the reference does (fileWriter.cpp:141-145).
    // > 0: every calling thread collects its records and writes them when it has this many bytes; flush( ) -- or the
    // destructor -- writes what is left.  For graphs with more threads than the host grants cores: a thread that is descheduled
    // while it holds the writer's lock stalls all the others (measured: 54 of 60 us per read at 32 threads).  The records and
    // their order per thread are the same; call flush( ) (from one thread, after the graph threads are done) before the stream
    // is read.
    size_t uiBufferBytes = 0;

  private:
    struct ThreadBuffer
    {
        std::string sText;
    };
    mutable std::mutex xBuffersMutex;
    mutable std::vector<std::unique_ptr<ThreadBuffer>> vBuffers; // one per thread that ever emitted through this sink
    const uint64_t uiSinkId = nextSinkId( );
    static uint64_t nextSinkId( )
    {
        static std::atomic<uint64_t> uiNext{ 1 };
        return uiNext++;
    }
    ThreadBuffer& myBuffer( ) const
    {
        // this thread's buffers, by sink id (ids are never reused, so an entry of a sink that is gone is simply never found again)
        static thread_local std::vector<std::pair<uint64_t, ThreadBuffer*>> vMine;
        for( auto& rEntry : vMine )
            if( rEntry.first == uiSinkId )
                return *rEntry.second;
        std::lock_guard<std::mutex> xGuard( xBuffersMutex );
        vBuffers.emplace_back( new ThreadBuffer( ) );
        vMine.emplace_back( uiSinkId, vBuffers.back( ).get( ) );
        return *vBuffers.back( );
    }

  protected:
    SamSink( std::shared_ptr<OutStream> pOut, std::shared_ptr<std::mutex> pLock, const SamOptions& rOptions )
        : pOut( pOut ), pLock( pLock ), xOptions( rOptions )
    {}
    ~SamSink( )
    {
        flush( );
    }
    // the records of one read (or pair) leave as one block
    void emit( const std::string& sRecords ) const
    {
        if( sRecords.empty( ) )
            return;
        if( uiBufferBytes != 0 )
        {
            ThreadBuffer& rMine = myBuffer( );
            rMine.sText += sRecords;
            if( rMine.sText.size( ) < uiBufferBytes )
                return;
            std::unique_lock<std::mutex> xTurn( *pLock );
            pOut->write( rMine.sText.data( ), rMine.sText.size( ) );
            xTurn.unlock( );
            rMine.sText.clear( );
            return;
        }
        std::unique_lock<std::mutex> xTurn( *pLock );
        pOut->write( sRecords.data( ), sRecords.size( ) );
    }

  public:
    // writes what the threads' buffers still hold (uiBufferBytes > 0); not to be called while other threads emit
    void flush( ) const
    {
        std::lock_guard<std::mutex> xGuard( xBuffersMutex );
        for( auto& pBuffer : vBuffers )
            if( !pBuffer->sText.empty( ) )
            {
                std::unique_lock<std::mutex> xTurn( *pLock );
                pOut->write( pBuffer->sText.data( ), pBuffer->sText.size( ) );
                xTurn.unlock( );
                pBuffer->sText.clear( );
            }
    }
};

class FileWriter : public libMS::Module<libMS::Container, false, NucSeq, libMS::ContainerVector<std::shared_ptr<Alignment>>, Pack>,
                   public SamSink
{
  public:

    // fileWriter.h:385-400: "stdout" or a file name; header with tab-separated @SQ fields
    FileWriter( const ParameterSetManager& rParameters, std::string sFileName, std::shared_ptr<Pack> pPackContainer )
        : SamSink( sam::openSink( sFileName ), std::make_shared<std::mutex>( ), rParameters.xSam )
    {
        sam::writeHeader( *pOut, *pPackContainer, true );
    }
    // fileWriter.h:407-422 (sic: a blank, not a tab, before LN)
    FileWriter( const ParameterSetManager& rParameters, std::shared_ptr<OutStream> pOut_, std::shared_ptr<Pack> pPackContainer )
        : SamSink( pOut_, std::make_shared<std::mutex>( ), rParameters.xSam )
    {
        sam::writeHeader( *pOut, *pPackContainer, false );
    }
    // fileWriter.h:430-436: a second writer on the same stream
    FileWriter( const ParameterSetManager& rParameters, std::shared_ptr<FileWriter> pOther )
        : SamSink( pOther->pOut, pOther->pLock, rParameters.xSam )
    {}

    // fileWriter.cpp:11-158
    virtual std::shared_ptr<libMS::Container>
    execute( std::shared_ptr<NucSeq> pQuery, std::shared_ptr<libMS::ContainerVector<std::shared_ptr<Alignment>>> pAlignments,
             std::shared_ptr<Pack> pPack ) override
    {
        std::string sCombined;
        sCombined.reserve( pAlignments->size( ) * ( 2 * pQuery->length( ) + 96 ) + 64 );
        auto unmapped = [ & ]( const char* sMapQ ) { // fileWriter.cpp:126-140
            sCombined += pQuery->sName;
            sCombined.push_back( '\t' );
            sam::appendNumber( sCombined, MA_SAM_SEGMENT_UNMAPPED );
            sCombined += "\t*\t0\t";
            sCombined += sMapQ;
            sCombined += "\t*\t*\t0\t0\t";
            sam::appendFromTo( sCombined, *pQuery, 0, pQuery->length( ) );
            sCombined.push_back( '\t' );
            sam::appendFromToQual( sCombined, *pQuery, 0, pQuery->length( ) );
            sCombined.push_back( '\n' );
        };
        for( const std::shared_ptr<Alignment>& pAlignment : *pAlignments )
        {
            const Alignment& rA = *pAlignment;
            if( sam::length( rA ) == 0 )
                continue;
            if( ( xOptions.bNoSecondary && rA.bSecondary ) || ( xOptions.bNoSupplementary && rA.bSupplementary ) )
                continue;
            const bool bLong = xOptions.bCGTag && rA.data.size( ) >= uiMaxCigarLen;
            const bool bRev = sam::bPositionIsOnReversStrand( *pPack, rA.uiBeginOnRef );
            if( xOptions.bEmulateNgmlrTags && bRev ) // fileWriter.cpp:27-31
                sam::invertSuccessiveInsertionAndDeletion( *pAlignment );
            // QNAME FLAG RNAME POS MAPQ
            sCombined += pQuery->sName;
            sCombined.push_back( '\t' );
            sam::appendNumber( sCombined, sam::getSamFlag( rA, *pPack ) );
            sCombined.push_back( '\t' );
            sCombined += pPack->vNames[ (size_t)sam::uiSequenceIdForPosition( *pPack, rA.uiBeginOnRef ) ];
            sCombined.push_back( '\t' );
            sam::appendNumber( sCombined, sam::getSamPosition( rA, *pPack ) );
            sCombined.push_back( '\t' );
            if( std::isnan( rA.fMappingQuality ) )
                sCombined += "255";
            else
                sCombined += std::to_string( static_cast<int>( std::ceil( rA.fMappingQuality * 254 ) ) );
            sCombined.push_back( '\t' );
            // CIGAR RNEXT PNEXT TLEN
            if( bLong )
            {
                sam::appendNumber( sCombined, rA.uiEndOnQuery - rA.uiBeginOnQuery );
                sCombined.push_back( 'S' );
            }
            else
                sam::appendCigar( sCombined, rA, *pPack, pQuery->length( ), xOptions.bSoftClip,
                                  xOptions.bOutputMCigar || xOptions.bEmulateNgmlrTags );
            sCombined += "\t*\t0\t0\t";
            // SEQ: the whole read when soft clipping, else the aligned part; reverse-complemented on the reverse strand
            const nucSeqIndex uiFrom = xOptions.bSoftClip ? 0 : rA.uiBeginOnQuery, uiTo = xOptions.bSoftClip ? pQuery->length( ) : rA.uiEndOnQuery;
            const size_t uiBefore = sCombined.size( );
            if( bRev )
                sam::appendFromToComplement( sCombined, *pQuery, uiFrom, uiTo );
            else
                sam::appendFromTo( sCombined, *pQuery, uiFrom, uiTo );
            if( !xOptions.bSoftClip )
            {
                const int64_t iOff = (int64_t)( sCombined.size( ) - uiBefore ) - (int64_t)( rA.uiEndOnQuery - rA.uiBeginOnQuery );
                if( iOff != 0 )
                    throw std::runtime_error( "Query length is off by " + std::to_string( iOff ) + "." );
            }
            sCombined.push_back( '\t' );
            // QUAL (sic) not reversed for reverse-strand alignments (alignment.h:611-614), then the tags
            sam::appendFromToQual( sCombined, *pQuery, rA.uiBeginOnQuery, rA.uiEndOnQuery );
            if( xOptions.bEmulateNgmlrTags )
                sCombined += sam::ngmlrTags( *pQuery, pAlignment, *pPack, *pAlignments, xOptions.bSoftClip );
            if( bLong )
                sCombined += sam::computeTag( rA, bLong );
            sCombined.push_back( '\n' );
        }
        if( pAlignments->empty( ) )
            unmapped( "255" );
        else if( sCombined.empty( ) ) // every alignment was filtered out
            unmapped( "0" );
        emit( sCombined );
        return std::make_shared<libMS::Container>( );
    }
    virtual bool requiresLock( ) const
    {
        return false; // the writer serialises its own output
    }
};
// PairedFileWriter (fileWriter.h:456-545, fileWriter.cpp:158-380): SAM records of a mate pair
class PairedFileWriter : public libMS::Module<libMS::Container, false, NucSeq, NucSeq,
                                              libMS::ContainerVector<std::shared_ptr<Alignment>>, Pack>,
                         public SamSink
{
    void init( const SamOptions& rO )
    {
        if( rO.bEmulateNgmlrTags )
            throw std::runtime_error( "PairedFileWriter: the NGMLR tag emulation is not available in the MI355X host layer" );
    }

  public:
    // fileWriter.h:474-491
    PairedFileWriter( const ParameterSetManager& rParameters, std::string sFileName, std::shared_ptr<Pack> pPackContainer )
        : SamSink( sam::openSink( sFileName ), std::make_shared<std::mutex>( ), rParameters.xSam )
    {
        init( xOptions );
        sam::writeHeader( *pOut, *pPackContainer, true );
    }
    // fileWriter.h:498-511 (sic: a blank, not a tab, before LN)
    PairedFileWriter( const ParameterSetManager& rParameters, std::shared_ptr<OutStream> pOut_, std::shared_ptr<Pack> pPackContainer )
        : SamSink( pOut_, std::make_shared<std::mutex>( ), rParameters.xSam )
    {
        init( xOptions );
        sam::writeHeader( *pOut, *pPackContainer, false );
    }
    // fileWriter.h:520-541: further writers on the same stream
    PairedFileWriter( const ParameterSetManager& rParameters, std::shared_ptr<FileWriter> pOther )
        : SamSink( pOther->pOut, pOther->pLock, rParameters.xSam )
    {
        init( xOptions );
    }
    PairedFileWriter( const ParameterSetManager& rParameters, std::shared_ptr<PairedFileWriter> pOther )
        : SamSink( pOther->pOut, pOther->pLock, rParameters.xSam )
    {
        init( xOptions );
    }

    // fileWriter.cpp:158-383.  Every record of the pair goes through ONE emitter (sam::Columns below appends the eleven
    // mandatory columns to the pair's buffer in order); the three shapes of a record differ only in what the columns hold:
    //   aligned mate          FLAG = strand | secondary | supplementary | 0x1 | 0x2 | first/last | mate strand, RNEXT / PNEXT = the mate's
    //                         alignment ("=" on the same contig), CIGAR clipped against the length of the FIRST mate (sic, :191-193)
    //   pair without any      FLAG = 0x4 | 0x1 | first/last | 0x8, everything else empty, QUAL printed
    //   one mate unaligned    placed at the first alignment of the list (:348-366), RNEXT "=", QUAL "*"
    virtual std::shared_ptr<libMS::Container>
    execute( std::shared_ptr<NucSeq> pQuery1, std::shared_ptr<NucSeq> pQuery2,
             std::shared_ptr<libMS::ContainerVector<std::shared_ptr<Alignment>>> pAlignments, std::shared_ptr<Pack> pPack ) override
    {
        const NucSeq* const apMate[ 2 ] = { pQuery1.get( ), pQuery2.get( ) };
        const uint32_t aMateFlag[ 2 ] = { MA_SAM_FIRST_IN_TEMPLATE, MA_SAM_LAST_IN_TEMPLATE };
        bool aHasRecord[ 2 ] = { false, false };
        std::string sPair;
        sPair.reserve( pAlignments->size( ) * ( pQuery1->length( ) + pQuery2->length( ) + 96 ) + 2 * ( pQuery1->length( ) + pQuery2->length( ) ) + 128 );
        for( const std::shared_ptr<Alignment>& pAlignment : *pAlignments )
        {
            const Alignment& rA = *pAlignment;
            if( sam::length( rA ) == 0 || ( xOptions.bNoSecondary && rA.bSecondary ) || ( xOptions.bNoSupplementary && rA.bSupplementary ) )
                continue;
            const int iMate = rA.xStats.bFirst ? 0 : 1;
            const NucSeq& rMate = *apMate[ iMate ];
            aHasRecord[ iMate ] = true;
            const bool bRev = sam::bPositionIsOnReversStrand( *pPack, rA.uiBeginOnRef );
            const bool bLong = xOptions.bCGTag && rA.data.size( ) >= uiMaxCigarLen;
            const size_t uiContig = (size_t)sam::uiSequenceIdForPosition( *pPack, rA.uiBeginOnRef );
            const std::shared_ptr<Alignment> pPartner = rA.xStats.pOther.lock( );
            uint32_t uiFlag = sam::getSamFlag( rA, *pPack ) | MA_SAM_MULTIPLE_SEGMENTS_IN_TEMPLATE | MA_SAM_SEGMENT_PROPERLY_ALIGNED | aMateFlag[ iMate ];
            if( pPartner != nullptr && sam::bPositionIsOnReversStrand( *pPack, pPartner->uiBeginOnRef ) )
                uiFlag |= MA_SAM_NEXT_REVERSE_COMPLEMENTED;
            sam::Columns xLine( sPair );
            xLine.text( rMate.sName ).number( uiFlag ).text( pPack->vNames[ uiContig ] ).number( sam::getSamPosition( rA, *pPack ) );
            if( std::isnan( rA.fMappingQuality ) )
                xLine.text( "255" );
            else
                xLine.number( (uint64_t)std::min( static_cast<int>( std::ceil( rA.fMappingQuality * 254 ) ), 255 ) );
            xLine.column( [ & ]( std::string& rOut ) {
                if( bLong )
                {
                    sam::appendNumber( rOut, rA.uiEndOnQuery - rA.uiBeginOnQuery );
                    rOut.push_back( 'S' );
                }
                else
                    sam::appendCigar( rOut, rA, *pPack, pQuery1->length( ), xOptions.bSoftClip, xOptions.bOutputMCigar );
            } );
            if( pPartner != nullptr )
            {
                const size_t uiPartnerContig = (size_t)sam::uiSequenceIdForPosition( *pPack, pPartner->uiBeginOnRef );
                xLine.text( pPack->vNames[ uiPartnerContig ] == pPack->vNames[ uiContig ] ? std::string( "=" ) : pPack->vNames[ uiPartnerContig ] );
                xLine.number( sam::getSamPosition( *pPartner, *pPack ) );
            }
            else
                xLine.text( "*" ).text( "0" );
            xLine.text( "0" ); // the template length is not output by the reference (fileWriter.cpp:317)
            xLine.column( [ & ]( std::string& rOut ) {
                const nucSeqIndex uiFrom = xOptions.bSoftClip ? 0 : rA.uiBeginOnQuery, uiTo = xOptions.bSoftClip ? rMate.length( ) : rA.uiEndOnQuery;
                const size_t uiBefore = rOut.size( );
                if( bRev )
                    sam::appendFromToComplement( rOut, rMate, uiFrom, uiTo );
                else
                    sam::appendFromTo( rOut, rMate, uiFrom, uiTo );
                if( !xOptions.bSoftClip && rOut.size( ) - uiBefore != rA.uiEndOnQuery - rA.uiBeginOnQuery )
                    throw std::runtime_error( "Query length is off by " +
                                              std::to_string( (int64_t)( rOut.size( ) - uiBefore ) - (int64_t)( rA.uiEndOnQuery - rA.uiBeginOnQuery ) ) + "." );
            } );
            xLine.column( [ & ]( std::string& rOut ) { sam::appendFromToQual( rOut, rMate, rA.uiBeginOnQuery, rA.uiEndOnQuery ); } );
            if( bLong )
                sPair += sam::computeTag( rA, bLong );
            xLine.end( );
        }
        auto unaligned = [ & ]( int iMate, uint32_t uiExtraFlags, const std::string& sContig, const std::string& sPos, const char* sNext,
                                bool bWithQuality ) {
            const NucSeq& rMate = *apMate[ iMate ];
            sam::Columns xLine( sPair );
            xLine.text( rMate.sName ).number( MA_SAM_SEGMENT_UNMAPPED | MA_SAM_MULTIPLE_SEGMENTS_IN_TEMPLATE | aMateFlag[ iMate ] | uiExtraFlags );
            xLine.text( sContig ).text( sPos ).text( "0" ).text( "*" ).text( sNext ).text( sPos ).text( "0" );
            xLine.column( [ & ]( std::string& rOut ) { sam::appendFromTo( rOut, rMate, 0, rMate.length( ) ); } );
            if( bWithQuality )
                xLine.column( [ & ]( std::string& rOut ) { sam::appendFromToQual( rOut, rMate, 0, rMate.length( ) ); } );
            else
                xLine.text( "*" );
            xLine.end( );
        };
        if( !aHasRecord[ 0 ] && !aHasRecord[ 1 ] )
            for( int iMate = 0; iMate < 2; iMate++ )
                unaligned( iMate, MA_SAM_NEXT_SEGMENT_UNMAPPED, "*", "0", "*", true );
        else if( aHasRecord[ 0 ] != aHasRecord[ 1 ] )
        {
            const Alignment& rAnchor = *( *pAlignments )[ 0 ];
            unaligned( aHasRecord[ 0 ] ? 1 : 0, 0, sam::nameOfSequenceForPosition( *pPack, rAnchor.uiBeginOnRef ),
                       std::to_string( sam::getSamPosition( rAnchor, *pPack ) ), "=", false );
        }
        emit( sPair );
        return std::make_shared<libMS::Container>( );
    }
    virtual bool requiresLock( ) const
    {
        return false;
    }
};
// ---- FASTA / FASTQ reading (fileReader.h:28-200,475-496; fileReader.cpp:12-196 with WITH_QUALITY == 1) --------
class FileStream : public libMS::Container
{
  public:
    std::mutex xMutex;
    virtual bool eof( ) const = 0;
    virtual char peek( ) = 0;
    virtual char pop( ) = 0;
    virtual std::string fileName( ) = 0;
    virtual void safeGetLine( std::string& t ) = 0;
    // Optional: from now on append every character that is consumed to *pText (nullptr ends it).  A stream that can do so
    // lets BatchFileReader cut a batch of records out of the file under the stream's lock at memory speed and build the
    // reads outside of it.
    virtual bool capture( std::string* )
    {
        return false;
    }
    // The next line as a view that stays valid until the next call on the stream.  The default goes through safeGetLine;
    // a stream that holds its text in memory hands out a pointer into it and copies nothing.
    virtual void lineView( const char*& rpLine, size_t& ruiLength )
    {
        safeGetLine( sLineOfView );
        rpLine = sLineOfView.data( );
        ruiLength = sLineOfView.size( );
    }

  private:
    std::string sLineOfView;
};
// Character source over any std::istream, read a block at a time: peek / pop / one line work on memory, so a FASTQ file is
// scanned at memory speed (the per-character virtual calls and streambuf round trips of a plain istream wrapper cost ~10 ns
// per byte).  What callers observe is what the reference's StdFileStream / StringStream show them (fileReader.h:100-200):
//   - "\n", "\r\n" and a lone "\r" end a line; a last line without a line end still counts
//   - eof() turns true when a peek / pop / line finds nothing left -- not before (the record parser peeks first)
template <typename SOURCE> class BlockFileStream : public FileStream
{
  protected:
    SOURCE xSource;

  private:
    std::vector<char> vBlock;
    const char* pBlock = nullptr; // the current block: vBlock, or a text the stream adopted as its only block
    std::string sAdopted;
    bool bTextAdopted = false;
    size_t uiAt = 0, uiFilled = 0;
    bool bSawEnd = false;
    std::string sStraddling;
    std::string* pCaptured = nullptr; // consumed text goes here too ...
    size_t uiCapturedTo = 0; // ... what lies before this offset of the block already did

    // first '\n' or '\r' in [pFrom, pEnd), pEnd if there is none (two vectorised passes instead of a byte loop)
    static const char* lineEnd( const char* pFrom, const char* pEnd )
    {
        const char* pFeed = (const char*)memchr( pFrom, '\n', (size_t)( pEnd - pFrom ) );
        if( pFeed == nullptr )
            pFeed = pEnd;
        const char* const pReturn = (const char*)memchr( pFrom, '\r', (size_t)( pFeed - pFrom ) );
        return pReturn != nullptr ? pReturn : pFeed;
    }
    void flushCapture( )
    {
        if( pCaptured != nullptr && uiAt > uiCapturedTo )
            pCaptured->append( pBlock + uiCapturedTo, uiAt - uiCapturedTo );
        uiCapturedTo = uiAt;
    }
    bool available( ) // at least one unread character in the block
    {
        if( uiAt < uiFilled )
            return true;
        flushCapture( );
        uiAt = uiCapturedTo = uiFilled = 0;
        if( bTextAdopted )
            return false; // the adopted text was the whole stream
        if( vBlock.empty( ) )
            vBlock.resize( std::max<size_t>( uiBlockBytes, 1 ) );
        pBlock = vBlock.data( );
        xSource.read( vBlock.data( ), (std::streamsize)vBlock.size( ) );
        uiFilled = (size_t)xSource.gcount( );
        return uiFilled > 0;
    }

  public:
    size_t uiBlockBytes = 1u << 20; // size of the block; read before the first character is (tests shrink it)
    template <typename... ARGS> explicit BlockFileStream( ARGS&&... args ) : xSource( std::forward<ARGS>( args )... )
    {}
    // The stream IS this text from now on (taken over without a copy); the source is not read.
    void adoptText( std::string&& sText )
    {
        sAdopted = std::move( sText );
        bTextAdopted = true;
        pBlock = sAdopted.data( );
        uiAt = uiCapturedTo = 0;
        uiFilled = sAdopted.size( );
    }
    bool capture( std::string* pText ) override
    {
        flushCapture( );
        pCaptured = pText;
        uiCapturedTo = uiAt;
        return true;
    }
    bool eof( ) const override
    {
        return bSawEnd;
    }
    char peek( ) override
    {
        if( !available( ) )
        {
            bSawEnd = true;
            return (char)std::char_traits<char>::eof( );
        }
        return pBlock[ uiAt ];
    }
    char pop( ) override
    {
        const char cNext = peek( );
        if( !bSawEnd )
            uiAt++;
        return cNext;
    }
    void lineView( const char*& rpLine, size_t& ruiLength ) override
    {
        if( available( ) )
        {
            const char* const pFrom = pBlock + uiAt;
            const char* const pEnd = pBlock + uiFilled;
            const char* const pStop = lineEnd( pFrom, pEnd );
            // the whole line and what follows its end lie in the block: no copy
            if( pStop != pEnd && ( *pStop == '\n' || pStop + 1 != pEnd ) )
            {
                rpLine = pFrom;
                ruiLength = (size_t)( pStop - pFrom );
                uiAt += ruiLength + 1;
                if( *pStop == '\r' && pBlock[ uiAt ] == '\n' )
                    uiAt++;
                return;
            }
        }
        safeGetLine( sStraddling ); // a line across two blocks (or the last one of the text) is assembled
        rpLine = sStraddling.data( );
        ruiLength = sStraddling.size( );
    }
    void safeGetLine( std::string& rLine ) override
    {
        rLine.clear( );
        while( available( ) )
        {
            const char* const pFrom = pBlock + uiAt;
            const char* const pEnd = pBlock + uiFilled;
            const char* const pStop = lineEnd( pFrom, pEnd );
            rLine.append( pFrom, pStop );
            uiAt += (size_t)( pStop - pFrom );
            if( pStop == pEnd )
                continue; // the line goes on in the next block
            uiAt++; // the line end itself
            if( *pStop == '\r' && available( ) && pBlock[ uiAt ] == '\n' )
                uiAt++; // the second half of a Windows line end
            return;
        }
        if( rLine.empty( ) )
            bSawEnd = true; // nothing was left
    }
};
class StdFileStream : public BlockFileStream<std::ifstream>
{
    const std::string sFileName;

  public:
    StdFileStream( const std::string& sFilename ) : BlockFileStream<std::ifstream>( sFilename ), sFileName( sFilename )
    {
        if( !xSource.is_open( ) )
            throw std::runtime_error( "Unable to open file " + sFilename );
    }
    std::string fileName( ) override
    {
        return sFileName;
    }
};
#ifdef MA_WITH_ZLIB
// GzFileStream (fileReader.h:292-403, WITH_ZLIB): a gzip-compressed FASTA / FASTQ file read one byte ahead; a '\r' is
// dropped and a '\n' ends the line; (sic) pop() does not track the end of the file, safeGetLine does
class GzFileStream : public FileStream
{
    gzFile pFile = nullptr;
    int iLastRead = 0; // what the last gzread returned: 1 = a byte, 0 = end of the file, < 0 = failure
    unsigned char cBuff = 0;
    const std::string sFileName;
    void open( )
    {
        if( pFile == nullptr )
        {
            pFile = gzopen( sFileName.c_str( ), "rb" );
            iLastRead = pFile != nullptr ? gzread( pFile, &cBuff, 1 ) : -1;
        }
    }

  public:
    GzFileStream( const std::string& sFilename ) : sFileName( sFilename )
    {
        if( !std::ifstream( sFilename, std::ios::binary ).is_open( ) ) // the reference probes the file the same way before zlib opens it lazily
            throw std::runtime_error( "Unable to open file " + sFilename );
    }
    ~GzFileStream( )
    {
        if( pFile != nullptr )
            gzclose( pFile );
    }
    bool eof( ) const override
    {
        return iLastRead != 1;
    }
    char peek( ) override
    {
        open( );
        return (char)cBuff;
    }
    char pop( ) override
    {
        open( );
        const char cRet = (char)cBuff;
        gzread( pFile, &cBuff, 1 );
        return cRet;
    }
    std::string fileName( ) override // the stem, like fs::path::stem (fileReader.h:373-376)
    {
        const size_t uiSlash = sFileName.find_last_of( '/' );
        std::string sName = uiSlash == std::string::npos ? sFileName : sFileName.substr( uiSlash + 1 );
        const size_t uiDot = sName.find_last_of( '.' );
        return uiDot == std::string::npos || uiDot == 0 ? sName : sName.substr( 0, uiDot );
    }
    // One line of the decompressed text through the one-byte look-ahead cBuff.  (sic) A "\r" is dropped wherever it
    // stands -- before a "\n" it makes a Windows line end, anywhere else the line simply goes on (fileReader.h:395-420).
    void safeGetLine( std::string& rLine ) override
    {
        open( );
        rLine.clear( );
        auto have = [ this ]( ) { return iLastRead == 1; };
        auto advance = [ this ]( ) { iLastRead = gzread( pFile, &cBuff, 1 ); };
        while( have( ) )
        {
            if( cBuff == '\r' )
            {
                advance( );
                if( !have( ) )
                    return;
            }
            if( cBuff == '\n' )
            {
                advance( ); // step over the line end
                return;
            }
            rLine.push_back( (char)cBuff );
            advance( );
        }
    }
};
#endif
class StringStream : public BlockFileStream<std::istringstream>
{
  public:
    StringStream( const std::string& sString ) : BlockFileStream<std::istringstream>( )
    {
        adoptText( std::string( sString ) );
    }
    StringStream( std::string&& sString ) : BlockFileStream<std::istringstream>( )
    {
        adoptText( std::move( sString ) );
    }
    std::string fileName( ) override
    {
        return "StringStream";
    }
};

// FileStreamFromPath (fileReader.h:407-423): .gz files through zlib when the host layer is built with MA_WITH_ZLIB
inline std::shared_ptr<FileStream> fileStreamFromPath( const std::string& sFileName )
{
#ifdef MA_WITH_ZLIB
    if( sFileName.size( ) >= 3 && sFileName.compare( sFileName.size( ) - 3, 3, ".gz" ) == 0 )
        return std::make_shared<GzFileStream>( sFileName );
#endif
    return std::make_shared<StdFileStream>( sFileName );
}
// FileReader (fileReader.h:28-200,475-496; fileReader.cpp:12-203): one FASTA or FASTQ record per call.  The record grammar
// the reference accepts, as a scanner over the stream's lines:
//   header    '>' or '@' line; the name ends at the first blank
//   sequence  lines up to one that starts with the section's terminator ('+' for FASTQ, '>' for FASTA) or with a blank;
//             a line contributes its characters up to the last IUPAC nucleotide letter on it; empty lines are skipped
//   quality   (FASTQ, after a '+' line) lines up to one that starts with '@' -- but at least one, whatever it starts with:
//             a quality string may begin with '@'; characters beyond the sequence length are dropped
//   then everything up to the next '>' or '@' is skipped
class FileReader : public libMS::Module<NucSeq, true, FileStream>
{
    // character classes of the scanner: bit 0 = IUPAC nucleotide letter (either case), bit 1 = line end
    struct CharClasses
    {
        uint8_t a[ 256 ], aCode[ 256 ]; // aCode: A0 C1 G2 T3 (either case), everything else 4 (nucSeq.cpp:17-28)
        CharClasses( )
        {
            memset( a, 0, sizeof( a ) );
            memset( aCode, 4, sizeof( aCode ) );
            for( int k = 0; k < 4; k++ )
                aCode[ (uint8_t)"ACGT"[ k ] ] = aCode[ (uint8_t)"acgt"[ k ] ] = (uint8_t)k;
            for( const char* p = "ACGTUNRYKMSWBDHV"; *p; p++ )
                a[ (uint8_t)*p ] |= 1, a[ (uint8_t)tolower( *p ) ] |= 1;
            a[ (uint8_t)'\n' ] |= 2, a[ (uint8_t)'\r' ] |= 2;
        }
    };
    static const CharClasses& table( )
    {
        static const CharClasses xClasses;
        return xClasses;
    }
    static const uint8_t* classes( )
    {
        return table( ).a;
    }
    // length of a line without its trailing characters of the other class
    struct Line // view of one line of the stream (valid until the next one is asked for)
    {
        const char* p = nullptr;
        size_t n = 0;
        bool empty( ) const
        {
            return n == 0;
        }
        char operator[]( size_t i ) const
        {
            return p[ i ];
        }
    };
    static size_t trimmed( const Line& rLine, uint8_t uiKeepBit, bool bKeepIfSet )
    {
        size_t n = rLine.n;
        while( n > 0 && ( ( classes( )[ (uint8_t)rLine[ n - 1 ] ] & uiKeepBit ) != 0 ) != bKeepIfSet )
            n--;
        return n;
    }
    struct Scanner
    {
        FileStream& rS;
        bool more( ) const
        {
            return !rS.eof( );
        }
        bool nextStartsWith( char c )
        {
            return rS.peek( ) == c;
        }
        Line line( )
        {
            Line xLine;
            rS.lineView( xLine.p, xLine.n );
            return xLine;
        }
    };

  public:
    FileReader( const ParameterSetManager& )
    {}
    // One record off the stream, nullptr at its end.  The caller holds the stream's lock.
    static std::shared_ptr<NucSeq> parseRecord( FileStream& rStream )
    {
        return scanRecord<true>( rStream );
    }
    // The same walk over one record without building it: true if there was one.  (With FileStream::capture this cuts the
    // record's text out of the stream.)
    static bool skipRecord( FileStream& rStream )
    {
        return scanRecord<false>( rStream ) != nullptr;
    }

  private:
    template <bool BUILD> static std::shared_ptr<NucSeq> scanRecord( FileStream& rStream )
    {
        Scanner xIn{ rStream };
        rStream.peek( );
        if( !xIn.more( ) )
            return nullptr;
        const char cKind = rStream.peek( );
        if( cKind != '>' && cKind != '@' )
            throw std::runtime_error( "Error while reading file.\nIs your input really in FASTA/Q format?\nError occurred in file: " +
                                      rStream.fileName( ) + "\npeek was:" + cKind );
        const char cTerminator = cKind == '@' ? '+' : '>';
        auto pRead = std::make_shared<NucSeq>( );
        size_t uiBases = 0;
        // ---- header
        {
            const Line xHeader = xIn.line( );
            if( xHeader.empty( ) )
                throw std::runtime_error( "Invalid line in fasta" );
            const char* const pBlank = (const char*)memchr( xHeader.p, ' ', xHeader.n );
            const size_t uiNameEnd = pBlank == nullptr ? xHeader.n : std::max<size_t>( (size_t)( pBlank - xHeader.p ), 1 );
            pRead->sName.assign( xHeader.p + 1, uiNameEnd - 1 );
        }
        // ---- sequence
        while( xIn.more( ) && !xIn.nextStartsWith( cTerminator ) && !xIn.nextStartsWith( ' ' ) )
        {
            const Line sBases = xIn.line( );
            const size_t n = trimmed( sBases, 1, true );
            uiBases += n;
            if( BUILD )
            {
                const size_t uiOld = pRead->xCodes.size( );
                pRead->xCodes.resize( uiOld + n );
                const uint8_t* const pCodeOf = table( ).aCode;
                for( size_t i = 0; i < n; i++ )
                    pRead->xCodes[ uiOld + i ] = pCodeOf[ (uint8_t)sBases[ i ] ];
            }
        }
        // ---- quality
        if( cKind == '@' )
        {
            if( BUILD )
                pRead->xQuality.assign( pRead->xCodes.size( ), 0 );
            const Line sPlus = xIn.line( );
            if( !sPlus.empty( ) && sPlus[ 0 ] == '+' )
            {
                size_t uiFilled = 0;
                while( xIn.more( ) && ( uiFilled == 0 || !xIn.nextStartsWith( '@' ) ) )
                {
                    const Line sQual = xIn.line( );
                    const size_t n = trimmed( sQual, 2, false );
                    if( BUILD && uiFilled < pRead->xQuality.size( ) )
                        memcpy( pRead->xQuality.data( ) + uiFilled, sQual.p, std::min( n, pRead->xQuality.size( ) - uiFilled ) );
                    uiFilled += n;
                }
            }
        }
        if( uiBases == 0 )
            throw std::runtime_error( "found empty read: " + pRead->sName );
        // ---- whatever follows, up to the next record
        for( rStream.peek( ); xIn.more( ) && !xIn.nextStartsWith( '>' ) && !xIn.nextStartsWith( '@' ); rStream.peek( ) )
            rStream.pop( );
        return pRead;
    }

  public:
    // nullptr = end of file (volatile source, module.h:688-695)
    virtual std::shared_ptr<NucSeq> execute( std::shared_ptr<FileStream> pStream ) override
    {
        std::lock_guard<std::mutex> xLock( pStream->xMutex );
        return parseRecord( *pStream );
    }
};
// PairedFileStream / PairedFileReader (fileReader.h:499-617): one read from each of two streams per call
class PairedFileStream : public FileStream, public std::pair<std::shared_ptr<FileStream>, std::shared_ptr<FileStream>>
{
    // a pair of streams is not itself a character stream: the single-stream operations have no meaning here
    [[noreturn]] static void noSingleStream( )
    {
        throw std::runtime_error( "This function should have been overridden" );
    }

  public:
    typedef std::pair<std::shared_ptr<FileStream>, std::shared_ptr<FileStream>> TP_MATES;
    using TP_MATES::TP_MATES;
    bool eof( ) const override // the shorter file ends the pair stream
    {
        return first->eof( ) ? true : second->eof( );
    }
    std::string fileName( ) override
    {
        std::string sBoth = first->fileName( );
        sBoth += ',';
        sBoth += second->fileName( );
        return sBoth;
    }
    char peek( ) override
    {
        noSingleStream( );
    }
    char pop( ) override
    {
        noSingleStream( );
    }
    void safeGetLine( std::string& ) override
    {
        noSingleStream( );
    }
};
typedef libMS::ContainerVector<std::shared_ptr<NucSeq>> PairedReadsContainer;
class PairedFileReader : public libMS::Module<PairedReadsContainer, true, PairedFileStream>
{
  public:
    FileReader xFileReader;
    const bool bRevCompMate;
    PairedFileReader( const ParameterSetManager& rParameters )
        : xFileReader( rParameters ), bRevCompMate( rParameters.bRevCompPairedReadMates )
    {}
    virtual std::shared_ptr<PairedReadsContainer> execute( std::shared_ptr<PairedFileStream> pFileStreamIn ) override
    {
        // one record off each file; the pair stream ends with the shorter file (both files are read before the check, like
        // the reference does: fileReader.h:590-600)
        const std::shared_ptr<NucSeq> apMate[ 2 ] = { xFileReader.execute( pFileStreamIn->first ), xFileReader.execute( pFileStreamIn->second ) };
        if( apMate[ 0 ] == nullptr || apMate[ 1 ] == nullptr )
            return nullptr;
        if( bRevCompMate )
        {
            // NucSeq::vReverse (nucSeq.h:373-381, with the qualities) + vSwitchAllBasePairsToComplement (537-543)
            NucSeq& rMate = *apMate[ 1 ];
            std::reverse( rMate.xCodes.begin( ), rMate.xCodes.end( ) );
            std::reverse( rMate.xQuality.begin( ), rMate.xQuality.end( ) );
            for( uint8_t& c : rMate.xCodes )
                c = c < 4 ? (uint8_t)( 3 - c ) : (uint8_t)5;
        }
        return std::make_shared<PairedReadsContainer>( std::begin( apMate ), std::end( apMate ) );
    }
};

// Genome FASTA -> index on the GPU: the contigs as FileReader returns them (names up to the first blank, IUPAC codes and N
// become N), Ns replaced like Pack::vAppendSequence does (buildIndex).  (The reference ingests genomes with a second FASTA
// parser, FastaStreamReader in pack.cpp:510-537; record names and sequences agree for plain FASTA files.)
inline void buildIndexFromFasta( const std::string& sFastaFilePath, std::shared_ptr<Pack>& pPack, std::shared_ptr<FMIndex>& pFM )
{
    ParameterSetManager xParameters;
    FileReader xReader( xParameters );
    auto pStream = fileStreamFromPath( sFastaFilePath );
    std::vector<std::shared_ptr<NucSeq>> vContigs;
    while( auto pContig = xReader.execute( pStream ) )
        vContigs.push_back( pContig );
    if( vContigs.empty( ) )
        throw std::runtime_error( "File open error." );
    buildIndex( vContigs, pPack, pFM );
}
} // namespace libMA
