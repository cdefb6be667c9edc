// ma_sam.h -- SAM emission of the drop-in host layer: FileWriter with the reference's name,
// Module signature, constructors, options and output bytes (libs/ma/inc/ma/module/fileWriter.h:21-78,364-440,
// libs/ma/src/module/fileWriter.cpp:11-158), plus the Alignment / Pack / NucSeq string helpers it calls
// (alignment.h:367-467,576-623; pack.h:900-997,1063-1067; nucSeq.h:558-713).  Pure host code: SAM text is
// formatting of what the device path produced, there is nothing here to put on the GPU.
// Not implemented: the NGMLR tag emulation ("Emulate NGMLR's tag output", off by default) -- requesting it throws.
#pragma once
#include "ma_modules.h"

#include <algorithm>
#include <iostream>
#include <mutex>

namespace libMA
{
#define MA_SAM_SEGMENT_UNMAPPED 0x004 // alignment.h:16-25
#define MA_SAM_REVERSE_COMPLEMENTED 0x010
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
inline int64_t uiSequenceIdForAbsolute( const Pack& rPack, int64_t iAbsPosition ) // binary search of pack.h:945-990
{
    uint64_t uiLeft = 0, uiMid = 0, uiRight = rPack.vStarts.size( );
    while( uiLeft < uiRight )
    {
        uiMid = ( uiLeft + uiRight ) / 2;
        if( iAbsPosition >= (int64_t)rPack.vStarts[ uiMid ] )
        {
            if( uiMid == rPack.vStarts.size( ) - 1 )
                break;
            if( iAbsPosition < (int64_t)rPack.vStarts[ uiMid + 1 ] )
                break;
            uiLeft = uiMid + 1;
        }
        else
            uiRight = uiMid;
    }
    return (int64_t)uiMid;
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

// ---- NucSeq (nucSeq.h:558-569,605-626,667-713)
inline char charOf( uint8_t c )
{
    static const char chars[ 4 ] = { 'A', 'C', 'G', 'T' };
    return c < 4 ? chars[ c ] : 'N';
}
inline std::string fromTo( const NucSeq& rQ, nucSeqIndex uiStart, nucSeqIndex uiEnd )
{
    std::string ret;
    for( nucSeqIndex i = uiStart; i < uiEnd && i < rQ.length( ); i++ )
        ret += charOf( rQ.xCodes[ i ] );
    return ret;
}
inline std::string fromToComplement( const NucSeq& rQ, nucSeqIndex uiStart, nucSeqIndex uiEnd )
{
    std::string ret;
    for( nucSeqIndex i = uiEnd; i > uiStart; i-- )
    {
        if( i - 1 >= rQ.length( ) )
            throw std::runtime_error( "Index out of range (compCharAt)" );
        const uint8_t c = rQ.xCodes[ i - 1 ];
        ret += charOf( c < 4 ? (uint8_t)( 3 - c ) : (uint8_t)5 ); // nucleotideComplement nucSeq.h:524-532
    }
    return ret;
}
inline std::string toString( const NucSeq& rQ )
{
    return fromTo( rQ, 0, rQ.length( ) );
}

// ---- Alignment (alignment.h:367-467,576-623)
inline nucSeqIndex length( const Alignment& rA )
{
    nucSeqIndex n = 0;
    for( auto& x : rA.data )
        n += x.second;
    return n;
}
inline std::string clip( nucSeqIndex n, bool bSoftClip )
{
    return std::to_string( n ) + ( bSoftClip ? "S" : "H" );
}
inline std::string cigarString( const Alignment& rA, const Pack& rPack, size_t uiQuerySize, bool bSoftClip, bool bM )
{
    const bool bRev = bPositionIsOnReversStrand( rPack, rA.uiBeginOnRef );
    std::string sCigar;
    if( bRev )
    {
        if( rA.uiEndOnQuery < uiQuerySize )
            sCigar += clip( uiQuerySize - rA.uiEndOnQuery, bSoftClip );
    }
    else if( rA.uiBeginOnQuery > 0 )
        sCigar += clip( rA.uiBeginOnQuery, bSoftClip );
    std::vector<std::pair<MatchType, nucSeqIndex>> vData( rA.data );
    if( bRev )
        std::reverse( vData.begin( ), vData.end( ) );
    size_t uiSequentialM = 0;
    for( auto& section : vData )
        switch( section.first )
        {
            case MatchType::seed:
            case MatchType::match:
                if( bM )
                    uiSequentialM += section.second;
                else
                    sCigar += std::to_string( section.second ) + "=";
                break;
            case MatchType::missmatch:
                if( bM )
                    uiSequentialM += section.second;
                else
                    sCigar += std::to_string( section.second ) + "X";
                break;
            case MatchType::insertion:
            case MatchType::deletion:
                if( bM && uiSequentialM > 0 )
                {
                    sCigar += std::to_string( uiSequentialM ) + "M";
                    uiSequentialM = 0;
                }
                sCigar += std::to_string( section.second ) + ( section.first == MatchType::insertion ? "I" : "D" );
                break;
            default:
                std::cerr << "WARNING invalid cigar symbol" << std::endl;
                break;
        }
    if( bM && uiSequentialM > 0 )
        sCigar += std::to_string( uiSequentialM ) + "M";
    if( bRev )
    {
        if( rA.uiBeginOnQuery > 0 )
            sCigar += clip( rA.uiBeginOnQuery, bSoftClip );
    }
    else if( rA.uiEndOnQuery < uiQuerySize )
        sCigar += clip( uiQuerySize - rA.uiEndOnQuery, bSoftClip );
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
} // namespace sam

class OutStream // fileWriter.h:21-34
{
  public:
    virtual OutStream& operator<<( std::string )
    {
        return *this;
    }
    virtual ~OutStream( )
    {}
};
class StdOutStream : public OutStream
{
  public:
    StdOutStream& operator<<( std::string s ) override
    {
        std::cout << s << std::flush;
        return *this;
    }
};
class FileOutStream : public OutStream
{
  public:
    std::ofstream file;
    FileOutStream( std::string sFileName ) : file( sFileName, std::ofstream::out | std::ofstream::trunc )
    {
        if( !file.good( ) )
            throw std::runtime_error( "Unable to open file" + sFileName );
    }
    ~FileOutStream( )
    {
        file.close( );
    }
    FileOutStream& operator<<( std::string s ) override
    {
        file << s << std::flush;
        return *this;
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
};

class FileWriter : public libMS::Module<libMS::Container, false, NucSeq, libMS::ContainerVector<std::shared_ptr<Alignment>>, Pack>
{
    static const size_t uiMaxCigarLen = 0x10000;
    void init( const SamOptions& rO )
    {
        if( rO.bEmulateNgmlrTags )
            throw std::runtime_error( "FileWriter: the NGMLR tag emulation is not available in the MI355X host layer" );
    }

  public:
    std::shared_ptr<OutStream> pOut;
    std::shared_ptr<std::mutex> pLock;
    const SamOptions xOptions;

    // fileWriter.h:385-400: "stdout" or a file name; header with tab-separated @SQ fields
    FileWriter( const ParameterSetManager& rParameters, std::string sFileName, std::shared_ptr<Pack> pPackContainer )
        : pLock( new std::mutex ), xOptions( rParameters.xSam )
    {
        init( xOptions );
        if( sFileName != "stdout" )
            pOut = std::shared_ptr<OutStream>( new FileOutStream( sFileName ) );
        else
            pOut = std::shared_ptr<OutStream>( new StdOutStream( ) );
        for( size_t i = 0; i < pPackContainer->vNames.size( ); i++ )
            *pOut << "@SQ\tSN:" << pPackContainer->vNames[ i ] << "\tLN:" << std::to_string( pPackContainer->vLengths[ i ] )
                  << "\n";
        *pOut << "@PG\tID:ma\tPN:ma\tVN:0.1.0\tCL:na\n";
    }
    // fileWriter.h:407-422 (sic: a blank, not a tab, before LN)
    FileWriter( const ParameterSetManager& rParameters, std::shared_ptr<OutStream> pOut_, std::shared_ptr<Pack> pPackContainer )
        : pOut( pOut_ ), pLock( new std::mutex ), xOptions( rParameters.xSam )
    {
        init( xOptions );
        for( size_t i = 0; i < pPackContainer->vNames.size( ); i++ )
            *pOut << "@SQ\tSN:" << pPackContainer->vNames[ i ] << " LN:" << std::to_string( pPackContainer->vLengths[ i ] )
                  << "\n";
        *pOut << "@PG\tID:ma\tPN:ma\tVN:0.1.0\tCL:na\n";
    }
    // fileWriter.h:430-436: a second writer on the same stream
    FileWriter( const ParameterSetManager& rParameters, std::shared_ptr<FileWriter> pOther )
        : pOut( pOther->pOut ), pLock( pOther->pLock ), xOptions( rParameters.xSam )
    {
        init( xOptions );
    }

    // fileWriter.cpp:11-158
    virtual std::shared_ptr<libMS::Container>
    execute( std::shared_ptr<NucSeq> pQuery, std::shared_ptr<libMS::ContainerVector<std::shared_ptr<Alignment>>> pAlignments,
             std::shared_ptr<Pack> pPack ) override
    {
        std::string sCombined;
        for( std::shared_ptr<Alignment> pAlignment : *pAlignments )
        {
            if( sam::length( *pAlignment ) == 0 )
                continue;
            if( xOptions.bNoSecondary && pAlignment->bSecondary )
                continue;
            if( xOptions.bNoSupplementary && pAlignment->bSupplementary )
                continue;
            const bool bLong = xOptions.bCGTag && pAlignment->data.size( ) >= uiMaxCigarLen;
            std::string sCigar;
            if( bLong )
                sCigar = std::to_string( pAlignment->uiEndOnQuery - pAlignment->uiBeginOnQuery ).append( "S" );
            else
                sCigar = sam::cigarString( *pAlignment, *pPack, pQuery->length( ), xOptions.bSoftClip, xOptions.bOutputMCigar );
            const uint32_t flag = sam::getSamFlag( *pAlignment, *pPack );
            std::string sSegment;
            if( xOptions.bSoftClip )
                sSegment = sam::bPositionIsOnReversStrand( *pPack, pAlignment->uiBeginOnRef )
                               ? sam::fromToComplement( *pQuery, 0, pQuery->length( ) )
                               : sam::toString( *pQuery );
            else
                sSegment = sam::getQuerySequence( *pAlignment, *pQuery, *pPack );
            const std::string sQual = "*"; // reads carry no qualities here (nucSeq.h:697-709 without WITH_QUALITY data)
            const std::string sRefName = sam::nameOfSequenceForPosition( *pPack, pAlignment->uiBeginOnRef );
            const nucSeqIndex uiRefPos = sam::getSamPosition( *pAlignment, *pPack );
            std::string sTag;
            if( bLong ) // fileWriter.h:327-357
            {
                sTag.append( "\tCG:B:I" );
                for( auto& rPair : pAlignment->data )
                {
                    uint32_t uiOperation = 0;
                    switch( rPair.first )
                    {
                        case MatchType::seed:
                        case MatchType::match:
                            uiOperation = 7;
                            break;
                        case MatchType::missmatch:
                            uiOperation = 8;
                            break;
                        case MatchType::insertion:
                            uiOperation = 1;
                            break;
                        case MatchType::deletion:
                            uiOperation = 2;
                            break;
                        default:
                            break;
                    }
                    sTag.append( "," ).append( std::to_string( (uint32_t)( rPair.second << 4 ) | uiOperation ) );
                }
            }
            std::string sMapQual;
            if( std::isnan( pAlignment->fMappingQuality ) )
                sMapQual = "255";
            else
                sMapQual = std::to_string( static_cast<int>( std::ceil( pAlignment->fMappingQuality * 254 ) ) );
            sCombined += pQuery->sName + "\t" + std::to_string( flag ) + "\t" + sRefName + "\t" + std::to_string( uiRefPos ) +
                         "\t" + sMapQual + "\t" + sCigar + "\t*\t0\t0\t" + sSegment + "\t" + sQual + sTag + "\n";
        }
        if( pAlignments->size( ) == 0 )
            sCombined += pQuery->sName + "\t" + std::to_string( MA_SAM_SEGMENT_UNMAPPED ) + "\t*\t0\t255\t*\t*\t0\t0\t" +
                         sam::toString( *pQuery ) + "\t*\n";
        if( sCombined.size( ) == 0 )
            sCombined += pQuery->sName + "\t" + std::to_string( MA_SAM_SEGMENT_UNMAPPED ) + "\t*\t0\t0\t*\t*\t0\t0\t" +
                         sam::toString( *pQuery ) + "\t*\n";
        {
            std::lock_guard<std::mutex> xGuard( *pLock );
            *pOut << sCombined;
        }
        return std::make_shared<libMS::Container>( );
    }
    virtual bool requiresLock( ) const
    {
        return false; // the writer serialises its own output
    }
};
} // namespace libMA
