// TEST INFRASTRUCTURE ONLY -- drives the REAL reference (ITBE-Lab/ma, compiled from /root/reference into
// oracle/_ref/libma_ref.so by Makefile.ref) and dumps per-stage results in the text format that
// oracle_dump (our CPU restatement) also emits, so the two can be diffed byte-for-byte.
// This file is our own harness; it includes the reference's public headers where they lie and
// never travels to the GPU box in compiled-from-reference form other than oracle/_ref/.
//
// usage:
//   ref_dump index <case> <out_prefix>                 -> <prefix>.pac/.ann/.amb/.bwt/.sa (reference writers)
//   ref_dump pipe  <case> <preset> <srand_seed> <out>   -> per-read stage dump
//   ref_dump ext   <case> <out>                        -> extend_backward traces
//   ref_dump ksw   <kswcase> <out> [dirty]             -> kswcpp_dispatch results
//   ref_dump time  <case> <preset> <threads>           -> reads/s of the reference's modules (seeding .. mapping quality)
//   ref_dump pipeidx <prefix> <case> <preset> <seed> <out> / timeidx <prefix> <case> <preset> <threads>
//                                                      -> the same with the index LOADED from <prefix>.* files
//   ref_dump readpair <in1> <in2> <out> <revcomp mate 0|1> -> mate pairs of the PairedFileReader
//   ref_dump f4    <case> <preset> <seed> <out> <inversions 0|1> <paired 0|1> <zdrop_inversion> [<out.sam> [<sam options>]]
//                                                      -> SmallInversions / PairedReads lists (+ SAM of the (Paired)FileWriter)
#include "ma/container/fMIndex.h"
#include "ma/container/pack.h"
#include "ma/module/binarySeeding.h"
#include "ma/module/fileReader.h"
#include "ma/module/fileWriter.h"
#include "ma/module/harmonization.h"
#include "ma/module/mappingQuality.h"
#include "ma/module/needlemanWunsch.h"
#include "ma/module/pairedReads.h"
#include "ma/module/smallInversions.h"
#include "ma/module/stripOfConsideration.h"
#include "kswcpp.h"
#include "dump_format.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

using namespace libMA;
using namespace libMS;

static std::shared_ptr<NucSeq> mkSeq( const std::vector<uint8_t>& v )
{
    auto p = std::make_shared<NucSeq>( );
    if( !v.empty( ) )
        p->vAppend( v.data( ), v.size( ) );
    return p;
}

struct RefIndex
{
    std::shared_ptr<Pack> pPack;
    std::shared_ptr<FMIndex> pFM;
};

static const char* g_sIndexPrefix = nullptr; // pipeidx / timeidx: load <prefix>.bwt/.sa/.pac/.ann/.amb instead of building

static RefIndex buildIndex( const CaseFile& c )
{
    RefIndex r;
    if( g_sIndexPrefix )
    {
        // the reference's own loaders (pack.h:799-812, fMIndex.h:886-900): proves that an index written by the MI355X
        // host layer (storeIndex) is one the reference accepts
        r.pPack = std::make_shared<Pack>( );
        r.pPack->vLoadCollection( g_sIndexPrefix );
        r.pFM = std::make_shared<FMIndex>( );
        r.pFM->vLoadFMIndex( g_sIndexPrefix );
        return r;
    }
    r.pPack = std::make_shared<Pack>( );
    for( size_t i = 0; i < c.contigs.size( ); i++ )
        r.pPack->vAppendSequence( c.names[ i ], "", *mkSeq( c.contigs[ i ] ) );
    r.pFM = std::make_shared<FMIndex>( r.pPack );
    return r;
}

// "<preset>+mems" = the preset with "Seeding Technique" switched to its third choice, MEMs (parameter.h:671-675;
// no preset of the reference selects it)
static void selectPreset( ParameterSetManager& xParams, const char* sPreset )
{
    std::string s( sPreset );
    const bool bMems = s.size( ) > 5 && s.compare( s.size( ) - 5, 5, "+mems" ) == 0;
    if( bMems )
        s.resize( s.size( ) - 5 );
    xParams.setSelected( s );
    if( bMems )
        xParams.getSelected( )->xSeedingTechnique->set( 2 );
}

static int cmdIndex( const char* sCase, const char* sPrefix )
{
    CaseFile c = readCase( sCase );
    RefIndex idx = buildIndex( c );
    idx.pPack->vStoreCollection( sPrefix );
    idx.pFM->vStoreFMIndex( sPrefix );
    return 0;
}

static void dumpSeeds( FILE* f, const char* tag, Seeds& s )
{
    for( auto& x : s )
        fprintf( f, "%s %llu %llu %llu %u %d %llu\n", tag, (unsigned long long)x.start( ), (unsigned long long)x.size( ),
                 (unsigned long long)x.start_ref( ), x.uiAmbiguity, (int)x.bOnForwStrand,
                 (unsigned long long)x.uiDelta );
}

static int cmdPipe( const char* sCase, const char* sPreset, unsigned uiSeed, const char* sOut )
{
    CaseFile c = readCase( sCase );
    RefIndex idx = buildIndex( c );
    ParameterSetManager xParams;
    selectPreset( xParams, sPreset );
    BinarySeeding xSeeding( xParams );
    StripOfConsideration xSoc( xParams );
    Harmonization xHarm( xParams );
    NeedlemanWunsch xDp( xParams );
    MappingQuality xMq( xParams );
    FILE* f = fopen( sOut, "w" );
    for( size_t i = 0; i < c.reads.size( ); i++ )
    {
        auto pQ = mkSeq( c.reads[ i ] );
        pQ->sName = "r" + std::to_string( i );
        fprintf( f, "R %zu %zu\n", i, c.reads[ i ].size( ) );
        auto pSegs = xSeeding.execute( idx.pFM, pQ );
        fprintf( f, "SEG %zu\n", pSegs->size( ) );
        for( auto& s : *pSegs )
            fprintf( f, "s %llu %llu %lld %lld %lld\n", (unsigned long long)s.start( ), (unsigned long long)s.size( ),
                     (long long)s.saInterval( ).start( ), (long long)s.saInterval( ).startRevComp( ),
                     (long long)s.saInterval( ).size( ) );
        {
            auto pSeeds = xSoc.xExtractHelper.execute( pSegs, idx.pFM, pQ, idx.pPack );
            fprintf( f, "SEED %zu\n", pSeeds->size( ) );
            dumpSeeds( f, "d", *pSeeds );
            // SoC pop order on a private queue
            auto pSocs = xSoc.xHelper.execute( pSeeds, pQ, idx.pPack );
            fprintf( f, "SOC %zu\n", pSocs->size( ) );
            while( !pSocs->empty( ) )
            {
                auto uiScore = std::get<0>( pSocs->vMaxima.front( ) ).uiAccumulativeLength;
                auto uiAmb = std::get<0>( pSocs->vMaxima.front( ) ).uiSeedAmbiguity;
                auto p = pSocs->pop( );
                fprintf( f, "c %u %llu %u %zu\n", p->xStats.index_of_strip, (unsigned long long)uiScore, uiAmb,
                         p->size( ) );
                dumpSeeds( f, "e", *p );
            }
        }
        auto pSocs = xSoc.execute( pSegs, pQ, idx.pPack, idx.pFM );
        srand( uiSeed );
        auto pHarm = xHarm.execute( pSocs, pQ, idx.pFM );
        fprintf( f, "HARM %zu\n", pHarm->size( ) );
        for( auto& pS : *pHarm )
        {
            fprintf( f, "h %u %zu\n", pS->xStats.index_of_strip, pS->size( ) );
            dumpSeeds( f, "g", *pS );
        }
        auto pAlns = xDp.execute( pHarm, pQ, idx.pPack );
        fprintf( f, "ALN %zu\n", pAlns->size( ) );
        for( auto& pA : *pAlns )
        {
            fprintf( f, "a %llu %llu %llu %llu %lld %u %zu", (unsigned long long)pA->uiBeginOnRef,
                     (unsigned long long)pA->uiEndOnRef, (unsigned long long)pA->uiBeginOnQuery,
                     (unsigned long long)pA->uiEndOnQuery, (long long)pA->iScore, pA->xStats.index_of_strip,
                     pA->data.size( ) );
            for( auto& d : pA->data )
                fprintf( f, " %d:%llu", (int)d.first, (unsigned long long)d.second );
            fprintf( f, "\n" );
        }
        auto pMq = xMq.execute( pQ, pAlns );
        fprintf( f, "MQ %zu\n", pMq->size( ) );
        for( auto& pA : *pMq )
            fprintf( f, "m %llu %llu %llu %llu %lld %d %d %.17g\n", (unsigned long long)pA->uiBeginOnRef,
                     (unsigned long long)pA->uiEndOnRef, (unsigned long long)pA->uiBeginOnQuery,
                     (unsigned long long)pA->uiEndOnQuery, (long long)pA->iScore, (int)pA->bSecondary,
                     (int)pA->bSupplementary, pA->fMappingQuality );
    }
    fclose( f );
    return 0;
}

// SAM records of the reference's FileWriter (fileWriter.cpp:11-158) for the reads of a case, written through its
// OutStream constructor (fileWriter.h:407-422); options: bit 0 = soft clip, bit 1 = =/X cigars instead of M
struct CaptureStream : public OutStream
{
    FILE* f;
    CaptureStream( FILE* f_ ) : f( f_ )
    {}
    OutStream& operator<<( std::string s )
    {
        fputs( s.c_str( ), f );
        return *this;
    }
};
static int cmdSam( const char* sCase, const char* sPreset, unsigned uiSeed, const char* sOut, int iOptions,
                   const char* sReadsFile )
{
    CaseFile c = readCase( sCase );
    RefIndex idx = buildIndex( c );
    ParameterSetManager xParams;
    selectPreset( xParams, sPreset );
    xParams.getSelected( )->xSoftClip->set( ( iOptions & 1 ) != 0 );
    xParams.getSelected( )->xOutputMCigar->set( ( iOptions & 2 ) == 0 );
    xParams.getSelected( )->xEmulateNgmlrTags->set( ( iOptions & 4 ) != 0 );
    BinarySeeding xSeeding( xParams );
    StripOfConsideration xSoc( xParams );
    Harmonization xHarm( xParams );
    NeedlemanWunsch xDp( xParams );
    MappingQuality xMq( xParams );
    FILE* f = fopen( sOut, "w" );
    auto pStream = std::make_shared<CaptureStream>( f );
    FileWriter xWriter( xParams, pStream, idx.pPack );
    // reads: the case's own, or (sReadsFile) whatever the reference's FileReader returns for a FASTA/FASTQ file
    std::vector<std::shared_ptr<NucSeq>> vReads;
    if( sReadsFile )
    {
        FileReader xReader( xParams );
        auto pIn = std::make_shared<StdFileStream>( fs::path( sReadsFile ) );
        while( auto pQ = xReader.execute( pIn ) )
            vReads.push_back( pQ );
    }
    else
        for( size_t i = 0; i < c.reads.size( ); i++ )
        {
            vReads.push_back( mkSeq( c.reads[ i ] ) );
            vReads.back( )->sName = "r" + std::to_string( i );
        }
    for( size_t i = 0; i < vReads.size( ); i++ )
    {
        auto pQ = vReads[ i ];
        auto pSegs = xSeeding.execute( idx.pFM, pQ );
        auto pSocs = xSoc.execute( pSegs, pQ, idx.pPack, idx.pFM );
        srand( uiSeed );
        auto pHarm = xHarm.execute( pSocs, pQ, idx.pFM );
        auto pAlns = xDp.execute( pHarm, pQ, idx.pPack );
        auto pMq = xMq.execute( pQ, pAlns );
        xWriter.execute( pQ, pMq, idx.pPack );
    }
    fclose( f );
    return 0;
}

// MappingQuality -> [SmallInversions] -> [PairedReads] -> (Paired)FileWriter, as wired in
// setUpCompGraph / setUpCompGraphPaired (export.cpp:72-202); reads 2k and 2k+1 of the case are the mates of pair k
static void dumpF4Line( FILE* f, const char* tag, const std::shared_ptr<Alignment>& pA, int iOther )
{
    fprintf( f, "%s %d %d %llu %llu %llu %llu %lld %u %d %d %.17g %zu", tag, (int)pA->xStats.bFirst, iOther,
             (unsigned long long)pA->uiBeginOnRef, (unsigned long long)pA->uiEndOnRef, (unsigned long long)pA->uiBeginOnQuery,
             (unsigned long long)pA->uiEndOnQuery, (long long)pA->iScore, pA->xStats.index_of_strip, (int)pA->bSecondary,
             (int)pA->bSupplementary, pA->fMappingQuality, pA->data.size( ) );
    for( auto& d : pA->data )
        fprintf( f, " %d:%llu", (int)d.first, (unsigned long long)d.second );
    fprintf( f, "\n" );
}
static int cmdF4( const char* sCase, const char* sPreset, unsigned uiSeed, const char* sOut, bool bInv, bool bPaired,
                  int iZDropInv, const char* sSam, int iOptions )
{
    CaseFile c = readCase( sCase );
    RefIndex idx = buildIndex( c );
    ParameterSetManager xParams;
    selectPreset( xParams, sPreset );
    xParams.getSelected( )->xSearchInversions->set( bInv );
    xParams.getSelected( )->xZDropInversion->set( iZDropInv );
    xParams.getSelected( )->xSoftClip->set( ( iOptions & 1 ) != 0 );
    xParams.getSelected( )->xOutputMCigar->set( ( iOptions & 2 ) == 0 );
    BinarySeeding xSeeding( xParams );
    StripOfConsideration xSoc( xParams );
    Harmonization xHarm( xParams );
    NeedlemanWunsch xDp( xParams );
    MappingQuality xMq( xParams );
    SmallInversions xInv( xParams );
    PairedReads xPair( xParams );
    FILE* f = fopen( sOut, "w" );
    FILE* fSam = sSam ? fopen( sSam, "w" ) : nullptr;
    std::shared_ptr<FileWriter> pWriter;
    std::shared_ptr<PairedFileWriter> pPairedWriter;
    if( fSam && bPaired )
        pPairedWriter = std::make_shared<PairedFileWriter>( xParams, std::make_shared<CaptureStream>( fSam ), idx.pPack );
    else if( fSam )
        pWriter = std::make_shared<FileWriter>( xParams, std::make_shared<CaptureStream>( fSam ), idx.pPack );
    auto fFinal = [ & ]( std::shared_ptr<NucSeq> pQ ) {
        auto pSegs = xSeeding.execute( idx.pFM, pQ );
        auto pSocs = xSoc.execute( pSegs, pQ, idx.pPack, idx.pFM );
        srand( uiSeed );
        auto pHarm = xHarm.execute( pSocs, pQ, idx.pFM );
        auto pAlns = xDp.execute( pHarm, pQ, idx.pPack );
        auto pMq = xMq.execute( pQ, pAlns );
        return bInv ? xInv.execute( pMq, pQ, idx.pPack ) : pMq;
    };
    std::vector<std::shared_ptr<NucSeq>> vReads;
    for( size_t i = 0; i < c.reads.size( ); i++ )
    {
        vReads.push_back( mkSeq( c.reads[ i ] ) );
        vReads.back( )->sName = "r" + std::to_string( i );
    }
    if( !bPaired )
        for( size_t i = 0; i < vReads.size( ); i++ )
        {
            auto pFin = fFinal( vReads[ i ] );
            fprintf( f, "R %zu %llu\nFIN 0 %zu\n", i, (unsigned long long)vReads[ i ]->length( ), pFin->size( ) );
            for( auto& pA : *pFin )
                dumpF4Line( f, "f", pA, -1 );
            if( pWriter )
                pWriter->execute( vReads[ i ], pFin, idx.pPack );
        }
    else
        for( size_t k = 0; 2 * k + 1 < vReads.size( ); k++ )
        {
            auto pQ1 = vReads[ 2 * k ], pQ2 = vReads[ 2 * k + 1 ];
            auto pFin1 = fFinal( pQ1 ), pFin2 = fFinal( pQ2 );
            auto pPair = xPair.execute( pQ1, pQ2, pFin1, pFin2, idx.pPack );
            fprintf( f, "P %zu %llu %llu\n", k, (unsigned long long)pQ1->length( ), (unsigned long long)pQ2->length( ) );
            fprintf( f, "FIN 0 %zu\n", pFin1->size( ) );
            for( auto& pA : *pFin1 )
                dumpF4Line( f, "f", pA, -1 );
            fprintf( f, "FIN 1 %zu\n", pFin2->size( ) );
            for( auto& pA : *pFin2 )
                dumpF4Line( f, "f", pA, -1 );
            fprintf( f, "PAIR %zu\n", pPair->size( ) );
            for( auto& pA : *pPair )
            {
                int iOther = -1;
                auto pO = pA->xStats.pOther.lock( );
                for( size_t j = 0; pO != nullptr && j < pPair->size( ); j++ )
                    if( ( *pPair )[ j ] == pO )
                        iOther = (int)j;
                dumpF4Line( f, "p", pA, iOther );
            }
            if( pPairedWriter )
                pPairedWriter->execute( pQ1, pQ2, pPair, idx.pPack );
        }
    fclose( f );
    if( fSam )
    {
        pWriter.reset( );
        pPairedWriter.reset( );
        fclose( fSam );
    }
    return 0;
}

// reads of a FASTA / FASTQ file as the reference's FileReader (fileReader.cpp:37-196) returns them
static int cmdRead( const char* sIn, const char* sOut )
{
    ParameterSetManager xParams;
    FileReader xReader( xParams );
    // .gz files go through the reference's GzFileStream like in FileStreamFromPath (fileReader.h:407-423)
    std::shared_ptr<FileStream> pStream;
    if( fs::path( sIn ).extension( ).string( ) == ".gz" )
        pStream = std::make_shared<GzFileStream>( fs::path( sIn ) );
    else
        pStream = std::make_shared<StdFileStream>( fs::path( sIn ) );
    FILE* f = fopen( sOut, "w" );
    while( true )
    {
        std::shared_ptr<NucSeq> pQ;
        try
        {
            pQ = xReader.execute( pStream );
        }
        catch( const std::runtime_error& e )
        {
            fprintf( f, "ERROR %s\n", e.what( ) );
            break;
        }
        if( pQ == nullptr )
            break;
        fprintf( f, "%s %llu ", pQ->sName.c_str( ), (unsigned long long)pQ->length( ) );
        for( size_t i = 0; i < pQ->length( ); i++ )
            fputc( '0' + ( *pQ )[ i ], f );
        fputc( '\n', f );
    }
    fclose( f );
    return 0;
}

// mate pairs as the reference's PairedFileReader (fileReader.h:568-617) returns them; bRevComp = "Paired Mate - Mate Pair"
static int cmdReadPair( const char* sIn1, const char* sIn2, const char* sOut, bool bRevComp )
{
    ParameterSetManager xParams;
    xParams.getSelected( )->xRevCompPairedReadMates->set( bRevComp );
    PairedFileReader xReader( xParams );
    auto pStream = std::make_shared<PairedFileStream>( std::make_shared<StdFileStream>( fs::path( sIn1 ) ),
                                                       std::make_shared<StdFileStream>( fs::path( sIn2 ) ) );
    FILE* f = fopen( sOut, "w" );
    while( true )
    {
        std::shared_ptr<PairedReadsContainer> pPair;
        try
        {
            pPair = xReader.execute( pStream );
        }
        catch( const std::runtime_error& e )
        {
            fprintf( f, "ERROR %s\n", e.what( ) );
            break;
        }
        if( pPair == nullptr )
            break;
        for( auto pQ : *pPair )
        {
            fprintf( f, "%s %llu ", pQ->sName.c_str( ), (unsigned long long)pQ->length( ) );
            for( size_t i = 0; i < pQ->length( ); i++ )
                fputc( '0' + ( *pQ )[ i ], f );
            fprintf( f, " %s\n", pQ->toQualString( ).c_str( ) );
        }
    }
    fclose( f );
    return 0;
}

// wall time of the reference's own modules (seeding .. mapping quality) over the reads of a case, T threads each with
// its own pass over a slice of the reads; prints reads/s (sanity check of the oracle's speed)
#include <chrono>
#include <thread>
static int cmdTime( const char* sCase, const char* sPreset, int iThreads )
{
    CaseFile c = readCase( sCase );
    RefIndex idx = buildIndex( c );
    ParameterSetManager xParams;
    selectPreset( xParams, sPreset );
    BinarySeeding xSeeding( xParams );
    StripOfConsideration xSoc( xParams );
    Harmonization xHarm( xParams );
    NeedlemanWunsch xDp( xParams );
    MappingQuality xMq( xParams );
    std::vector<std::shared_ptr<NucSeq>> vReads;
    for( auto& r : c.reads )
        vReads.push_back( mkSeq( r ) );
    std::vector<size_t> vAligned( iThreads, 0 );
    const auto t0 = std::chrono::steady_clock::now( );
    std::vector<std::thread> vT;
    for( int t = 0; t < iThreads; t++ )
        vT.emplace_back( [ &, t ]( ) {
            for( size_t i = t; i < vReads.size( ); i += iThreads )
            {
                auto pQ = vReads[ i ];
                auto pSegs = xSeeding.execute( idx.pFM, pQ );
                auto pSocs = xSoc.execute( pSegs, pQ, idx.pPack, idx.pFM );
                auto pHarm = xHarm.execute( pSocs, pQ, idx.pFM );
                auto pAlns = xDp.execute( pHarm, pQ, idx.pPack );
                auto pMq = xMq.execute( pQ, pAlns );
                if( !pMq->empty( ) )
                    vAligned[ t ]++;
            }
        } );
    for( auto& t : vT )
        t.join( );
    const double dt = std::chrono::duration<double>( std::chrono::steady_clock::now( ) - t0 ).count( );
    size_t n = 0;
    for( size_t a : vAligned )
        n += a;
    printf( "reference: %zu reads (%zu aligned) in %.3f s on %d threads = %.1f reads/s\n", vReads.size( ), n, dt, iThreads,
            vReads.size( ) / dt );
    return 0;
}

static int cmdExt( const char* sCase, const char* sOut )
{
    CaseFile c = readCase( sCase );
    RefIndex idx = buildIndex( c );
    FILE* f = fopen( sOut, "w" );
    for( size_t i = 0; i < c.reads.size( ); i++ )
    {
        auto& q = c.reads[ i ];
        fprintf( f, "R %zu %zu\n", i, q.size( ) );
        if( q.empty( ) || q.back( ) >= 4 )
            continue;
        SAInterval ik = idx.pFM->init_interval( q.back( ) );
        fprintf( f, "i %lld %lld %lld\n", (long long)ik.start( ), (long long)ik.startRevComp( ), (long long)ik.size( ) );
        for( size_t j = q.size( ) - 1; j-- > 0 && ik.size( ) > 0; )
        {
            for( uint8_t cc = 0; cc < 5; cc++ )
            {
                SAInterval ok = idx.pFM->extend_backward( ik, cc );
                fprintf( f, "x %d %lld %lld %lld\n", (int)cc, (long long)ok.start( ), (long long)ok.startRevComp( ),
                         (long long)ok.size( ) );
            }
            ik = idx.pFM->extend_backward( ik, q[ j ] );
        }
        // SA lookups over the final (small) interval
        if( ik.size( ) > 0 && ik.size( ) <= 64 )
            for( auto p = ik.start( ); p < ik.end( ); p++ )
                fprintf( f, "p %lld %lld\n", (long long)p, (long long)idx.pFM->bwt_sa( p ) );
    }
    fclose( f );
    return 0;
}

static int cmdKsw( const char* sCase, const char* sOut, bool bDirty, const int* aSc )
{
    std::vector<KswCase> v = readKswCases( sCase );
    FILE* f = fopen( sOut, "w" );
    KswCppParam<5> xP( aSc[ 0 ], aSc[ 1 ], aSc[ 2 ], aSc[ 3 ], aSc[ 4 ], aSc[ 5 ] ); // match, mismatch, gap, extend, gap2, extend2
    AlignedMemoryManager xShared;
    for( size_t i = 0; i < v.size( ); i++ )
    {
        auto& k = v[ i ];
        kswcpp_extz_t ez{ };
        if( bDirty )
            kswcpp_dispatch( (int)k.q.size( ), k.q.data( ), (int)k.t.size( ), k.t.data( ), xP, k.w, k.zdrop, k.flag, &ez,
                             xShared );
        else
        {
            AlignedMemoryManager xMem;
            kswcpp_dispatch( (int)k.q.size( ), k.q.data( ), (int)k.t.size( ), k.t.data( ), xP, k.w, k.zdrop, k.flag, &ez,
                             xMem );
        }
        fprintf( f, "k %zu %u %u %d %d %d %d %d %d %d %d %d", i, (unsigned)ez.max, (unsigned)ez.zdropped, ez.max_q,
                 ez.max_t, ez.mqe, ez.mqe_t, ez.mte, ez.mte_q, ez.score, ez.reach_end, ez.n_cigar );
        for( int j = 0; j < ez.n_cigar; j++ )
            fprintf( f, " %u", ez.cigar[ j ] );
        fprintf( f, "\n" );
        free( ez.cigar );
    }
    fclose( f );
    return 0;
}

int main( int argc, char** argv )
{
    if( argc >= 4 && !strcmp( argv[ 1 ], "index" ) )
        return cmdIndex( argv[ 2 ], argv[ 3 ] );
    if( argc >= 12 && !strcmp( argv[ 1 ], "pipe" ) ) // pipe <case> <preset> <seed> <out> match mismatch gap extend gap2 extend2
    {
        pGlobalParams->iMatch->set( atoi( argv[ 6 ] ) );
        pGlobalParams->iMissMatch->set( atoi( argv[ 7 ] ) );
        pGlobalParams->iGap->set( atoi( argv[ 8 ] ) );
        pGlobalParams->iExtend->set( atoi( argv[ 9 ] ) );
        pGlobalParams->iGap2->set( atoi( argv[ 10 ] ) );
        pGlobalParams->iExtend2->set( atoi( argv[ 11 ] ) );
    }
    if( argc >= 6 && !strcmp( argv[ 1 ], "pipe" ) )
        return cmdPipe( argv[ 2 ], argv[ 3 ], (unsigned)atoi( argv[ 4 ] ), argv[ 5 ] );
    if( argc >= 6 && !strcmp( argv[ 1 ], "sam" ) )
        return cmdSam( argv[ 2 ], argv[ 3 ], (unsigned)atoi( argv[ 4 ] ), argv[ 5 ], argc >= 7 ? atoi( argv[ 6 ] ) : 0,
                       argc >= 8 ? argv[ 7 ] : nullptr );
    if( argc >= 9 && !strcmp( argv[ 1 ], "f4" ) )
        return cmdF4( argv[ 2 ], argv[ 3 ], (unsigned)atoi( argv[ 4 ] ), argv[ 5 ], atoi( argv[ 6 ] ) != 0, atoi( argv[ 7 ] ) != 0,
                      atoi( argv[ 8 ] ), argc >= 10 ? argv[ 9 ] : nullptr, argc >= 11 ? atoi( argv[ 10 ] ) : 0 );
    if( argc >= 6 && !strcmp( argv[ 1 ], "readpair" ) )
        return cmdReadPair( argv[ 2 ], argv[ 3 ], argv[ 4 ], atoi( argv[ 5 ] ) != 0 );
    if( argc >= 7 && !strcmp( argv[ 1 ], "pipeidx" ) ) // pipeidx <prefix> <case> <preset> <seed> <out>
    {
        g_sIndexPrefix = argv[ 2 ];
        return cmdPipe( argv[ 3 ], argv[ 4 ], (unsigned)atoi( argv[ 5 ] ), argv[ 6 ] );
    }
    if( argc >= 6 && !strcmp( argv[ 1 ], "timeidx" ) ) // timeidx <prefix> <case (reads only)> <preset> <threads>
    {
        g_sIndexPrefix = argv[ 2 ];
        return cmdTime( argv[ 3 ], argv[ 4 ], atoi( argv[ 5 ] ) );
    }
    if( argc >= 4 && !strcmp( argv[ 1 ], "fastapack" ) ) // fastapack <genome.fa> <prefix>: Pack::vAppendFASTA + vStoreCollection
    {
        Pack xPack;
        xPack.vAppendFASTA( argv[ 2 ] );
        xPack.vStoreCollection( argv[ 3 ] );
        return 0;
    }
    if( argc >= 5 && !strcmp( argv[ 1 ], "time" ) )
        return cmdTime( argv[ 2 ], argv[ 3 ], atoi( argv[ 4 ] ) );
    if( argc >= 4 && !strcmp( argv[ 1 ], "read" ) )
        return cmdRead( argv[ 2 ], argv[ 3 ] );
    if( argc >= 4 && !strcmp( argv[ 1 ], "ext" ) )
        return cmdExt( argv[ 2 ], argv[ 3 ] );
    if( argc >= 4 && !strcmp( argv[ 1 ], "ksw" ) )
    {
        // ksw <cases> <out> [dirty|clean [match mismatch gap extend gap2 extend2]]
        int aSc[ 6 ] = { 2, 4, 4, 2, 24, 1 };
        if( argc >= 11 )
            for( int i = 0; i < 6; i++ )
                aSc[ i ] = atoi( argv[ 5 + i ] );
        return cmdKsw( argv[ 2 ], argv[ 3 ], argc >= 5 && !strcmp( argv[ 4 ], "dirty" ), aSc );
    }
    fprintf( stderr, "usage: ref_dump index|pipe|sam|f4|read|ext|ksw ...\n" );
    return 2;
}
