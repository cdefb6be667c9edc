// CPU-only check of the paired part of the host layer (ma_amd/host/ma_modules.h PairedReads, ma_sam.h PairedFileWriter /
// FileWriter): the per-mate alignment lists are taken from an f4 dump of the compiled reference (tests/golden/f4.*.f4.gz:
// "f" records = lists after MappingQuality [+ SmallInversions]), PairedReads picks the pair, the writer prints SAM; both
// must equal what the reference produced ("p" records, *.sam.gz).  SmallInversions itself needs the GPU (its DP runs
// through ma_ksw_batch) and is covered by tests/emul/f4_graph_test.cpp under -m gpu.
// usage: f4_test <case> <f4 dump> <preset> <paired 0|1> <sam options> <out.f4> <out.sam>
#include "../../oracle/dump_format.h"
#include "ma_sam.h"

#include <cstdio>
#include <sstream>

using namespace libMA;
typedef libMS::ContainerVector<std::shared_ptr<Alignment>> AlnVec;

static void dumpLine( FILE* f, const char* tag, const Alignment& a, int iOther )
{
    fprintf( f, "%s %d %d %llu %llu %llu %llu %lld %u %d %d %.17g %zu", tag, (int)a.xStats.bFirst, iOther,
             (unsigned long long)a.uiBeginOnRef, (unsigned long long)a.uiEndOnRef, (unsigned long long)a.uiBeginOnQuery,
             (unsigned long long)a.uiEndOnQuery, (long long)a.iScore, a.index_of_strip, (int)a.bSecondary, (int)a.bSupplementary,
             a.fMappingQuality, a.data.size( ) );
    for( auto& d : a.data )
        fprintf( f, " %d:%llu", (int)d.first, (unsigned long long)d.second );
    fprintf( f, "\n" );
}

int main( int argc, char** argv )
{
    if( argc < 8 )
        return 2;
    CaseFile c = readCase( argv[ 1 ] );
    const bool bPaired = atoi( argv[ 4 ] ) != 0;
    const int iOptions = atoi( argv[ 5 ] );
    auto pPack = std::make_shared<Pack>( );
    uint64_t off = 0;
    for( size_t i = 0; i < c.contigs.size( ); i++ )
    {
        pPack->vNames.push_back( c.names[ i ] );
        pPack->vStarts.push_back( off );
        pPack->vLengths.push_back( c.contigs[ i ].size( ) );
        off += c.contigs[ i ].size( );
    }
    ParameterSetManager xParams;
    xParams.setSelected( argv[ 3 ] );
    xParams.xSam.bSoftClip = ( iOptions & 1 ) != 0;
    xParams.xSam.bOutputMCigar = ( iOptions & 2 ) == 0;
    auto pStream = std::make_shared<StringOutStream>( );
    std::shared_ptr<FileWriter> pWriter;
    std::shared_ptr<PairedFileWriter> pPairedWriter;
    if( bPaired )
        pPairedWriter = std::make_shared<PairedFileWriter>( xParams, std::static_pointer_cast<OutStream>( pStream ), pPack );
    else
        pWriter = std::make_shared<FileWriter>( xParams, std::static_pointer_cast<OutStream>( pStream ), pPack );
    PairedReads xPair( xParams );
    auto mkRead = [ & ]( size_t i ) {
        auto p = std::make_shared<NucSeq>( );
        p->xCodes = c.reads[ i ];
        p->sName = "r" + std::to_string( i );
        return p;
    };
    // the dump was written AFTER the reference's PairedReads ran, i.e. the lists already carry the changes of its pick
    // (flags, mapq); the pick and its mapq depend on scores, positions and list sizes only, so running PairedReads on
    // these lists must reproduce the "p" records
    std::ifstream f( argv[ 2 ] );
    FILE* fo = fopen( argv[ 6 ], "w" );
    std::string line;
    std::shared_ptr<AlnVec> fin[ 2 ] = { std::make_shared<AlnVec>( ), std::make_shared<AlnVec>( ) };
    long unit = -1;
    unsigned long long l1 = 0, l2 = 0;
    int cur = 0;
    auto flush = [ & ]( ) {
        if( unit < 0 )
            return;
        if( !bPaired )
        {
            auto pQ = mkRead( (size_t)unit );
            fprintf( fo, "R %ld %llu\nFIN 0 %zu\n", unit, l1, fin[ 0 ]->size( ) );
            for( auto& a : *fin[ 0 ] )
                dumpLine( fo, "f", *a, -1 );
            pWriter->execute( pQ, fin[ 0 ], pPack );
        }
        else
        {
            auto pQ1 = mkRead( 2 * (size_t)unit ), pQ2 = mkRead( 2 * (size_t)unit + 1 );
            auto pPair = xPair.execute( pQ1, pQ2, fin[ 0 ], fin[ 1 ], pPack );
            fprintf( fo, "P %ld %llu %llu\n", unit, l1, l2 );
            for( int m = 0; m < 2; m++ )
            {
                fprintf( fo, "FIN %d %zu\n", m, fin[ m ]->size( ) );
                for( auto& a : *fin[ m ] )
                    dumpLine( fo, "f", *a, -1 );
            }
            fprintf( fo, "PAIR %zu\n", pPair->size( ) );
            for( auto& a : *pPair )
            {
                int iOther = -1;
                auto pO = a->xStats.pOther.lock( );
                for( size_t j = 0; pO != nullptr && j < pPair->size( ); j++ )
                    if( ( *pPair )[ j ] == pO )
                        iOther = (int)j;
                dumpLine( fo, "p", *a, iOther );
            }
            pPairedWriter->execute( pQ1, pQ2, pPair, pPack );
        }
        fin[ 0 ] = std::make_shared<AlnVec>( );
        fin[ 1 ] = std::make_shared<AlnVec>( );
    };
    while( std::getline( f, line ) )
    {
        std::istringstream ss( line );
        std::string tag;
        ss >> tag;
        if( tag == "R" || tag == "P" )
        {
            flush( );
            ss >> unit >> l1 >> l2;
        }
        else if( tag == "FIN" )
            ss >> cur;
        else if( tag == "f" )
        {
            auto a = std::make_shared<Alignment>( );
            int first, other, sec, supp;
            size_t nops;
            ss >> first >> other >> a->uiBeginOnRef >> a->uiEndOnRef >> a->uiBeginOnQuery >> a->uiEndOnQuery >> a->iScore >>
                a->index_of_strip >> sec >> supp;
            std::string sMq;
            ss >> sMq >> nops;
            a->fMappingQuality = sMq == "nan" ? NAN : strtod( sMq.c_str( ), nullptr );
            a->bSecondary = sec != 0;
            a->bSupplementary = supp != 0;
            for( size_t k = 0; k < nops; k++ )
            {
                std::string op;
                ss >> op;
                const size_t colon = op.find( ':' );
                a->data.emplace_back( (MatchType)atoi( op.substr( 0, colon ).c_str( ) ), strtoull( op.c_str( ) + colon + 1, nullptr, 10 ) );
            }
            fin[ cur ]->push_back( a );
        }
    }
    flush( );
    fclose( fo );
    FILE* fs = fopen( argv[ 7 ], "w" );
    fputs( pStream->sText.c_str( ), fs );
    fclose( fs );
    return 0;
}
