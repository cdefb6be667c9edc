// CPU-only check of the SAM writer of the host layer (ma_amd/host/ma_sam.h): alignments are taken from a pipeline
// dump (golden of the compiled reference, MQ + ALN records), the reads and contigs from the case file, and the SAM
// text must equal what the reference's FileWriter printed for the same reads (tests/golden/*.sam.gz).
// usage: sam_test <case> <pipe dump> <out.sam> <options: bit0 soft clip, bit1 =/X cigar, bit2 NGMLR tags> [reads.fastq]
// with a FASTA/FASTQ file the reads (names, qualities) come from the host layer's FileReader; they must be the first
// reads of the case in order, because the alignments are looked up in the dump by read index
#include "../../oracle/dump_format.h"
#include "ma_sam.h"

#include <cstdio>
#include <map>
#include <sstream>

using namespace libMA;

int main( int argc, char** argv )
{
    if( argc < 5 )
        return 2;
    CaseFile c = readCase( argv[ 1 ] );
    const int iOptions = atoi( argv[ 4 ] );
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
    xParams.xSam.bSoftClip = ( iOptions & 1 ) != 0;
    xParams.xSam.bOutputMCigar = ( iOptions & 2 ) == 0;
    xParams.xSam.bEmulateNgmlrTags = ( iOptions & 4 ) != 0; // needs reference bases: the contigs, packed on the host
    if( iOptions & 4 )
    {
        pPack->vPacHost.assign( ( off + 3 ) / 4 + 1, 0 );
        uint64_t p = 0;
        for( auto& rContig : c.contigs )
            for( uint8_t b : rContig )
            {
                pPack->vPacHost[ p >> 2 ] |= (uint8_t)( ( b & 3 ) << ( ( ~p & 3 ) << 1 ) );
                p++;
            }
    }
    auto pStream = std::make_shared<StringOutStream>( );
    FileWriter xWriter( xParams, std::static_pointer_cast<OutStream>( pStream ), pPack );
    if( iOptions & 8 ) // per-thread buffering of the writer: the same bytes once flush( ) has run
        xWriter.uiBufferBytes = 4096;
    std::vector<std::shared_ptr<NucSeq>> vFileReads;
    if( argc >= 6 )
    {
        FileReader xReader( xParams );
        auto pIn = std::make_shared<StdFileStream>( argv[ 5 ] );
        while( auto pQ = xReader.execute( pIn ) )
            vFileReads.push_back( pQ );
    }
    // parse the dump: per read the "a" records (with ops) and the "m" records (MappingQuality order + flags)
    std::ifstream f( argv[ 2 ] );
    std::string line;
    struct Rec
    {
        unsigned long long br, er, bq, eq;
        long long score;
        std::vector<std::pair<MatchType, nucSeqIndex>> ops;
    };
    std::vector<Rec> alns;
    std::vector<std::shared_ptr<Alignment>> mq;
    long read = -1;
    auto flush = [ & ]( ) {
        if( read < 0 )
            return;
        if( argc >= 6 && (size_t)read >= vFileReads.size( ) )
            return;
        auto pQ = std::make_shared<NucSeq>( );
        pQ->xCodes = c.reads[ (size_t)read ];
        pQ->sName = "r" + std::to_string( read );
        if( argc >= 6 )
        {
            if( vFileReads[ (size_t)read ]->xCodes != pQ->xCodes )
                throw std::runtime_error( "reads file does not match the case" );
            pQ = vFileReads[ (size_t)read ];
        }
        auto pV = std::make_shared<libMS::ContainerVector<std::shared_ptr<Alignment>>>( );
        for( auto& p : mq )
            pV->push_back( p );
        xWriter.execute( pQ, pV, pPack );
    };
    while( std::getline( f, line ) )
    {
        std::istringstream is( line );
        std::string tag;
        is >> tag;
        if( tag == "R" )
        {
            flush( );
            is >> read;
            alns.clear( );
            mq.clear( );
        }
        else if( tag == "a" )
        {
            Rec r;
            unsigned soc;
            size_t n;
            is >> r.br >> r.er >> r.bq >> r.eq >> r.score >> soc >> n;
            for( size_t k = 0; k < n; k++ )
            {
                std::string t;
                is >> t;
                const size_t colon = t.find( ':' );
                r.ops.emplace_back( (MatchType)atoi( t.substr( 0, colon ).c_str( ) ),
                                    (nucSeqIndex)strtoull( t.substr( colon + 1 ).c_str( ), nullptr, 10 ) );
            }
            alns.push_back( r );
        }
        else if( tag == "m" )
        {
            unsigned long long br, er, bq, eq;
            long long score;
            int sec, sup;
            double q;
            is >> br >> er >> bq >> eq >> score >> sec >> sup >> q;
            auto pA = std::make_shared<Alignment>( );
            pA->uiBeginOnRef = br, pA->uiEndOnRef = er, pA->uiBeginOnQuery = bq, pA->uiEndOnQuery = eq;
            pA->iScore = score, pA->bSecondary = sec != 0, pA->bSupplementary = sup != 0, pA->fMappingQuality = q;
            for( auto& r : alns ) // the MQ record is one of the NW alignments
                if( r.br == br && r.er == er && r.bq == bq && r.eq == eq && r.score == score )
                {
                    pA->data = r.ops;
                    break;
                }
            mq.push_back( pA );
        }
    }
    flush( );
    xWriter.flush( );
    FILE* o = fopen( argv[ 3 ], "w" );
    fputs( pStream->sText.c_str( ), o );
    fclose( o );
    return 0;
}
