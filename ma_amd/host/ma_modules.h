// ma_modules.h -- drop-in MI355X graph nodes for MA's seed-and-extend path.  Same class names, template
// signatures, constructor argument and error behaviour (std::runtime_error) as the reference modules:
//   BinarySeeding        : Module<SegmentVector,false,SuffixArrayInterface,NucSeq>           (binarySeeding.h:26)
//   StripOfConsideration : Module<SoCPriorityQueue,false,SegmentVector,NucSeq,Pack,FMIndex>  (stripOfConsideration.h:164)
//   Harmonization        : Module<ContainerVector<shared_ptr<Seeds>>,false,SoCPriorityQueue,NucSeq,FMIndex>
//                                                                                            (harmonization.h:34-35)
//   NeedlemanWunsch      : Module<ContainerVector<shared_ptr<Alignment>>,false,ContainerVector<shared_ptr<Seeds>>,
//                                 NucSeq,Pack>                                               (needlemanWunsch.h:51-52)
//   MappingQuality       : Module<ContainerVector<shared_ptr<Alignment>>,false,NucSeq,ContainerVector<...>>
//                                                                                            (mappingQuality.h:22-23)
// so that libMA::setUpCompGraph (libs/ma/src/util/export.cpp:104-108) builds unchanged.  All compute
// goes through the C ABI in include/ma_amd.h; a non-zero status becomes std::runtime_error (the
// convention of module.h:339-377).  The per-read execute() runs a batch of one read on the GPU -- correct
// but latency-bound; BatchAligner below is the throughput API (one ma_batch per call, 10^5..10^6 reads).
#pragma once
#include "../../include/ma_amd.h"
#include "ms_graph.h"
#include <cmath>
#include <limits>
#include <tuple>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace libMA
{
typedef uint64_t nucSeqIndex;
typedef int64_t t_bwtIndex;

inline void maCheck( int rc )
{
    if( rc != 0 )
        throw std::runtime_error( ma_last_error( ) );
}

// SAM options of the selected parameter set (parameter.h:557-564, defaults 726-754); read by FileWriter (ma_sam.h)
struct SamOptions
{
    bool bNoSecondary = false; // "Omit Secondary Alignments"
    bool bNoSupplementary = false; // "Omit Supplementary Alignments"
    bool bEmulateNgmlrTags = false; // "Emulate NGMLR's tag output"
    bool bOutputMCigar = true; // "Use M in CIGAR"
    bool bCGTag = true; // "Output long cigars in CG tag"
    bool bSoftClip = false; // "Soft clip"
};

// ParameterSetManager (parameter.h:1067-1201) reduced to the preset selection the path reads
class ParameterSetManager
{
  public:
    ma_params xSelected;
    SamOptions xSam;
    ParameterSetManager( )
    {
        ma_params_default( &xSelected );
    }
    void setSelected( const std::string& sKey )
    {
        std::string k;
        for( char c : sKey )
            k += (char)tolower( c );
        if( k == "default" )
            ma_params_default( &xSelected );
        else if( k == "illumina" )
            ma_params_illumina( &xSelected );
        else if( k == "illuminapaired" ) // parameter.h:1089-1094
        {
            ma_params_illumina( &xSelected );
            xSelected.use_paired_reads = 1;
        }
        else if( k == "pacbio" || k == "nanopore" ) // parameter.h:1096-1104
        {
            ma_params_default( &xSelected );
            xSelected.max_supplementary = 100;
            xSelected.min_num_soc = 5;
            if( k == "nanopore" )
                xSelected.seeding_technique = 1;
        }
        else
            throw std::runtime_error( "The presetting '" + sKey + "' can not be found." );
    }
    const ma_params* getSelected( ) const
    {
        return &xSelected;
    }
    ma_params* getSelected( )
    {
        return &xSelected;
    }
    bool bRevCompPairedReadMates = true; // "Paired Mate - Mate Pair" (parameter.h:785-788), read by PairedFileReader
};

class NucSeq : public libMS::Container // nucSeq.h:61-160: codes A0 C1 G2 T3 N4
{
  public:
    std::vector<uint8_t> xCodes;
    std::vector<uint8_t> xQuality; // FASTQ quality characters (empty = none, nucSeq.h:105-108)
    std::string sName = "unknown";
    NucSeq( )
    {}
    NucSeq( const std::string& sText )
    {
        for( char c : sText ) // nucSeq.cpp:17-28
            xCodes.push_back( c == 'A' || c == 'a' ? 0 : c == 'C' || c == 'c' ? 1 : c == 'G' || c == 'g' ? 2
                                                                              : c == 'T' || c == 't' ? 3 : 4 );
    }
    nucSeqIndex length( ) const
    {
        return xCodes.size( );
    }
};

// One device-resident index shared by the Pack and FMIndex views (both are read-only after load).
struct DeviceIndex
{
    ma_index* p = nullptr;
    ~DeviceIndex( )
    {
        if( p )
            ma_index_destroy( p );
    }
};

// SAInterval (fMIndex.h:44-150): bi-interval (start, start of the reverse complement's interval, size)
class SAInterval
{
  public:
    int64_t iStart = 0, iStartRevComp = 0, iSize = 0;
    SAInterval( )
    {}
    SAInterval( int64_t iStart, int64_t iStartRevComp, int64_t iSize ) : iStart( iStart ), iStartRevComp( iStartRevComp ), iSize( iSize )
    {}
    int64_t start( ) const
    {
        return iStart;
    }
    int64_t startRevComp( ) const
    {
        return iStartRevComp;
    }
    int64_t size( ) const
    {
        return iSize;
    }
    int64_t end( ) const
    {
        return iStart + iSize;
    }
    SAInterval revComp( ) const // fMIndex.h:85-88
    {
        return SAInterval( iStartRevComp, iStart, iSize );
    }
};
// SuffixArrayInterface (fMIndex.h:155-175): the fine-grained seam of the seeding stage.  One call = one tiny GPU launch;
// it is there for code written against the interface (and for tests), the modules use the batch entry points.
class SuffixArrayInterface : public libMS::Container
{
  public:
    std::shared_ptr<DeviceIndex> pDev;
    virtual SAInterval extend_backward( const SAInterval& ik, const uint8_t c ) // fMIndex.cpp:21-101
    {
        const int64_t in[ 3 ] = { ik.iStart, ik.iStartRevComp, ik.iSize };
        int64_t out[ 3 ];
        maCheck( ma_extend_backward_batch( pDev->p, in, &c, 1, out ) );
        return SAInterval( out[ 0 ], out[ 1 ], out[ 2 ] );
    }
    virtual SAInterval init_interval( uint8_t c ) // fMIndex.h:768-775
    {
        uint64_t L2[ 5 ];
        maCheck( ma_index_download( pDev->p, nullptr, nullptr, L2, nullptr, nullptr, nullptr, nullptr ) );
        if( c >= 4 )
            return SAInterval( 0, 0, 0 );
        return SAInterval( (int64_t)L2[ c ] + 1, (int64_t)L2[ 3 - c ] + 1, (int64_t)( L2[ c + 1 ] - L2[ c ] ) );
    }
    virtual uint64_t getRefSeqLength( ) const
    {
        uint64_t n = 0;
        maCheck( ma_index_sizes( pDev->p, nullptr, nullptr, &n, nullptr ) );
        return n;
    }
    virtual ~SuffixArrayInterface( )
    {}
};
class FMIndex : public SuffixArrayInterface
{
  public:
    int64_t bwt_sa( int64_t iRow ) // fMIndex.h:788-814: text position of a suffix array row
    {
        int64_t iPos = 0;
        maCheck( ma_bwt_sa_batch( pDev->p, &iRow, 1, &iPos ) );
        return iPos;
    }
};
class Pack : public libMS::Container
{
  public:
    std::shared_ptr<DeviceIndex> pDev;
    std::vector<std::string> vNames;
    std::vector<uint64_t> vStarts, vLengths;
    // runs of N of the input that were replaced by random bases (pack.h:630-666): offset, length; written to .amb
    std::vector<std::pair<uint64_t, uint64_t>> vHoles;
    std::vector<uint64_t> vNumHoles; // per contig, for .ann
    uint64_t uiUnpackedSizeForwardPlusReverse( ) const // pack.h:881-884: twice the forward strand
    {
        if( !vStarts.empty( ) )
            return 2 * ( vStarts.back( ) + vLengths.back( ) );
        uint64_t n = 0;
        maCheck( ma_index_sizes( pDev->p, nullptr, nullptr, &n, nullptr ) );
        return n;
    }
};

// Loads the reference's own index files <prefix>.bwt/.sa/.pac/.ann (fMIndex.h:555-663, pack.h:271-470)
// and uploads them; replaces `makePledge<Pack>(prefix)` / `makePledge<FMIndex>(prefix)` of
// execution-context.h:60-93.
inline void loadIndex( const std::string& sPrefix, std::shared_ptr<Pack>& pPack, std::shared_ptr<FMIndex>& pFM )
{
    auto slurp = []( const std::string& p ) {
        std::ifstream f( p, std::ios::binary );
        if( f.fail( ) )
            throw std::runtime_error( "File opening error: " + p );
        return std::vector<char>( ( std::istreambuf_iterator<char>( f ) ), std::istreambuf_iterator<char>( ) );
    };
    std::vector<char> bwt = slurp( sPrefix + ".bwt" ), sa = slurp( sPrefix + ".sa" ), pac = slurp( sPrefix + ".pac" );
    if( bwt.size( ) < 40 || sa.size( ) < 52 )
        throw std::runtime_error( "Unexpected fail after reading BWT from stream. " );
    int64_t primary, saPrimary;
    uint64_t L2[ 5 ] = { 0, 0, 0, 0, 0 };
    memcpy( &primary, bwt.data( ), 8 );
    memcpy( &L2[ 1 ], bwt.data( ) + 8, 32 );
    memcpy( &saPrimary, sa.data( ), 8 );
    if( primary != saPrimary )
        throw std::runtime_error( "BWT and suffix array have different primary." );
    uint64_t seqLen;
    memcpy( &seqLen, sa.data( ) + 44, 8 );
    if( seqLen != L2[ 4 ] )
        throw std::runtime_error( "SA-BWT inconsistency: suffix array has non matching sequence length stored." );
    // the files are not trusted: the sampling interval is read (the device code assumes 32, the value every index the
    // reference writes has, fMIndex.h:391), every array length is checked against the sequence length before a byte is
    // copied or uploaded (vRestoreBWT / vRestoreSuffixArray, fMIndex.h:555-663, fail the same way on short files)
    int32_t iSaIntv = 0;
    memcpy( &iSaIntv, sa.data( ) + 40, 4 );
    if( iSaIntv != 32 )
        throw std::runtime_error( "Suffix array sampling interval " + std::to_string( iSaIntv ) +
                                  " is not supported (the MI355X index expects 32)." );
    const uint64_t nWords = ( bwt.size( ) - 40 ) / 4, nSa = ( seqLen + 32 ) / 32;
    // occ-injected BWT: ceil(n / 16) symbol words + 8 counter words per 128-nt block incl. the final one (fMIndex.cpp:204-264)
    if( ( bwt.size( ) - 40 ) % 4 != 0 || nWords != ( seqLen + 15 ) / 16 + ( ( seqLen + 127 ) / 128 + 1 ) * 8 )
        throw std::runtime_error( "Unexpected fail after reading BWT from stream. " );
    if( sa.size( ) < 52 + ( nSa - 1 ) * 8 )
        throw std::runtime_error( "Unexpected bad after reading suffix array from stream. " );
    if( sa.size( ) != 52 + ( nSa - 1 ) * 8 )
        throw std::runtime_error( "Reading suffix array from file system failed due to non matching expected size." );
    if( seqLen % 2 != 0 || pac.size( ) < ( seqLen / 2 + 3 ) / 4 + 1 )
        throw std::runtime_error( "Pack and FM index do not match: unexpected size of " + sPrefix + ".pac" );
    std::vector<int64_t> vSa( nSa );
    vSa[ 0 ] = -1;
    memcpy( &vSa[ 1 ], sa.data( ) + 52, ( nSa - 1 ) * 8 );
    // .ann: "<fwd size> <n contigs> <seed>\n" then per contig "<gi> <name> <comment>\n<offset> <len> <n holes>\n"
    pPack = std::make_shared<Pack>( );
    {
        std::ifstream f( sPrefix + ".ann" );
        if( f.fail( ) )
            throw std::runtime_error( "File opening error: " + sPrefix + ".ann" );
        std::string line;
        std::getline( f, line );
        std::istringstream h( line );
        uint64_t uiFwd, nSeq;
        h >> uiFwd >> nSeq;
        for( uint64_t i = 0; i < nSeq; i++ )
        {
            std::getline( f, line );
            std::istringstream a( line );
            std::string gi, name;
            a >> gi >> name;
            std::getline( f, line );
            std::istringstream b( line );
            uint64_t off, len;
            b >> off >> len;
            pPack->vNames.push_back( name );
            pPack->vStarts.push_back( off );
            pPack->vLengths.push_back( len );
        }
    }
    auto pDev = std::make_shared<DeviceIndex>( );
    maCheck( ma_index_create( (const uint32_t*)( bwt.data( ) + 40 ), nWords, vSa.data( ), nSa, L2, primary, seqLen,
                              (const uint8_t*)pac.data( ), (int32_t)pPack->vStarts.size( ), pPack->vStarts.data( ),
                              pPack->vLengths.data( ), &pDev->p ) );
    pPack->pDev = pDev;
    pFM = std::make_shared<FMIndex>( );
    pFM->pDev = pDev;
}

// Builds the index on the GPU from N-free contigs (codes 0..3).
inline void buildIndex( const std::vector<std::shared_ptr<NucSeq>>& vContigs, std::shared_ptr<Pack>& pPack,
                        std::shared_ptr<FMIndex>& pFM )
{
    std::vector<uint64_t> lens;
    std::vector<uint8_t> cat;
    pPack = std::make_shared<Pack>( );
    uint64_t off = 0;
    for( auto& c : vContigs )
    {
        if( c->length( ) == 0 ) // Pack::vAppendSequence skips empty sequences (pack.h:596-600)
            continue;
        lens.push_back( c->length( ) );
        // Pack::vAppendSequence (pack.h:625-666): every N (code >= 4) becomes rand() & 3 (the caller seeds libc's rand();
        // the reference seeds it with the time, so N regions differ from run to run there), runs of N are recorded
        uint64_t uiHoles = 0;
        unsigned int uiPrev = 0;
        for( size_t i = 0; i < c->xCodes.size( ); i++ )
        {
            unsigned int uiCode = c->xCodes[ i ];
            if( uiCode >= 4 )
            {
                if( uiPrev == uiCode )
                    pPack->vHoles.back( ).second++;
                else
                {
                    pPack->vHoles.emplace_back( off + i, 1 );
                    uiHoles++;
                }
            }
            uiPrev = uiCode;
            cat.push_back( uiCode >= 4 ? (uint8_t)( rand( ) & 3 ) : (uint8_t)uiCode );
        }
        pPack->vNumHoles.push_back( uiHoles );
        pPack->vNames.push_back( c->sName );
        pPack->vStarts.push_back( off );
        pPack->vLengths.push_back( c->length( ) );
        off += c->length( );
    }
    auto pDev = std::make_shared<DeviceIndex>( );
    maCheck( ma_index_build( (int32_t)lens.size( ), lens.data( ), cat.data( ), &pDev->p ) );
    pPack->pDev = pDev;
    pFM = std::make_shared<FMIndex>( );
    pFM->pDev = pDev;
}

// Writes the device-resident index as the reference's own files <prefix>.bwt/.sa/.pac/.ann/.amb, i.e. what
// FMIndex::vStoreFMIndex (fMIndex.h:515-549) and Pack::vStoreCollection (pack.h:230-269, 725-770) write for the same
// genome: the reference (maCMD -x, FMIndex(prefix), Pack::vLoadCollection) loads an index built on the GPU.  The .bwt,
// .sa and .pac bytes equal the reference's; .ann carries the reference's time-based seed, here 0.
// With a genome title also writes <folder of prefix>/<title>.json, the file `maCMD -x` takes (GenomeManager::
// createGenomeJSON / loadGenome, execution-context.h:60-136; nlohmann layout: keys sorted, 4 blanks).
inline void storeIndex( const std::string& sPrefix, const std::shared_ptr<Pack>& pPack, const std::shared_ptr<FMIndex>& pFM,
                        const std::string& sGenomeTitle = "" )
{
    if( !sGenomeTitle.empty( ) )
    {
        const size_t uiSlash = sPrefix.find_last_of( '/' );
        const std::string sFolder = uiSlash == std::string::npos ? "" : sPrefix.substr( 0, uiSlash + 1 );
        const std::string sStem = uiSlash == std::string::npos ? sPrefix : sPrefix.substr( uiSlash + 1 );
        std::ofstream f( sFolder + sGenomeTitle + ".json" );
        f << "{\n    \"name\": \"" << sGenomeTitle << "\",\n    \"prefix\": \"" << sStem << "\",\n    \"type\": \"MA Genome\",\n"
          << "    \"version\": {\n        \"major\": 1,\n        \"minor\": 0\n    }\n}\n";
    }
    uint64_t nWords = 0, nSa = 0, uiN = 0;
    int32_t nContigs = 0;
    maCheck( ma_index_sizes( pFM->pDev->p, &nWords, &nSa, &uiN, &nContigs ) );
    const uint64_t uiF = uiN / 2;
    std::vector<uint32_t> vBwt( nWords );
    std::vector<int64_t> vSa( nSa );
    std::vector<uint8_t> vPac( ( uiF + 3 ) / 4 );
    std::vector<uint64_t> vStarts( nContigs ), vLens( nContigs );
    uint64_t L2[ 5 ];
    int64_t primary = 0;
    maCheck( ma_index_download( pFM->pDev->p, vBwt.data( ), vSa.data( ), L2, &primary, vPac.data( ), vStarts.data( ), vLens.data( ) ) );
    auto open = []( const std::string& p ) {
        FILE* f = fopen( p.c_str( ), "wb" );
        if( !f )
            throw std::runtime_error( "File opening error: " + p );
        return f;
    };
    auto put = []( FILE* f, const void* p, size_t n ) {
        if( n && fwrite( p, 1, n, f ) != n )
            throw std::runtime_error( "Writing the index failed" );
    };
    {
        FILE* f = open( sPrefix + ".bwt" ); // primary, L2[1..4], words
        put( f, &primary, 8 );
        put( f, &L2[ 1 ], 32 );
        put( f, vBwt.data( ), nWords * 4 );
        fclose( f );
    }
    {
        FILE* f = open( sPrefix + ".sa" ); // primary, L2[1..4], sampling interval, sequence length, sa[1..]
        const int32_t iInterval = 32;
        put( f, &primary, 8 );
        put( f, &L2[ 1 ], 32 );
        put( f, &iInterval, 4 );
        put( f, &uiN, 8 );
        put( f, vSa.data( ) + 1, ( nSa - 1 ) * 8 );
        fclose( f );
    }
    {
        FILE* f = open( sPrefix + ".pac" ); // packed bases, a zero byte if the length is a multiple of 4, length % 4
        put( f, vPac.data( ), vPac.size( ) );
        const uint8_t zero = 0, check = (uint8_t)( uiF % 4 );
        if( uiF % 4 == 0 )
            put( f, &zero, 1 );
        put( f, &check, 1 );
        fclose( f );
    }
    {
        std::ofstream f( sPrefix + ".ann" );
        f << uiF << " " << nContigs << " " << 0 << "\n";
        for( int32_t i = 0; i < nContigs; i++ )
            f << 0 << " " << ( (size_t)i < pPack->vNames.size( ) ? pPack->vNames[ i ] : "chr" + std::to_string( i + 1 ) ) << " none\n"
              << vStarts[ i ] << " " << vLens[ i ] << " " << ( (size_t)i < pPack->vNumHoles.size( ) ? pPack->vNumHoles[ i ] : 0 )
              << "\n";
    }
    {
        std::ofstream f( sPrefix + ".amb" );
        f << uiF << " " << nContigs << " " << pPack->vHoles.size( ) << "\n";
        for( auto& rHole : pPack->vHoles )
            f << rHole.first << " " << rHole.second << " N\n";
    }
}

// Stage outputs keep the device batch alive so the next module continues where this one stopped.
struct DeviceBatch
{
    ma_batch* p = nullptr;
    ~DeviceBatch( )
    {
        if( p )
            ma_batch_destroy( p );
    }
};

class Segment : public libMS::Container // segment.h:31-113
{
  public:
    nucSeqIndex iStart = 0, iSize = 0;
    t_bwtIndex saStart = 0, saStartRevComp = 0, saSize = 0;
    nucSeqIndex start( ) const
    {
        return iStart;
    }
    nucSeqIndex size( ) const
    {
        return iSize;
    }
    nucSeqIndex end( ) const
    {
        return iStart + iSize;
    }
};
class SegmentVector : public libMS::Container, public std::vector<Segment> // segment.h:126-399
{
  public:
    std::shared_ptr<DeviceBatch> pBatch;
};
class Seed : public libMS::Container // seed.h:34-46
{
  public:
    nucSeqIndex iStart = 0, iSize = 0, uiPosOnReference = 0, uiDelta = 0;
    unsigned int uiAmbiguity = 0;
    bool bOnForwStrand = true;
    nucSeqIndex start( ) const
    {
        return iStart;
    }
    nucSeqIndex size( ) const
    {
        return iSize;
    }
    nucSeqIndex start_ref( ) const
    {
        return uiPosOnReference;
    }
};
class Seeds : public libMS::Container, public std::vector<Seed> // seed.h:249-638
{
  public:
    unsigned int index_of_strip = 0; // xStats.index_of_strip
};
class SoCPriorityQueue : public libMS::Container // soc.h:96-420; here an opaque handle on the extracted seeds
{
  public:
    std::shared_ptr<DeviceBatch> pBatch;
    std::shared_ptr<Seeds> pSeeds; // seeds after ExtractSeeds (read-only view)
    bool empty( ) const
    {
        return pSeeds == nullptr || pSeeds->empty( );
    }
};
enum MatchType // alignment.h:39-46
{
    seed,
    match,
    missmatch,
    insertion,
    deletion
};
class Alignment;
struct AlignmentStatistics // seed.h:219-248 (the members the paired path reads)
{
    std::weak_ptr<Alignment> pOther; // the mate's alignment picked by PairedReads
    bool bFirst = false; // alignment of the first mate?
};
class Alignment : public libMS::Container // alignment.h:55-84
{
  public:
    std::vector<std::pair<MatchType, nucSeqIndex>> data;
    nucSeqIndex uiBeginOnRef = 0, uiEndOnRef = 0, uiBeginOnQuery = 0, uiEndOnQuery = 0;
    int64_t iScore = 0;
    double fMappingQuality = NAN;
    unsigned int index_of_strip = 0;
    bool bSecondary = false, bSupplementary = false;
    AlignmentStatistics xStats;
    int64_t score( ) const
    {
        return iScore;
    }
    nucSeqIndex beginOnRef( ) const
    {
        return uiBeginOnRef;
    }
    nucSeqIndex length( ) const // alignment.h:760-770
    {
        nucSeqIndex n = 0;
        for( auto& x : data )
            n += x.second;
        return n;
    }
    size_t getNumSeeds( ) const // alignment.h:239-246
    {
        size_t uiRet = 0;
        for( auto& x : data )
            if( x.first == MatchType::seed )
                uiRet++;
        return uiRet;
    }
    // Alignment::append (alignment.cpp:10-98): run-length storage; the score follows the global scoring parameters,
    // an indel run costs at most the SV penalty
    void append( MatchType type, nucSeqIndex size, const ma_params& rP )
    {
        if( size == 0 )
            return;
        if( type == MatchType::seed || type == MatchType::match )
        {
            iScore += (int64_t)rP.match * (int64_t)size;
            uiEndOnRef += size;
            uiEndOnQuery += size;
        }
        else if( type == MatchType::missmatch )
        {
            iScore -= (int64_t)rP.mismatch * (int64_t)size;
            uiEndOnRef += size;
            uiEndOnQuery += size;
        }
        else
        {
            if( type == MatchType::insertion )
                uiEndOnQuery += size;
            else
                uiEndOnRef += size;
            auto cost = [ & ]( nucSeqIndex n ) {
                const nucSeqIndex c = (nucSeqIndex)rP.extend * n + (nucSeqIndex)rP.gap;
                return c < (nucSeqIndex)rP.sv_penalty ? c : (nucSeqIndex)rP.sv_penalty;
            };
            if( data.size( ) != 0 && data.back( ).first == type )
            {
                size += data.back( ).second;
                iScore += (int64_t)cost( data.back( ).second );
                data.pop_back( );
            }
            iScore -= (int64_t)cost( size );
        }
        if( data.size( ) != 0 && data.back( ).first == type )
            data.back( ).second += size;
        else
            data.push_back( std::make_pair( type, size ) );
    }
};
typedef libMS::ContainerVector<std::shared_ptr<Seeds>> SeedsSetVector;
class AlignmentVector : public libMS::ContainerVector<std::shared_ptr<Alignment>>
{
  public:
    std::shared_ptr<DeviceBatch> pBatch;
};
class HarmonizedSets : public SeedsSetVector
{
  public:
    std::shared_ptr<DeviceBatch> pBatch;
};

namespace detail
{
inline std::shared_ptr<DeviceBatch> requireBatch( const std::shared_ptr<DeviceBatch>& p, const char* sWho )
{
    if( p == nullptr || p->p == nullptr )
        throw std::runtime_error( std::string( sWho ) +
                                  ": input container was not produced by the preceding MI355X module of this graph" );
    return p;
}
inline void fillAlignments( ma_batch* b, bool bMapq, libMS::ContainerVector<std::shared_ptr<Alignment>>& out )
{
    uint64_t nAln = 0, nOps = 0;
    maCheck( ma_batch_counts( b, nullptr, nullptr, nullptr, nullptr, &nAln, &nOps, nullptr ) );
    std::vector<uint64_t> off( 2 ), ops( 2 * nOps + 2 );
    std::vector<ma_alignment> alns( nAln + 1 );
    maCheck( ( bMapq ? ma_batch_get_mapq_alignments : ma_batch_get_alignments )( b, off.data( ), alns.data( ), ops.data( ) ) );
    for( uint64_t i = 0; i < off[ 1 ]; i++ )
    {
        auto pA = std::make_shared<Alignment>( );
        pA->uiBeginOnRef = alns[ i ].begin_ref;
        pA->uiEndOnRef = alns[ i ].end_ref;
        pA->uiBeginOnQuery = alns[ i ].begin_q;
        pA->uiEndOnQuery = alns[ i ].end_q;
        pA->iScore = alns[ i ].score;
        pA->index_of_strip = alns[ i ].soc_index;
        pA->bSecondary = alns[ i ].secondary != 0;
        pA->bSupplementary = alns[ i ].supplementary != 0;
        pA->fMappingQuality = bMapq ? alns[ i ].mapq : NAN;
        for( uint32_t k = 0; k < alns[ i ].n_ops; k++ )
            pA->data.emplace_back( (MatchType)ops[ 2 * ( alns[ i ].ops_off + k ) ], ops[ 2 * ( alns[ i ].ops_off + k ) + 1 ] );
        out.push_back( pA );
    }
}
} // namespace detail

class BinarySeeding : public libMS::Module<SegmentVector, false, SuffixArrayInterface, NucSeq>
{
    ma_params xP;

  public:
    BinarySeeding( const ParameterSetManager& rParameters ) : xP( *rParameters.getSelected( ) )
    {}
    // binarySeeding.cpp:86-178
    virtual std::shared_ptr<SegmentVector> execute( std::shared_ptr<SuffixArrayInterface> pFM_index,
                                                    std::shared_ptr<NucSeq> pQuerySeq ) override
    {
        auto pRet = std::make_shared<SegmentVector>( );
        if( pQuerySeq == nullptr )
            return pRet;
        pRet->pBatch = std::make_shared<DeviceBatch>( );
        maCheck( ma_batch_create( pFM_index->pDev->p, &xP, 1, pQuerySeq->length( ) + 64, &pRet->pBatch->p ) );
        const uint64_t off[ 2 ] = { 0, pQuerySeq->length( ) };
        const uint8_t dummy = 0;
        maCheck( ma_batch_set_reads( pRet->pBatch->p, pQuerySeq->length( ) ? pQuerySeq->xCodes.data( ) : &dummy, off, 1 ) );
        maCheck( ma_seed_batch( pRet->pBatch->p ) );
        uint64_t nSeg = 0;
        maCheck( ma_batch_counts( pRet->pBatch->p, &nSeg, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr ) );
        std::vector<ma_segment> segs( nSeg + 1 );
        uint64_t so[ 2 ];
        maCheck( ma_batch_get_segments( pRet->pBatch->p, so, segs.data( ) ) );
        for( uint64_t i = 0; i < so[ 1 ]; i++ )
        {
            Segment s;
            s.iStart = segs[ i ].q_start;
            s.iSize = segs[ i ].q_size;
            s.saStart = segs[ i ].sa_start;
            s.saStartRevComp = segs[ i ].sa_start_rc;
            s.saSize = segs[ i ].sa_size;
            pRet->push_back( s );
        }
        return pRet;
    }
};

class StripOfConsideration : public libMS::Module<SoCPriorityQueue, false, SegmentVector, NucSeq, Pack, FMIndex>
{
  public:
    StripOfConsideration( const ParameterSetManager& )
    {}
    // stripOfConsideration.cpp:162-173 (ExtractSeeds; the sweep itself runs fused with Harmonization on the device)
    virtual std::shared_ptr<SoCPriorityQueue> execute( std::shared_ptr<SegmentVector> pSegments, std::shared_ptr<NucSeq>,
                                                       std::shared_ptr<Pack>, std::shared_ptr<FMIndex> ) override
    {
        auto pRet = std::make_shared<SoCPriorityQueue>( );
        pRet->pBatch = detail::requireBatch( pSegments->pBatch, "StripOfConsideration" );
        maCheck( ma_extract_seeds_batch( pRet->pBatch->p ) );
        uint64_t nSeeds = 0;
        maCheck( ma_batch_counts( pRet->pBatch->p, nullptr, &nSeeds, nullptr, nullptr, nullptr, nullptr, nullptr ) );
        std::vector<ma_seed> v( nSeeds + 1 );
        uint64_t so[ 2 ];
        maCheck( ma_batch_get_seeds( pRet->pBatch->p, so, v.data( ) ) );
        pRet->pSeeds = std::make_shared<Seeds>( );
        for( uint64_t i = 0; i < so[ 1 ]; i++ )
        {
            Seed s;
            s.iStart = v[ i ].q_start, s.iSize = v[ i ].len, s.uiPosOnReference = v[ i ].r_start;
            s.uiDelta = v[ i ].delta, s.uiAmbiguity = v[ i ].ambiguity, s.bOnForwStrand = v[ i ].on_forward != 0;
            pRet->pSeeds->push_back( s );
        }
        return pRet;
    }
};

class Harmonization : public libMS::Module<SeedsSetVector, false, SoCPriorityQueue, NucSeq, FMIndex>
{
  public:
    Harmonization( const ParameterSetManager& )
    {}
    // harmonization.cpp:374-555 (+ the SoC sweep of stripOfConsideration.cpp:12-161)
    virtual std::shared_ptr<SeedsSetVector> execute( std::shared_ptr<SoCPriorityQueue> pSoCIn, std::shared_ptr<NucSeq>,
                                                     std::shared_ptr<FMIndex> ) override
    {
        auto pRet = std::make_shared<HarmonizedSets>( );
        pRet->pBatch = detail::requireBatch( pSoCIn->pBatch, "Harmonization" );
        maCheck( ma_chain_batch( pRet->pBatch->p ) );
        uint64_t nSets = 0, nSeeds = 0;
        maCheck( ma_batch_counts( pRet->pBatch->p, nullptr, nullptr, &nSets, &nSeeds, nullptr, nullptr, nullptr ) );
        std::vector<uint64_t> hoff( 2 ), soff( nSets + 1 );
        std::vector<uint32_t> soc( nSets + 1 );
        std::vector<ma_seed> v( nSeeds + 1 );
        maCheck( ma_batch_get_hsets( pRet->pBatch->p, hoff.data( ), soff.data( ), soc.data( ), v.data( ) ) );
        for( uint64_t h = 0; h < nSets; h++ )
        {
            auto pS = std::make_shared<Seeds>( );
            pS->index_of_strip = soc[ h ];
            for( uint64_t i = soff[ h ]; i < soff[ h + 1 ]; i++ )
            {
                Seed s;
                s.iStart = v[ i ].q_start, s.iSize = v[ i ].len, s.uiPosOnReference = v[ i ].r_start;
                s.uiDelta = v[ i ].delta, s.uiAmbiguity = v[ i ].ambiguity, s.bOnForwStrand = v[ i ].on_forward != 0;
                pS->push_back( s );
            }
            pRet->push_back( pS );
        }
        return pRet;
    }
};

class NeedlemanWunsch
    : public libMS::Module<libMS::ContainerVector<std::shared_ptr<Alignment>>, false, SeedsSetVector, NucSeq, Pack>
{
  public:
    NeedlemanWunsch( const ParameterSetManager& )
    {}
    // needlemanWunsch.h:111-134
    virtual std::shared_ptr<libMS::ContainerVector<std::shared_ptr<Alignment>>>
    execute( std::shared_ptr<SeedsSetVector> pSeedSets, std::shared_ptr<NucSeq>, std::shared_ptr<Pack> ) override
    {
        auto pIn = std::dynamic_pointer_cast<HarmonizedSets>( pSeedSets );
        auto pRet = std::make_shared<AlignmentVector>( );
        pRet->pBatch = detail::requireBatch( pIn ? pIn->pBatch : nullptr, "NeedlemanWunsch" );
        maCheck( ma_dp_batch( pRet->pBatch->p ) );
        detail::fillAlignments( pRet->pBatch->p, false, *pRet );
        return pRet;
    }
};

class MappingQuality : public libMS::Module<libMS::ContainerVector<std::shared_ptr<Alignment>>, false, NucSeq,
                                            libMS::ContainerVector<std::shared_ptr<Alignment>>>
{
  public:
    MappingQuality( const ParameterSetManager& )
    {}
    // mappingQuality.cpp:11-131 (computed on the device together with the DP stage; fetched here)
    virtual std::shared_ptr<libMS::ContainerVector<std::shared_ptr<Alignment>>>
    execute( std::shared_ptr<NucSeq>, std::shared_ptr<libMS::ContainerVector<std::shared_ptr<Alignment>>> pAlignments ) override
    {
        auto pIn = std::dynamic_pointer_cast<AlignmentVector>( pAlignments );
        auto pRet = std::make_shared<AlignmentVector>( );
        pRet->pBatch = detail::requireBatch( pIn ? pIn->pBatch : nullptr, "MappingQuality" );
        detail::fillAlignments( pRet->pBatch->p, true, *pRet );
        return pRet;
    }
};

// SmallInversions (smallInversions.h:22-221): the second consumer of kswcpp.  The scan for z-drops between seeds and the
// assembly of the inversion alignments are host glue; the DP calls of ALL alignments handed in (one read, or a
// whole batch through executeBatch) go to the GPU in one ma_ksw_batch launch, their reference windows come from the
// device-resident pack in one ma_pack_extract call.
class SmallInversions : public libMS::Module<libMS::ContainerVector<std::shared_ptr<Alignment>>, false,
                                             libMS::ContainerVector<std::shared_ptr<Alignment>>, NucSeq, Pack>
{
    const ma_params xP;
    struct DropPos
    {
        size_t uiRead, uiAlignment;
        nucSeqIndex uiStartQ, uiStartR, uiEndQ, uiEndR;
    };

  public:
    typedef libMS::ContainerVector<std::shared_ptr<Alignment>> TP_ALIGNMENTS;
    SmallInversions( const ParameterSetManager& rParameters ) : xP( *rParameters.getSelected( ) )
    {}

    // smallInversions.h:54-120
    template <typename F> void forAllDropPos( F&& fDo, const Alignment& rAlignment ) const
    {
        nucSeqIndex uiMaxScorePosQ = rAlignment.uiBeginOnQuery, uiPosQ = uiMaxScorePosQ, uiStartQ = uiMaxScorePosQ;
        nucSeqIndex uiMaxScorePosR = rAlignment.uiBeginOnRef, uiPosR = uiMaxScorePosR, uiStartR = uiMaxScorePosR;
        int iMaxScore = std::numeric_limits<int>::min( ), iCurrScore = 0, iMaxDrop = 0;
        for( const std::pair<MatchType, nucSeqIndex>& section : rAlignment.data )
        {
            if( section.first == MatchType::seed )
            {
                if( iMaxDrop >= (int)(size_t)xP.zdrop_inversion )
                    fDo( uiStartQ, uiStartR, uiPosQ, uiPosR );
                uiStartQ = section.second + uiPosQ;
                uiStartR = section.second + uiPosR;
                iMaxDrop = 0;
                iCurrScore = 0;
                iMaxScore = std::numeric_limits<int>::min( );
            }
            switch( section.first )
            {
                case MatchType::seed: // the reference falls through from seed into match
                case MatchType::match:
                    iCurrScore += xP.match * (int)section.second;
                    uiPosQ += section.second;
                    uiPosR += section.second;
                    break;
                case MatchType::missmatch:
                    iCurrScore -= xP.mismatch * (int)section.second;
                    uiPosQ += section.second;
                    uiPosR += section.second;
                    break;
                case MatchType::insertion:
                    iCurrScore -= xP.gap + xP.extend * (int)section.second;
                    uiPosQ += section.second;
                    break;
                case MatchType::deletion:
                    iCurrScore -= xP.gap + xP.extend * (int)section.second;
                    uiPosR += section.second;
                    break;
            }
            if( iCurrScore >= iMaxScore )
            {
                iMaxScore = iCurrScore;
                uiMaxScorePosQ = uiPosQ;
                uiMaxScorePosR = uiPosR;
            }
            else
            {
                const int uiDiff = (int)std::max( uiPosQ - uiMaxScorePosQ, uiPosR - uiMaxScorePosR );
                iMaxDrop = std::max( iMaxDrop, iMaxScore - iCurrScore - uiDiff * xP.extend );
            }
        }
    }

    // smallInversions.h:196-218 for many reads at once; vIn[i] belongs to vQueries[i]
    std::vector<std::shared_ptr<TP_ALIGNMENTS>> executeBatch( const std::vector<std::shared_ptr<TP_ALIGNMENTS>>& vIn,
                                                               const std::vector<std::shared_ptr<NucSeq>>& vQueries,
                                                               std::shared_ptr<Pack> pRefPack ) const
    {
        // 1. where does the running score drop by more than "Z Drop Inversions" between two seeds?
        std::vector<DropPos> vPos;
        for( size_t r = 0; r < vIn.size( ); r++ )
            for( size_t a = 0; a < vIn[ r ]->size( ); a++ )
                forAllDropPos(
                    [ & ]( nucSeqIndex uiStartQ, nucSeqIndex uiStartR, nucSeqIndex uiEndQ, nucSeqIndex uiEndR ) {
                        vPos.push_back( DropPos{ r, a, uiStartQ, uiStartR, uiEndQ, uiEndR } );
                    },
                    *( *vIn[ r ] )[ a ] );
        // 2. the reverse-strand windows of those stretches (Pack::uiPositionToReverseStrand pack.h:924-927, vExtract)
        const uint64_t uiN = vPos.empty( ) ? 0 : pRefPack->uiUnpackedSizeForwardPlusReverse( );
        std::vector<uint64_t> vBegin, vEnd;
        std::vector<ma_ksw_job> vJobs;
        std::vector<uint8_t> vQ, vT;
        std::vector<int64_t> vJobOfPos;
        std::vector<uint64_t> vTOff;
        uint64_t uiCigarCap = 16;
        for( const DropPos& rD : vPos )
        {
            vBegin.push_back( uiN - ( rD.uiEndR + 1 ) );
            vEnd.push_back( uiN - ( rD.uiStartR + 1 ) );
            ma_ksw_job j;
            j.qlen = (int)rD.uiEndQ - (int)rD.uiStartQ;
            j.tlen = (int)( vEnd.back( ) - vBegin.back( ) );
            j.w = (int)(size_t)xP.bandwidth_ext;
            j.zdrop = (int)(size_t)xP.zdrop;
            j.flag = 0;
            j.reserved = 0;
            j.q_off = vQ.size( );
            j.t_off = vT.size( );
            vTOff.push_back( vT.size( ) );
            const NucSeq& rQuery = *vQueries[ rD.uiRead ];
            vQ.insert( vQ.end( ), rQuery.xCodes.begin( ) + rD.uiStartQ, rQuery.xCodes.begin( ) + rD.uiEndQ );
            vT.resize( vT.size( ) + (size_t)j.tlen );
            uiCigarCap += (uint64_t)j.qlen + (uint64_t)j.tlen + 2;
            // kswcpp returns an empty result for an empty side (kswcpp_core.h:362-364): nothing to launch
            vJobOfPos.push_back( j.qlen > 0 && j.tlen > 0 ? (int64_t)vJobs.size( ) : -1 );
            if( vJobOfPos.back( ) >= 0 )
                vJobs.push_back( j );
        }
        std::vector<ma_ez> vEz( vJobs.size( ) + 1 );
        std::vector<uint64_t> vCigOff( vJobs.size( ) + 2, 0 );
        std::vector<uint32_t> vCigar( uiCigarCap );
        if( !vPos.empty( ) )
            maCheck( ma_pack_extract( pRefPack->pDev->p, vBegin.data( ), vEnd.data( ), vBegin.size( ), vT.data( ) ) );
        vQ.push_back( 0 );
        vT.push_back( 0 );
        if( !vJobs.empty( ) )
        {
            // 3. kswcpp_dispatch( ..., xKswParameters, uiBandwidth, uiZDrop, 0, ... ) (smallInversions.h:131-133), all at once
            maCheck( ma_ksw_batch( &xP, vJobs.data( ), vJobs.size( ), vQ.data( ), vQ.size( ), vT.data( ), vT.size( ), vEz.data( ),
                                   vCigOff.data( ), vCigar.data( ), uiCigarCap ) );
        }
        // 4. cigars -> alignments (smallInversions.h:135-175), appended behind the alignment they were found in
        std::vector<std::shared_ptr<TP_ALIGNMENTS>> vRet;
        size_t k = 0;
        for( size_t r = 0; r < vIn.size( ); r++ )
        {
            auto pRet = std::make_shared<TP_ALIGNMENTS>( );
            for( size_t a = 0; a < vIn[ r ]->size( ); a++ )
            {
                std::shared_ptr<Alignment> pAlignment = ( *vIn[ r ] )[ a ];
                pRet->push_back( pAlignment );
                for( ; k < vPos.size( ) && vPos[ k ].uiRead == r && vPos[ k ].uiAlignment == a; k++ )
                {
                    const NucSeq& rQuery = *vQueries[ r ];
                    const uint8_t* pRef = vT.data( ) + vTOff[ k ];
                    auto pInv = std::make_shared<Alignment>( );
                    nucSeqIndex qPos = vPos[ k ].uiStartQ, rPos = 0;
                    const int64_t iJob = vJobOfPos[ k ];
                    // the cigar of job j is n_cigar words at cigar_off[ j ] (the jobs' cigars are not stored in job order)
                    for( uint64_t c = iJob < 0 ? 0 : vCigOff[ iJob ]; c < ( iJob < 0 ? 0 : vCigOff[ iJob ] + (uint64_t)vEz[ iJob ].n_cigar ); c++ )
                    {
                        const uint32_t uiSymbol = vCigar[ c ] & 0xf, uiAmount = vCigar[ c ] >> 4;
                        switch( uiSymbol )
                        {
                            case 0:
                                for( uint32_t uiPos = 0; uiPos < uiAmount; uiPos++ )
                                    pInv->append( rQuery.xCodes[ uiPos + qPos ] == pRef[ uiPos + rPos ] ? MatchType::match
                                                                                                     : MatchType::missmatch,
                                                  1, xP );
                                qPos += uiAmount;
                                rPos += uiAmount;
                                break;
                            case 1:
                                pInv->append( MatchType::insertion, uiAmount, xP );
                                qPos += uiAmount;
                                break;
                            case 2:
                                pInv->append( MatchType::deletion, uiAmount, xP );
                                rPos += uiAmount;
                                break;
                            default:
                                throw std::runtime_error( "obtained wierd symbol from ksw" );
                        }
                    }
                    if( xP.disable_heuristics || pInv->score( ) > xP.harm_score_min * xP.match )
                    {
                        pInv->uiBeginOnQuery += vPos[ k ].uiStartQ;
                        pInv->uiEndOnQuery += vPos[ k ].uiStartQ;
                        pInv->uiBeginOnRef += vBegin[ k ];
                        pInv->uiEndOnRef += vBegin[ k ];
                        pInv->bSupplementary = true;
                        pInv->xStats = pAlignment->xStats;
                        pInv->index_of_strip = pAlignment->index_of_strip;
                        pInv->fMappingQuality = 0;
                        pRet->push_back( pInv );
                    }
                }
            }
            vRet.push_back( pRet );
        }
        return vRet;
    }

    virtual std::shared_ptr<TP_ALIGNMENTS> execute( std::shared_ptr<TP_ALIGNMENTS> pAlignments, std::shared_ptr<NucSeq> pQuery,
                                                    std::shared_ptr<Pack> pRefPack ) override
    {
        return executeBatch( { pAlignments }, { pQuery }, pRefPack )[ 0 ];
    }
};

// PairedReads (pairedReads.h:23-63, pairedReads.cpp:14-131): picks the best combination of one alignment per mate
class PairedReads : public libMS::Module<libMS::ContainerVector<std::shared_ptr<Alignment>>, false, NucSeq, NucSeq,
                                         libMS::ContainerVector<std::shared_ptr<Alignment>>,
                                         libMS::ContainerVector<std::shared_ptr<Alignment>>, Pack>
{
    const ma_params xP;

  public:
    typedef libMS::ContainerVector<std::shared_ptr<Alignment>> TP_ALIGNMENTS;
    double u; // score factor of a proper pair
    size_t mean; // insert size
    double std;
    PairedReads( const ParameterSetManager& rParameters )
        : xP( *rParameters.getSelected( ) ), u( xP.paired_bonus ), mean( (size_t)xP.mean_paired_dist ), std( xP.std_paired_dist )
    {}
    virtual std::shared_ptr<TP_ALIGNMENTS> execute( std::shared_ptr<NucSeq> pQ1, std::shared_ptr<NucSeq> pQ2,
                                                    std::shared_ptr<TP_ALIGNMENTS> pAlignments1,
                                                    std::shared_ptr<TP_ALIGNMENTS> pAlignments2, std::shared_ptr<Pack> pPack ) override
    {
        for( auto& pA : *pAlignments1 )
            pA->xStats.bFirst = true;
        for( auto& pA : *pAlignments2 )
            pA->xStats.bFirst = false;
        if( pAlignments1->size( ) == 0 )
            return pAlignments2;
        if( pAlignments2->size( ) == 0 )
            return pAlignments1;
        const uint64_t uiN = pPack->uiUnpackedSizeForwardPlusReverse( ), uiF = uiN / 2;
        std::vector<std::tuple<int64_t, bool, size_t, size_t>> vScores;
        for( size_t i = 0; i < pAlignments1->size( ); i++ )
        {
            const Alignment& rA1 = *( *pAlignments1 )[ i ];
            if( rA1.length( ) == 0 )
                continue;
            for( size_t j = 0; j < pAlignments2->size( ); j++ )
            {
                const Alignment& rA2 = *( *pAlignments2 )[ j ];
                if( rA2.length( ) == 0 )
                    continue;
                int64_t iScore = rA1.score( ) + rA2.score( );
                bool bIsPaired = false;
                // illumina reads must be on opposite strands
                if( ( rA1.beginOnRef( ) >= uiF ) != ( rA2.beginOnRef( ) >= uiF ) )
                {
                    const nucSeqIndex uiP1 = rA1.beginOnRef( ), uiP2 = uiN - ( rA2.beginOnRef( ) + 1 );
                    const nucSeqIndex d = uiP1 < uiP2 ? uiP2 - uiP1 : uiP1 - uiP2;
                    if( ( (double)d ) >= ( (double)mean ) - std * 3 && ( (double)d ) <= ( (double)mean ) + std * 3 )
                    {
                        iScore = ( int64_t )( iScore * u );
                        bIsPaired = true;
                    }
                }
                vScores.emplace_back( iScore, bIsPaired, i, j );
            }
        }
        if( vScores.empty( ) )
            throw std::runtime_error( "PairedReads: no alignment of non-zero length to pair" );
        // same container, comparator and libstdc++ std::sort as the reference: ties keep its order
        std::sort( vScores.begin( ), vScores.end( ),
                   []( const std::tuple<int64_t, bool, size_t, size_t>& rtA, const std::tuple<int64_t, bool, size_t, size_t>& rtB ) {
                       if( std::get<0>( rtA ) == std::get<0>( rtB ) )
                           return std::get<1>( rtA ) && !std::get<1>( rtB );
                       return std::get<0>( rtA ) > std::get<0>( rtB );
                   } );
        auto pA = ( *pAlignments1 )[ std::get<2>( vScores[ 0 ] ) ];
        auto pB = ( *pAlignments2 )[ std::get<3>( vScores[ 0 ] ) ];
        pA->bSecondary = pB->bSecondary = false;
        pA->bSupplementary = pB->bSupplementary = false;
        pA->xStats.pOther = std::weak_ptr<Alignment>( pB );
        pB->xStats.pOther = std::weak_ptr<Alignment>( pA );
        if( std::get<1>( vScores[ 0 ] ) && vScores.size( ) > 1 )
        {
            float fMapQ = ( (float)( std::get<0>( vScores[ 0 ] ) - std::get<0>( vScores[ 1 ] ) ) ) / std::get<0>( vScores[ 0 ] );
            if( pA->getNumSeeds( ) <= 1 && pB->getNumSeeds( ) <= 1 )
                fMapQ /= 2;
            if( pA->score( ) >= xP.match * pQ1->length( ) * 0.8 && pAlignments1->size( ) >= 3 )
                fMapQ *= 2;
            else if( pB->score( ) >= xP.match * pQ2->length( ) * 0.8 && pAlignments2->size( ) >= 3 )
                fMapQ *= 2;
            if( fMapQ > 1 )
                fMapQ = 1;
            pA->fMappingQuality = fMapQ;
            pB->fMappingQuality = fMapQ;
        }
        auto pRet = std::make_shared<TP_ALIGNMENTS>( );
        pRet->push_back( pA );
        pRet->push_back( pB );
        return pRet;
    }
};

// Throughput API: a whole batch of reads through all stages in one go.
class BatchAligner
    : public libMS::Module<libMS::ContainerVector<std::shared_ptr<AlignmentVector>>, false, FMIndex,
                           libMS::ContainerVector<std::shared_ptr<NucSeq>>>
{
    ma_params xP;
    ParameterSetManager xParams;

  public:
    BatchAligner( const ParameterSetManager& rParameters ) : xP( *rParameters.getSelected( ) ), xParams( rParameters )
    {}
    virtual std::shared_ptr<libMS::ContainerVector<std::shared_ptr<AlignmentVector>>>
    execute( std::shared_ptr<FMIndex> pFM_index, std::shared_ptr<libMS::ContainerVector<std::shared_ptr<NucSeq>>> pQueries ) override
    {
        std::vector<uint64_t> off{ 0 };
        std::vector<uint8_t> cat;
        for( auto& q : *pQueries )
        {
            cat.insert( cat.end( ), q->xCodes.begin( ), q->xCodes.end( ) );
            off.push_back( cat.size( ) );
        }
        cat.push_back( 0 );
        DeviceBatch B;
        maCheck( ma_batch_create( pFM_index->pDev->p, &xP, pQueries->size( ) + 1, cat.size( ) + 64, &B.p ) );
        maCheck( ma_batch_set_reads( B.p, cat.data( ), off.data( ), pQueries->size( ) ) );
        maCheck( ma_align_batch( B.p ) );
        uint64_t nAln = 0, nOps = 0;
        maCheck( ma_batch_counts( B.p, nullptr, nullptr, nullptr, nullptr, &nAln, &nOps, nullptr ) );
        std::vector<uint64_t> aoff( pQueries->size( ) + 1 ), ops( 2 * nOps + 2 );
        std::vector<ma_alignment> alns( nAln + 1 );
        maCheck( ma_batch_get_mapq_alignments( B.p, aoff.data( ), alns.data( ), ops.data( ) ) );
        auto pRet = std::make_shared<libMS::ContainerVector<std::shared_ptr<AlignmentVector>>>( );
        for( size_t r = 0; r < pQueries->size( ); r++ )
        {
            auto pV = std::make_shared<AlignmentVector>( );
            for( uint64_t i = aoff[ r ]; i < aoff[ r + 1 ]; i++ )
            {
                auto pA = std::make_shared<Alignment>( );
                pA->uiBeginOnRef = alns[ i ].begin_ref, pA->uiEndOnRef = alns[ i ].end_ref;
                pA->uiBeginOnQuery = alns[ i ].begin_q, pA->uiEndOnQuery = alns[ i ].end_q;
                pA->iScore = alns[ i ].score, pA->index_of_strip = alns[ i ].soc_index;
                pA->bSecondary = alns[ i ].secondary != 0, pA->bSupplementary = alns[ i ].supplementary != 0;
                pA->fMappingQuality = alns[ i ].mapq;
                for( uint32_t k = 0; k < alns[ i ].n_ops; k++ )
                    pA->data.emplace_back( (MatchType)ops[ 2 * ( alns[ i ].ops_off + k ) ],
                                           ops[ 2 * ( alns[ i ].ops_off + k ) + 1 ] );
                pV->push_back( pA );
            }
            pRet->push_back( pV );
        }
        // "Detect Small Inversions" (export.cpp:118-121): all reads' inversion DP in one more GPU launch
        if( xP.search_inversions )
        {
            auto pPack = std::make_shared<Pack>( );
            pPack->pDev = pFM_index->pDev;
            std::vector<std::shared_ptr<SmallInversions::TP_ALIGNMENTS>> vIn;
            std::vector<std::shared_ptr<NucSeq>> vQ;
            for( size_t r = 0; r < pQueries->size( ); r++ )
            {
                vIn.push_back( ( *pRet )[ r ] );
                vQ.push_back( ( *pQueries )[ r ] );
            }
            auto vOut = SmallInversions( xParams ).executeBatch( vIn, vQ, pPack );
            for( size_t r = 0; r < pQueries->size( ); r++ )
            {
                auto pV = std::make_shared<AlignmentVector>( );
                for( auto& pA : *vOut[ r ] )
                    pV->push_back( pA );
                ( *pRet )[ r ] = pV;
            }
        }
        return pRet;
    }

    // Paired mode (setUpCompGraphPaired, export.cpp:130-202) for a batch: vMates holds the mates of pair k at 2k and
    // 2k + 1; both mates of all pairs go through ONE device batch, PairedReads then picks per pair on the host.
    std::shared_ptr<libMS::ContainerVector<std::shared_ptr<AlignmentVector>>>
    executePaired( std::shared_ptr<FMIndex> pFM_index, std::shared_ptr<libMS::ContainerVector<std::shared_ptr<NucSeq>>> vMates )
    {
        if( vMates->size( ) % 2 )
            throw std::runtime_error( "BatchAligner::executePaired: odd number of reads" );
        auto pPerRead = execute( pFM_index, vMates );
        auto pPack = std::make_shared<Pack>( );
        pPack->pDev = pFM_index->pDev;
        PairedReads xPairedReads( xParams );
        auto pRet = std::make_shared<libMS::ContainerVector<std::shared_ptr<AlignmentVector>>>( );
        for( size_t k = 0; 2 * k + 1 < vMates->size( ); k++ )
        {
            auto pPicked = xPairedReads.execute( ( *vMates )[ 2 * k ], ( *vMates )[ 2 * k + 1 ], ( *pPerRead )[ 2 * k ],
                                                 ( *pPerRead )[ 2 * k + 1 ], pPack );
            auto pV = std::make_shared<AlignmentVector>( );
            for( auto& pA : *pPicked )
                pV->push_back( pA );
            pRet->push_back( pV );
        }
        return pRet;
    }
};
} // namespace libMA
