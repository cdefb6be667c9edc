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
// convention of module.h:339-377).  The per-read execute() calls of all graph threads are funnelled into device batches
// (ma_engine.h: DeviceBatcher); BatchAligner / MultiDeviceAligner below are the throughput APIs for callers that hold
// many reads at once (one or several GPUs, 10^5..10^6 reads per device batch).
#pragma once
#include "../../include/ma_amd.h"
#include "ms_graph.h"
#include "ma_engine.h"
#include <algorithm>
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
namespace detail
{
using namespace ma_amd::engine; // Engine, DeviceBatcher, Ticket, BatchResult (reference-free, shared with ma_ref_binding.h)
}
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
        // the C ABI holds the preset table (parameter.h:1079-1104); an unknown key fails with the reference's text
        if( ma_params_preset( sKey.c_str( ), &xSelected ) != 0 )
            throw std::runtime_error( ma_last_error( ) );
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
    // set by PrefetchReader: this read already went through all stages on the GPU as read uiRead of that device batch
    // (the binding on the reference's own NucSeq carries it in a subclass, ma_ref_binding.h)
    detail::Ticket xTicket;
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
// vReplicas: copies of the same index on OTHER devices of the node (replicateIndex below).  The throughput forms --
// PrefetchReader, BatchAlign, BatchAligner -- rotate their device batches over p and the replicas, so the one FMIndex / Pack
// pledge of the graph (export.cpp:99-126) feeds all GPUs; everything else (single-stage calls, pack extraction) uses p.
struct DeviceIndex
{
    ma_index* p = nullptr;
    std::vector<std::shared_ptr<DeviceIndex>> vReplicas;
    ~DeviceIndex( )
    {
        if( p )
            ma_index_destroy( p );
    }
    std::vector<const ma_index*> all( ) const
    {
        std::vector<const ma_index*> v( 1, p );
        for( const auto& pR : vReplicas )
            v.push_back( pR->p );
        return v;
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
    // optional host copy of the packed forward strand (2 bit / base, MSB first): consumers of reference BASES on the host
    // (FileWriter's NGMLR tag emulation) use it when present and ask the device otherwise (ma_pack_extract)
    std::vector<uint8_t> vPacHost;
    uint64_t uiUnpackedSizeForwardPlusReverse( ) const // pack.h:881-884: twice the forward strand
    {
        if( !vStarts.empty( ) )
            return 2 * ( vStarts.back( ) + vLengths.back( ) );
        uint64_t n = 0;
        maCheck( ma_index_sizes( pDev->p, nullptr, nullptr, &n, nullptr ) );
        return n;
    }
};

// One copy of pDev's index on every device of vDevices (the index is replicated, 11.9 GB of 288 GB per GPU; the
// arrays are downloaded once and uploaded per device, each upload on a thread bound to its device).  The copies are attached
// to pDev (DeviceIndex::vReplicas) and returned.  A device may be named more than once and may be the original's own: tests
// on a one-GPU box run "virtual shards" on device 0.
inline std::vector<std::shared_ptr<DeviceIndex>> replicateIndex( const std::shared_ptr<DeviceIndex>& pDev, const std::vector<int>& vDevices )
{
    std::vector<std::shared_ptr<DeviceIndex>> vNew;
    for( ma_index* pCopy : detail::replicateOnDevices( pDev->p, vDevices ) )
    {
        vNew.push_back( std::make_shared<DeviceIndex>( ) );
        vNew.back( )->p = pCopy;
    }
    for( const auto& pCopy : vNew )
        pDev->vReplicas.push_back( pCopy );
    return vNew;
}

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

// Every stage output remembers, where it has one, the device batch its read was part of (a Ticket): the next module then
// only picks the read's slice of records that are already in host memory.  A container WITHOUT a ticket -- built by client
// code or by a module of the reference -- is uploaded instead and the stage runs for it alone (stage inputs of the C ABI:
// ma_batch_set_segments / _seeds / _hsets), so the MI355X modules can replace the reference's one stage at a time.
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
    SAInterval saInterval( ) const
    {
        return SAInterval( saStart, saStartRevComp, saSize );
    }
};
class SegmentVector : public libMS::Container, public std::vector<Segment> // segment.h:126-399
{
  public:
    detail::Ticket xTicket;
};
class Seed : public libMS::Container // seed.h:34-46
{
  public:
    nucSeqIndex iStart = 0, iSize = 0, uiPosOnReference = 0, uiDelta = 0;
    unsigned int uiAmbiguity = 0;
    bool bOnForwStrand = true;
    nucSeqIndex uiSoCNt = 0; // accumulated length of the strip the seed was popped with (soc.h:276)
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
namespace detail
{
inline Seed toSeed( const ma_seed& r )
{
    Seed s;
    s.iStart = r.q_start, s.iSize = r.len, s.uiPosOnReference = r.r_start;
    s.uiDelta = r.delta, s.uiAmbiguity = r.ambiguity, s.bOnForwStrand = r.on_forward != 0;
    return s;
}
inline ma_seed fromSeed( const Seed& s )
{
    ma_seed r;
    r.q_start = (int64_t)s.iStart, r.len = (int64_t)s.iSize, r.r_start = (int64_t)s.uiPosOnReference;
    r.delta = (int64_t)s.uiDelta, r.ambiguity = s.uiAmbiguity, r.on_forward = s.bOnForwStrand ? 1 : 0;
    return r;
}
// a device batch of exactly one read for the stage-at-a-time path
struct SingleRead
{
    ma_batch* p = nullptr;
    SingleRead( const ma_index* pIndex, const ma_params& rP, const NucSeq& rQuery )
    {
        maCheck( ma_batch_create( pIndex, &rP, 1, rQuery.length( ) + 64, &p ) );
        const uint64_t off[ 2 ] = { 0, rQuery.length( ) };
        const uint8_t dummy = 0;
        try
        {
            maCheck( ma_batch_set_reads( p, rQuery.length( ) ? rQuery.xCodes.data( ) : &dummy, off, 1 ) );
        }
        catch( ... )
        {
            ma_batch_destroy( p );
            throw;
        }
    }
    SingleRead( const SingleRead& ) = delete;
    ~SingleRead( )
    {
        ma_batch_destroy( p );
    }
};
} // namespace detail

// SoCPriorityQueue (soc.h:96-420): the strips of consideration of one read, best first.  pSeeds are the read's extracted
// seeds (ExtractSeeds order); pop() / empty() / size() serve the queue itself, which the device computes on demand
// (ma_batch_get_socs: the sweep of stripOfConsideration.cpp:12-161 and the heap pops of soc.h:240-284 run on the GPU,
// here the strips are only handed out).
class SoCPriorityQueue : public libMS::Container
{
    std::vector<ma_soc> vSocs;
    std::vector<ma_seed> vSorted;
    size_t uiNext = 0;
    bool bHaveQueue = false;
    void ensureQueue( )
    {
        if( bHaveQueue )
            return;
        bHaveQueue = true;
        if( pSeeds == nullptr || pSeeds->empty( ) )
            return;
        if( pFmIndex == nullptr || pQuery == nullptr )
            throw std::runtime_error( "SoCPriorityQueue: the queue was not produced by StripOfConsideration" );
        detail::SingleRead xB( pFmIndex->pDev->p, xP, *pQuery );
        std::vector<ma_seed> v;
        for( const Seed& s : *pSeeds )
            v.push_back( detail::fromSeed( s ) );
        const uint64_t off[ 2 ] = { 0, v.size( ) };
        maCheck( ma_batch_set_seeds( xB.p, off, v.data( ) ) );
        uint64_t nSocs = 0;
        maCheck( ma_batch_get_socs( xB.p, &nSocs, nullptr, nullptr, nullptr, nullptr ) );
        vSocs.resize( nSocs + 1 );
        vSorted.resize( v.size( ) + 1 );
        uint64_t so[ 2 ], sd[ 2 ];
        maCheck( ma_batch_get_socs( xB.p, &nSocs, so, vSocs.data( ), sd, vSorted.data( ) ) );
        vSocs.resize( nSocs );
    }

  public:
    detail::Ticket xTicket;
    std::shared_ptr<Seeds> pSeeds; // seeds after ExtractSeeds (read-only view)
    std::shared_ptr<FMIndex> pFmIndex; // (the handle StripOfConsideration was called with: copying it touches the caller's control block)
    std::shared_ptr<NucSeq> pQuery;
    ma_params xP;
    bool empty( )
    {
        ensureQueue( );
        return uiNext >= vSocs.size( );
    }
    size_t size( )
    {
        ensureQueue( );
        return vSocs.size( ) - uiNext;
    }
    // score and ambiguity of the strip pop() would return next (std::get<0>( vMaxima.front( ) ), soc.h:192)
    std::pair<nucSeqIndex, unsigned int> front( )
    {
        ensureQueue( );
        if( uiNext >= vSocs.size( ) )
            throw std::runtime_error( "SoCPriorityQueue::front on an empty queue" );
        return std::make_pair( (nucSeqIndex)vSocs[ uiNext ].acc_len, (unsigned int)vSocs[ uiNext ].ambiguity );
    }
    std::shared_ptr<Seeds> pop( ) // soc.h:240-284
    {
        ensureQueue( );
        if( uiNext >= vSocs.size( ) )
            throw std::runtime_error( "SoCPriorityQueue::pop on an empty queue" );
        const ma_soc& rSoc = vSocs[ uiNext ];
        auto pRet = std::make_shared<Seeds>( );
        pRet->index_of_strip = (unsigned int)uiNext++;
        for( uint32_t i = rSoc.begin; i < rSoc.end; i++ )
        {
            pRet->push_back( detail::toSeed( vSorted[ i ] ) );
            pRet->back( ).uiSoCNt = rSoc.acc_len;
        }
        return pRet;
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
    detail::Ticket xTicket;
    // stage-at-a-time path: the MappingQuality view of the same DP run (computed together on the device)
    std::shared_ptr<libMS::ContainerVector<std::shared_ptr<Alignment>>> pStandalone;
};
class HarmonizedSets : public SeedsSetVector
{
  public:
    detail::Ticket xTicket;
};

namespace detail
{
// alignment records [uiFrom, uiTo) of a download -> Alignment containers
inline void appendAlignments( const ma_alignment* vAlns, const uint64_t* vOps, uint64_t uiFrom,
                              uint64_t uiTo, bool bMapq, libMS::ContainerVector<std::shared_ptr<Alignment>>& rOut )
{
    for( uint64_t i = uiFrom; i < uiTo; i++ )
    {
        const ma_alignment& r = vAlns[ i ];
        auto pA = std::make_shared<Alignment>( );
        pA->uiBeginOnRef = r.begin_ref;
        pA->uiEndOnRef = r.end_ref;
        pA->uiBeginOnQuery = r.begin_q;
        pA->uiEndOnQuery = r.end_q;
        pA->iScore = r.score;
        pA->index_of_strip = r.soc_index;
        pA->bSecondary = r.secondary != 0;
        pA->bSupplementary = r.supplementary != 0;
        pA->fMappingQuality = bMapq ? r.mapq : NAN;
        pA->data.reserve( r.n_ops );
        for( uint32_t k = 0; k < r.n_ops; k++ )
            pA->data.emplace_back( (MatchType)vOps[ 2 * ( r.ops_off + k ) ], vOps[ 2 * ( r.ops_off + k ) + 1 ] );
        rOut.push_back( pA );
    }
}
inline void downloadAlignments( ma_batch* b, bool bMapq, libMS::ContainerVector<std::shared_ptr<Alignment>>& rOut )
{
    uint64_t nAln = 0, nOps = 0;
    maCheck( ma_batch_counts( b, nullptr, nullptr, nullptr, nullptr, &nAln, &nOps, nullptr ) );
    std::vector<uint64_t> off( 2 ), ops( 2 * nOps + 2 );
    std::vector<ma_alignment> alns( nAln + 1 );
    maCheck( ( bMapq ? ma_batch_get_mapq_alignments : ma_batch_get_alignments )( b, off.data( ), alns.data( ), ops.data( ) ) );
    appendAlignments( alns.data( ), ops.data( ), 0, off[ 1 ], bMapq, rOut );
}
// harmonized sets of one read out of CSR arrays
inline void appendHsets( const std::vector<uint64_t>& vHseedOff, const std::vector<uint32_t>& vSoc, const std::vector<ma_seed>& vSeeds,
                         uint64_t uiFrom, uint64_t uiTo, SeedsSetVector& rOut )
{
    for( uint64_t h = uiFrom; h < uiTo; h++ )
    {
        auto pS = std::make_shared<Seeds>( );
        pS->index_of_strip = vSoc[ h ];
        for( uint64_t i = vHseedOff[ h ]; i < vHseedOff[ h + 1 ]; i++ )
            pS->push_back( toSeed( vSeeds[ i ] ) );
        rOut.push_back( pS );
    }
}
// harmonized sets of one read -> CSR arrays for ma_batch_set_hsets
inline void uploadHsets( ma_batch* b, const SeedsSetVector& rSets )
{
    std::vector<uint64_t> hoff{ 0, rSets.size( ) }, soff{ 0 };
    std::vector<uint32_t> soc;
    std::vector<ma_seed> v;
    for( const auto& pS : rSets )
    {
        soc.push_back( pS->index_of_strip );
        for( const Seed& s : *pS )
            v.push_back( fromSeed( s ) );
        soff.push_back( v.size( ) );
    }
    soc.push_back( 0 );
    v.push_back( ma_seed( ) );
    maCheck( ma_batch_set_hsets( b, hoff.data( ), soff.data( ), soc.data( ), v.data( ) ) );
}
} // namespace detail

// Options of the funnel behind the per-read modules (process-wide default; a BinarySeeding instance copies it when it
// opens its batcher)
inline detail::BatcherOptions& defaultBatcherOptions( )
{
    static detail::BatcherOptions xOptions;
    return xOptions;
}

class BinarySeeding : public libMS::Module<SegmentVector, false, SuffixArrayInterface, NucSeq>
{
    ma_params xP;
    std::mutex xBatcherMutex;
    std::shared_ptr<DeviceIndex> pBatcherIndex;
    std::shared_ptr<detail::DeviceBatcher> pBatcher;

    std::shared_ptr<detail::DeviceBatcher> batcherFor( const std::shared_ptr<DeviceIndex>& pDev )
    {
        std::lock_guard<std::mutex> xGuard( xBatcherMutex );
        if( pBatcher == nullptr || pBatcherIndex != pDev )
        {
            pBatcher = std::make_shared<detail::DeviceBatcher>( pDev->p, xP, defaultBatcherOptions( ) );
            pBatcherIndex = pDev;
        }
        return pBatcher;
    }

  public:
    BinarySeeding( const ParameterSetManager& rParameters ) : xP( *rParameters.getSelected( ) )
    {}
    // binarySeeding.cpp:86-178.  Called concurrently by all graph threads: the reads of the callers that arrive together
    // go through the device as ONE batch, and through all stages at once (the modules downstream pick their slices).
    virtual std::shared_ptr<SegmentVector> execute( std::shared_ptr<SuffixArrayInterface> pFM_index,
                                                    std::shared_ptr<NucSeq> pQuerySeq ) override
    {
        auto pRet = std::make_shared<SegmentVector>( );
        if( pQuerySeq == nullptr )
            return pRet;
        // a read that came through PrefetchReader has been aligned already; otherwise it joins the funnel and this thread waits
        pRet->xTicket = pQuerySeq->xTicket ? pQuerySeq->xTicket : batcherFor( pFM_index->pDev )->align( pQuerySeq->xCodes );
        const detail::BatchResult& R = *pRet->xTicket.pResult;
        for( uint64_t i = R.bStages ? R.vSegOff[ pRet->xTicket.uiRead ] : 0; R.bStages && i < R.vSegOff[ pRet->xTicket.uiRead + 1 ]; i++ )
        {
            Segment s;
            s.iStart = R.vSegs[ i ].q_start;
            s.iSize = R.vSegs[ i ].q_size;
            s.saStart = R.vSegs[ i ].sa_start;
            s.saStartRevComp = R.vSegs[ i ].sa_start_rc;
            s.saSize = R.vSegs[ i ].sa_size;
            pRet->push_back( s );
        }
        return pRet;
    }
    std::shared_ptr<detail::DeviceBatcher> batcher( ) // diagnostics
    {
        std::lock_guard<std::mutex> xGuard( xBatcherMutex );
        return pBatcher;
    }
    // device batches run so far and the reads they carried (diagnostics)
    std::pair<uint64_t, uint64_t> batchStatistics( )
    {
        std::lock_guard<std::mutex> xGuard( xBatcherMutex );
        uint64_t b = 0, r = 0;
        if( pBatcher != nullptr )
            pBatcher->stats( b, r );
        return std::make_pair( b, r );
    }
};

// The funnel turned round: a volatile source with the signature of the reader it wraps
// (fileReader.h:475: FileReader : Module<NucSeq, true, FileStream>; here any Module<NucSeq, true, TP_ARGS...>).  It reads
// AHEAD: a device batch worth of reads is pulled from the wrapped reader, goes through ALL stages on the GPU, and every
// execute( ) hands the calling graph thread one read of a FINISHED batch with its ticket.  The five modules downstream find
// the ticket on the query and only pick their slices, so no graph thread ever waits for the GPU per read: the graph of
// export.cpp:99-126 stays as it is -- only the reader node is wrapped -- and runs with a few dozen threads instead of the
// thousand the per-read funnel (DeviceBatcher) needs to fill a device batch.
//     auto pReader = std::make_shared<PrefetchReader<FileStream>>( rParameters, std::make_shared<FileReader>( rParameters ), pFMDIndex );
template <typename... TP_ARGS> class PrefetchReader : public libMS::Module<NucSeq, true, TP_ARGS...>
{
    typedef libMS::Module<NucSeq, true, TP_ARGS...> TP_SOURCE;
    std::shared_ptr<TP_SOURCE> pSource;
    detail::PrefetchQueue<std::shared_ptr<NucSeq>> xQueue;

  public:
    PrefetchReader( const ParameterSetManager& rParameters, std::shared_ptr<TP_SOURCE> pSource, std::shared_ptr<FMIndex> pFM_index,
                    const detail::PrefetchOptions& rOpt = detail::PrefetchOptions( ) )
        : pSource( pSource ), xQueue( pFM_index->pDev->all( ), *rParameters.getSelected( ), rOpt ) // every replica of the index gets its engines
    {}
    // nullptr = the wrapped reader is exhausted and every read it gave has been handed out (module.h:688-695)
    virtual std::shared_ptr<NucSeq> execute( std::shared_ptr<TP_ARGS>... pArgs ) override
    {
        std::shared_ptr<NucSeq> pQuery;
        detail::Ticket xTicket;
        if( !xQueue.next(
                pQuery, xTicket, [ & ]( ) { return pSource->execute( pArgs... ); },
                []( const std::shared_ptr<NucSeq>& pQ ) { return detail::ReadRef( pQ->xCodes ); } ) )
            return nullptr;
        // a copy carries the ticket: it dies with this read's chain (the device batch's result is recycled then), and it is
        // allocated and freed by THIS thread (PrefetchQueue: why the pulled object itself is not handed out)
        auto pOut = std::make_shared<NucSeq>( *pQuery );
        pOut->xTicket = xTicket;
        return pOut;
    }
    void stats( uint64_t& rBatches, uint64_t& rReads, double& rRunSeconds, double& rPullSeconds )
    {
        xQueue.stats( rBatches, rReads, rRunSeconds, rPullSeconds );
    }
};

class StripOfConsideration : public libMS::Module<SoCPriorityQueue, false, SegmentVector, NucSeq, Pack, FMIndex>
{
    const ma_params xP;

  public:
    StripOfConsideration( const ParameterSetManager& rParameters ) : xP( *rParameters.getSelected( ) )
    {}
    // stripOfConsideration.cpp:162-173: ExtractSeeds here; the sweep runs fused with Harmonization on the device and is
    // materialised only if somebody pops the queue
    virtual std::shared_ptr<SoCPriorityQueue> execute( std::shared_ptr<SegmentVector> pSegments, std::shared_ptr<NucSeq> pQuery,
                                                       std::shared_ptr<Pack>, std::shared_ptr<FMIndex> pFM_index ) override
    {
        auto pRet = std::make_shared<SoCPriorityQueue>( );
        pRet->pSeeds = std::make_shared<Seeds>( );
        pRet->pFmIndex = pFM_index;
        pRet->pQuery = pQuery;
        pRet->xP = xP;
        if( pSegments->xTicket )
        {
            pRet->xTicket = pSegments->xTicket;
            const detail::BatchResult& R = *pRet->xTicket.pResult;
            for( uint64_t i = R.bStages ? R.vSeedOff[ pRet->xTicket.uiRead ] : 0; R.bStages && i < R.vSeedOff[ pRet->xTicket.uiRead + 1 ]; i++ )
                pRet->pSeeds->push_back( detail::toSeed( R.vSeeds[ i ] ) );
            return pRet;
        }
        // segments from elsewhere (e.g. the reference's BinarySeeding): upload them, extract on the device
        detail::SingleRead xB( pFM_index->pDev->p, xP, *pQuery );
        std::vector<ma_segment> vSegs;
        for( const Segment& s : *pSegments )
        {
            ma_segment r;
            r.q_start = (int64_t)s.iStart, r.q_size = (int64_t)s.iSize;
            r.sa_start = s.saStart, r.sa_start_rc = s.saStartRevComp, r.sa_size = s.saSize;
            vSegs.push_back( r );
        }
        const uint64_t off[ 2 ] = { 0, vSegs.size( ) };
        vSegs.push_back( ma_segment( ) );
        maCheck( ma_batch_set_segments( xB.p, off, vSegs.data( ) ) );
        maCheck( ma_extract_seeds_batch( xB.p ) );
        uint64_t nSeeds = 0;
        maCheck( ma_batch_counts( xB.p, nullptr, &nSeeds, nullptr, nullptr, nullptr, nullptr, nullptr ) );
        std::vector<ma_seed> v( nSeeds + 1 );
        uint64_t so[ 2 ];
        maCheck( ma_batch_get_seeds( xB.p, so, v.data( ) ) );
        for( uint64_t i = 0; i < so[ 1 ]; i++ )
            pRet->pSeeds->push_back( detail::toSeed( v[ i ] ) );
        return pRet;
    }
};

class Harmonization : public libMS::Module<SeedsSetVector, false, SoCPriorityQueue, NucSeq, FMIndex>
{
    const ma_params xP;

  public:
    Harmonization( const ParameterSetManager& rParameters ) : xP( *rParameters.getSelected( ) )
    {}
    // harmonization.cpp:374-555 (+ the SoC sweep of stripOfConsideration.cpp:12-161)
    virtual std::shared_ptr<SeedsSetVector> execute( std::shared_ptr<SoCPriorityQueue> pSoCIn, std::shared_ptr<NucSeq> pQuery,
                                                     std::shared_ptr<FMIndex> pFM_index ) override
    {
        auto pRet = std::make_shared<HarmonizedSets>( );
        if( pSoCIn->xTicket )
        {
            pRet->xTicket = pSoCIn->xTicket;
            const detail::BatchResult& R = *pRet->xTicket.pResult;
            if( R.bStages )
                detail::appendHsets( R.vHseedOff, R.vHsetSoc, R.vHseeds, R.vHsetOff[ pRet->xTicket.uiRead ],
                                     R.vHsetOff[ pRet->xTicket.uiRead + 1 ], *pRet );
            return pRet;
        }
        // a queue from elsewhere: its seeds are uploaded, sweep + harmonization run on the device
        if( pSoCIn->pSeeds == nullptr )
            throw std::runtime_error( "Harmonization: the SoC queue carries no seeds" );
        detail::SingleRead xB( pFM_index->pDev->p, xP, *pQuery );
        std::vector<ma_seed> v;
        for( const Seed& s : *pSoCIn->pSeeds )
            v.push_back( detail::fromSeed( s ) );
        const uint64_t off[ 2 ] = { 0, v.size( ) };
        v.push_back( ma_seed( ) );
        maCheck( ma_batch_set_seeds( xB.p, off, v.data( ) ) );
        maCheck( ma_chain_batch( xB.p ) );
        uint64_t nSets = 0, nSeeds = 0;
        maCheck( ma_batch_counts( xB.p, nullptr, nullptr, &nSets, &nSeeds, nullptr, nullptr, nullptr ) );
        std::vector<uint64_t> hoff( 2 ), soff( nSets + 1 );
        std::vector<uint32_t> soc( nSets + 1 );
        std::vector<ma_seed> hs( nSeeds + 1 );
        maCheck( ma_batch_get_hsets( xB.p, hoff.data( ), soff.data( ), soc.data( ), hs.data( ) ) );
        detail::appendHsets( soff, soc, hs, 0, nSets, *pRet );
        return pRet;
    }
};

class NeedlemanWunsch
    : public libMS::Module<libMS::ContainerVector<std::shared_ptr<Alignment>>, false, SeedsSetVector, NucSeq, Pack>
{
    const ma_params xP;

  public:
    NeedlemanWunsch( const ParameterSetManager& rParameters ) : xP( *rParameters.getSelected( ) )
    {}
    // needlemanWunsch.h:111-134
    virtual std::shared_ptr<libMS::ContainerVector<std::shared_ptr<Alignment>>>
    execute( std::shared_ptr<SeedsSetVector> pSeedSets, std::shared_ptr<NucSeq> pQuery, std::shared_ptr<Pack> pPack ) override
    {
        auto pIn = std::dynamic_pointer_cast<HarmonizedSets>( pSeedSets );
        auto pRet = std::make_shared<AlignmentVector>( );
        if( pIn != nullptr && pIn->xTicket )
        {
            pRet->xTicket = pIn->xTicket;
            const detail::BatchResult& R = *pRet->xTicket.pResult;
            if( R.bStages )
                detail::appendAlignments( R.vAlns.data( ), R.vAlnOps.data( ), R.vAlnOff[ pRet->xTicket.uiRead ], R.vAlnOff[ pRet->xTicket.uiRead + 1 ],
                                          false, *pRet );
            return pRet;
        }
        // seed sets from elsewhere: upload, run the DP stage (+ mapping quality, fetched by MappingQuality if it follows)
        detail::SingleRead xB( pPack->pDev->p, xP, *pQuery );
        detail::uploadHsets( xB.p, *pSeedSets );
        maCheck( ma_dp_batch( xB.p ) );
        detail::downloadAlignments( xB.p, false, *pRet );
        pRet->pStandalone = std::make_shared<libMS::ContainerVector<std::shared_ptr<Alignment>>>( );
        detail::downloadAlignments( xB.p, true, *pRet->pStandalone );
        return pRet;
    }
};

class MappingQuality : public libMS::Module<libMS::ContainerVector<std::shared_ptr<Alignment>>, false, NucSeq,
                                            libMS::ContainerVector<std::shared_ptr<Alignment>>>
{
  public:
    MappingQuality( const ParameterSetManager& )
    {}
    // mappingQuality.cpp:11-131: computed on the device together with the DP stage; handed out here
    virtual std::shared_ptr<libMS::ContainerVector<std::shared_ptr<Alignment>>>
    execute( std::shared_ptr<NucSeq>, std::shared_ptr<libMS::ContainerVector<std::shared_ptr<Alignment>>> pAlignments ) override
    {
        auto pIn = std::dynamic_pointer_cast<AlignmentVector>( pAlignments );
        auto pRet = std::make_shared<AlignmentVector>( );
        if( pIn != nullptr && pIn->xTicket )
        {
            pRet->xTicket = pIn->xTicket;
            const detail::BatchResult& R = *pRet->xTicket.pResult;
            detail::appendAlignments( R.vMq.data( ), R.vMqOps.data( ), R.vMqOff[ pRet->xTicket.uiRead ], R.vMqOff[ pRet->xTicket.uiRead + 1 ], true,
                                      *pRet );
            return pRet;
        }
        if( pIn != nullptr && pIn->pStandalone != nullptr )
        {
            for( auto& pA : *pIn->pStandalone )
                pRet->push_back( pA );
            return pRet;
        }
        throw std::runtime_error( "MappingQuality: the alignments were not produced by the MI355X NeedlemanWunsch module "
                                  "(mapping quality is computed on the device together with the DP stage)" );
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

    // Walks one alignment and reports every stretch between two seeds in which the running score falls by at least
    // "Z Drop Inversions" below its maximum (behaviour of smallInversions.h:54-120).  The walk is a small state machine: a
    // seed closes the stretch before it and opens the next one; within a stretch the score follows the global scoring
    // parameters and the drop is measured against the best prefix, less the gap cost of getting back to its diagonal.
    struct DropScan
    {
        const ma_params& rP;
        nucSeqIndex uiQ, uiR; // current position
        nucSeqIndex uiFromQ, uiFromR; // where the open stretch began (end of the last seed)
        nucSeqIndex uiBestQ, uiBestR; // position of the best prefix of the stretch
        int iScore = 0, iBest = std::numeric_limits<int>::min( ), iDrop = 0;
        DropScan( const ma_params& rP, const Alignment& rA )
            : rP( rP ), uiQ( rA.uiBeginOnQuery ), uiR( rA.uiBeginOnRef ), uiFromQ( uiQ ), uiFromR( uiR ), uiBestQ( uiQ ), uiBestR( uiR )
        {}
        bool dropped( ) const
        {
            return iDrop >= (int)(size_t)rP.zdrop_inversion;
        }
        void reopenBehind( nucSeqIndex uiSeedLen ) // the stretch restarts behind the seed that is about to be consumed
        {
            uiFromQ = uiQ + uiSeedLen;
            uiFromR = uiR + uiSeedLen;
            iScore = iDrop = 0;
            iBest = std::numeric_limits<int>::min( );
        }
        void consume( MatchType xType, nucSeqIndex uiLen )
        {
            const bool bOnQ = xType != MatchType::deletion, bOnR = xType != MatchType::insertion;
            if( xType == MatchType::seed || xType == MatchType::match )
                iScore += rP.match * (int)uiLen;
            else if( xType == MatchType::missmatch )
                iScore -= rP.mismatch * (int)uiLen;
            else
                iScore -= rP.gap + rP.extend * (int)uiLen;
            uiQ += bOnQ ? uiLen : 0;
            uiR += bOnR ? uiLen : 0;
            if( iScore >= iBest )
            {
                iBest = iScore;
                uiBestQ = uiQ;
                uiBestR = uiR;
                return;
            }
            const int iAway = (int)std::max( uiQ - uiBestQ, uiR - uiBestR );
            iDrop = std::max( iDrop, iBest - iScore - iAway * rP.extend );
        }
    };
    template <typename F> void forAllDropPos( F&& fDo, const Alignment& rAlignment ) const
    {
        DropScan xScan( xP, rAlignment );
        for( const auto& rSection : rAlignment.data )
        {
            if( rSection.first == MatchType::seed )
            {
                if( xScan.dropped( ) )
                    fDo( xScan.uiFromQ, xScan.uiFromR, xScan.uiQ, xScan.uiR );
                xScan.reopenBehind( rSection.second );
            }
            xScan.consume( rSection.first, rSection.second );
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
    // One candidate = one alignment of each mate.  Key: combined score (times the bonus when the two lie on opposite
    // strands at a plausible insert size), proper pairs first among equal scores (pairedReads.cpp:14-110).
    struct Candidate
    {
        int64_t iScore;
        bool bProper;
        size_t uiFirst, uiSecond;
        bool before( const Candidate& rO ) const
        {
            return iScore != rO.iScore ? iScore > rO.iScore : ( bProper && !rO.bProper );
        }
        bool sameKey( const Candidate& rO ) const
        {
            return iScore == rO.iScore && bProper == rO.bProper;
        }
    };
    Candidate rate( const Alignment& rA, const Alignment& rB, size_t i, size_t j, uint64_t uiN ) const
    {
        Candidate xC{ rA.score( ) + rB.score( ), false, i, j };
        const uint64_t uiF = uiN / 2;
        if( ( rA.beginOnRef( ) >= uiF ) == ( rB.beginOnRef( ) >= uiF ) )
            return xC; // same strand: never a proper Illumina pair
        const nucSeqIndex uiMirrored = uiN - ( rB.beginOnRef( ) + 1 );
        const double fDist = (double)( rA.beginOnRef( ) > uiMirrored ? rA.beginOnRef( ) - uiMirrored : uiMirrored - rA.beginOnRef( ) );
        if( fDist >= (double)mean - std * 3 && fDist <= (double)mean + std * 3 )
        {
            xC.iScore = (int64_t)( xC.iScore * u );
            xC.bProper = true;
        }
        return xC;
    }
    virtual std::shared_ptr<TP_ALIGNMENTS> execute( std::shared_ptr<NucSeq> pQ1, std::shared_ptr<NucSeq> pQ2,
                                                    std::shared_ptr<TP_ALIGNMENTS> pAlignments1,
                                                    std::shared_ptr<TP_ALIGNMENTS> pAlignments2, std::shared_ptr<Pack> pPack ) override
    {
        for( auto& pA : *pAlignments1 )
            pA->xStats.bFirst = true;
        for( auto& pA : *pAlignments2 )
            pA->xStats.bFirst = false;
        if( pAlignments1->empty( ) )
            return pAlignments2;
        if( pAlignments2->empty( ) )
            return pAlignments1;
        const uint64_t uiN = pPack->uiUnpackedSizeForwardPlusReverse( );
        std::vector<Candidate> vCand;
        vCand.reserve( pAlignments1->size( ) * pAlignments2->size( ) );
        for( size_t i = 0; i < pAlignments1->size( ); i++ )
            for( size_t j = 0; j < pAlignments2->size( ); j++ )
                if( ( *pAlignments1 )[ i ]->length( ) != 0 && ( *pAlignments2 )[ j ]->length( ) != 0 )
                    vCand.push_back( rate( *( *pAlignments1 )[ i ], *( *pAlignments2 )[ j ], i, j, uiN ) );
        if( vCand.empty( ) )
            throw std::runtime_error( "PairedReads: no alignment of non-zero length to pair" );
        // the winner and the runner-up's score.  One scan settles it when the best key is unique.  Among several
        // candidates with the best key the reference takes whichever its (unstable) std::sort over all candidates puts
        // first, so only then the candidates are sorted: same sequence, same comparisons, same libstdc++ => same pick.
        size_t uiBest = 0, uiTied = 1;
        for( size_t k = 1; k < vCand.size( ); k++ )
        {
            if( vCand[ k ].before( vCand[ uiBest ] ) )
                uiBest = k, uiTied = 1;
            else if( vCand[ k ].sameKey( vCand[ uiBest ] ) )
                uiTied++;
        }
        Candidate xWin = vCand[ uiBest ];
        int64_t iRunnerUp = xWin.iScore;
        if( uiTied > 1 )
        {
            std::sort( vCand.begin( ), vCand.end( ), []( const Candidate& rX, const Candidate& rY ) { return rX.before( rY ); } );
            xWin = vCand[ 0 ];
        }
        else if( vCand.size( ) > 1 )
        {
            iRunnerUp = std::numeric_limits<int64_t>::min( );
            for( size_t k = 0; k < vCand.size( ); k++ )
                if( k != uiBest )
                    iRunnerUp = std::max( iRunnerUp, vCand[ k ].iScore );
        }
        auto pA = ( *pAlignments1 )[ xWin.uiFirst ];
        auto pB = ( *pAlignments2 )[ xWin.uiSecond ];
        for( auto& pX : { pA, pB } )
            pX->bSecondary = pX->bSupplementary = false;
        pA->xStats.pOther = pB;
        pB->xStats.pOther = pA;
        if( xWin.bProper && vCand.size( ) > 1 )
        {
            // confidence of the pair: lead over the runner-up relative to the winner (single precision like the reference)
            float fConfidence = ( (float)( xWin.iScore - iRunnerUp ) ) / xWin.iScore;
            if( pA->getNumSeeds( ) <= 1 && pB->getNumSeeds( ) <= 1 )
                fConfidence /= 2;
            const bool bStrongA = pA->score( ) >= xP.match * pQ1->length( ) * 0.8 && pAlignments1->size( ) >= 3;
            const bool bStrongB = pB->score( ) >= xP.match * pQ2->length( ) * 0.8 && pAlignments2->size( ) >= 3;
            if( bStrongA || bStrongB )
                fConfidence *= 2;
            pA->fMappingQuality = pB->fMappingQuality = fConfidence > 1 ? 1 : fConfidence;
        }
        auto pRet = std::make_shared<TP_ALIGNMENTS>( );
        pRet->push_back( pA );
        pRet->push_back( pB );
        return pRet;
    }
};

typedef libMS::ContainerVector<std::shared_ptr<NucSeq>> ReadVector;

// The result of one device batch as a FLAT view (ma_engine.h: BatchResult): for read i of the batch the MappingQuality
// records [offsets()[i], offsets()[i+1]) of alignments(), their (type, length) pairs in ops() (ma_alignment::ops_off counts
// pairs).  One header array + one ops array per batch in page-locked memory the engine recycles; Alignment containers are
// built only where somebody asks for them (alignmentsOf) -- building 1.3 M shared_ptr<Alignment> + op vectors per 1 M reads
// was what capped the host-fed path at 3-4 M reads/s.
class AlignedBatch : public libMS::Container
{
  public:
    std::shared_ptr<ReadVector> pReads; // the whole read set ...
    size_t uiFirst = 0; // ... of which this batch is reads [uiFirst, uiFirst + size())
    std::shared_ptr<const detail::BatchResult> pResult;
    size_t size( ) const
    {
        return pResult == nullptr ? 0 : pResult->uiReads;
    }
    const std::shared_ptr<NucSeq>& read( size_t i ) const
    {
        return ( *pReads )[ uiFirst + i ];
    }
    const uint64_t* offsets( ) const
    {
        return pResult->vMqOff.data( );
    }
    const ma_alignment* alignments( ) const
    {
        return pResult->vMq.data( );
    }
    const uint64_t* ops( ) const
    {
        return pResult->vMqOps.data( );
    }
    uint64_t alignedReads( ) const
    {
        return pResult == nullptr ? 0 : pResult->uiAlignedReads;
    }
    std::shared_ptr<AlignmentVector> alignmentsOf( size_t uiRead ) const
    {
        auto pV = std::make_shared<AlignmentVector>( );
        detail::appendAlignments( alignments( ), ops( ), offsets( )[ uiRead ], offsets( )[ uiRead + 1 ], true, *pV );
        return pV;
    }
};

// Phase times of a throughput run (seconds, summed over the device batches; the phases of different batches overlap when
// several are in flight, so their sum can exceed the wall time)
struct AlignerTiming
{
    double fWall = 0, fPack = 0, fH2D = 0, fKernels = 0, fD2H = 0, fContainers = 0;
    uint64_t uiReads = 0, uiBatches = 0, uiAlignedReads = 0;
    std::vector<double> vBatchSeconds; // gather + upload + stages + download of every device batch (one slow one = an engine that allocated)
    // slowest batch over the median batch: a steady-state leg stays below ~3 (the boundary benchmark reports it)
    double maxOverMedian( ) const
    {
        if( vBatchSeconds.empty( ) )
            return 0;
        std::vector<double> v( vBatchSeconds );
        std::sort( v.begin( ), v.end( ) );
        const double fMedian = v[ v.size( ) / 2 ];
        return fMedian > 0 ? v.back( ) / fMedian : 0;
    }
};

// Throughput API, one GPU: reads in host memory -> device batches of uiBatchReads reads, uiInflight of them in flight
// (own stream and host thread each: while one batch is in its kernels the next one is uploaded and the previous one is
// turned into Alignment containers) -> per read the MappingQuality-annotated alignments, in input order.
class BatchAligner
    : public libMS::Module<libMS::ContainerVector<std::shared_ptr<AlignmentVector>>, false, FMIndex,
                           libMS::ContainerVector<std::shared_ptr<NucSeq>>>
{
    ma_params xP;
    ParameterSetManager xParams;
    // engines (stream + device batch + page-locked staging) live as long as the aligner: creating one allocates GBs of
    // device memory and page-locks host memory, which stalls every other stream of the process for 0.3-0.8 s
    mutable std::mutex xEngineMutex;
    mutable std::vector<std::pair<const ma_index*, std::unique_ptr<detail::Engine>>> vIdleEngines;

    std::unique_ptr<detail::Engine> takeEngine( const ma_index* pIndex ) const
    {
        {
            std::lock_guard<std::mutex> xGuard( xEngineMutex );
            for( size_t k = 0; k < vIdleEngines.size( ); k++ )
                if( vIdleEngines[ k ].first == pIndex )
                {
                    auto pEngine = std::move( vIdleEngines[ k ].second );
                    vIdleEngines.erase( vIdleEngines.begin( ) + k );
                    return pEngine;
                }
        }
        return std::unique_ptr<detail::Engine>( new detail::Engine( pIndex, xP ) );
    }
    void giveEngine( const ma_index* pIndex, std::unique_ptr<detail::Engine> pEngine ) const
    {
        std::lock_guard<std::mutex> xGuard( xEngineMutex );
        vIdleEngines.emplace_back( pIndex, std::move( pEngine ) );
    }

  public:
    typedef libMS::ContainerVector<std::shared_ptr<AlignmentVector>> TP_RESULT;
    size_t uiBatchReads = 1u << 18;
    size_t uiInflight = 2;
    // opt-in: the worker threads alignRangeOn starts itself are pinned to the CPUs next to their replica's GPU
    // (ma_host_bind_thread, always inside the mask the caller runs under); off, they inherit the caller's placement
    bool bPinWorkers = false;
    AlignerTiming xLast; // of the last execute()
    std::vector<AlignerTiming> vLastPerIndex; // the same per replica of the index (one entry without replicas)

    BatchAligner( const ParameterSetManager& rParameters ) : xP( *rParameters.getSelected( ) ), xParams( rParameters )
    {}

    typedef libMS::ContainerVector<std::shared_ptr<AlignedBatch>> TP_FLAT;

    // Aligns reads [uiFrom, uiTo) of rQueries on the devices of vIndices (replicas of ONE index; one entry = one GPU, or a
    // "virtual shard" on the same GPU).  The range is cut into device batches of uiBatchReads reads; every replica has uiInflight
    // workers (host thread + engine: stream, device pools, page-locked staging), and all workers take the next batch from ONE
    // counter, so the replicas share the work by batches -- no collective, no inter-GPU traffic.  pFlat: one
    // AlignedBatch per device batch in input order (entry k covers reads [uiFrom + k * uiBatchReads, ...)); pOut: per read its
    // Alignment containers (rOut[ i ]).  pPerIndex: phase times per replica.
    void alignRangeOn( const std::vector<const ma_index*>& vIndices, const libMS::ContainerVector<std::shared_ptr<NucSeq>>& rQueries,
                       size_t uiFrom, size_t uiTo, TP_RESULT* pOut, AlignerTiming& rT, TP_FLAT* pFlat = nullptr,
                       std::shared_ptr<ReadVector> pReadsOfFlat = nullptr, std::vector<AlignerTiming>* pPerIndex = nullptr ) const
    {
        if( vIndices.empty( ) )
            throw std::runtime_error( "BatchAligner: no index" );
        std::mutex xNext;
        size_t uiNext = uiFrom;
        std::string sFailure;
        const size_t uiBatch = std::max<size_t>( uiBatchReads, 1 );
        const size_t uiBatchesAll = ( uiTo - uiFrom + uiBatch - 1 ) / uiBatch;
        const size_t uiFlatBase = pFlat ? pFlat->size( ) : 0;
        if( pFlat )
            pFlat->resize( uiFlatBase + uiBatchesAll );
        if( pPerIndex )
            pPerIndex->assign( vIndices.size( ), AlignerTiming( ) );
        // one device batch [lo, hi) through an engine; results into pFlat / pOut, phase times into rT
        auto runBatch = [ & ]( detail::Engine& xEngine, size_t uiOfIndex, size_t lo, size_t hi ) {
            std::vector<detail::ReadRef> vReads;
            vReads.reserve( hi - lo );
            for( size_t i = lo; i < hi; i++ )
                vReads.emplace_back( rQueries[ i ]->xCodes );
            auto pRes = xEngine.run( vReads, false );
            const auto t0 = std::chrono::steady_clock::now( );
            if( pFlat )
            {
                auto pB = std::make_shared<AlignedBatch>( );
                pB->pReads = pReadsOfFlat;
                pB->uiFirst = lo;
                pB->pResult = pRes;
                ( *pFlat )[ uiFlatBase + ( lo - uiFrom ) / uiBatch ] = pB;
            }
            if( pOut )
                for( size_t i = lo; i < hi; i++ )
                {
                    auto pV = std::make_shared<AlignmentVector>( );
                    detail::appendAlignments( pRes->vMq.data( ), pRes->vMqOps.data( ), pRes->vMqOff[ i - lo ], pRes->vMqOff[ i - lo + 1 ], true, *pV );
                    ( *pOut )[ i ] = pV;
                }
            const double fContainers = detail::secondsSince( t0 );
            std::lock_guard<std::mutex> xGuard( xNext );
            auto account = [ & ]( AlignerTiming& rA ) {
                rA.fPack += pRes->fPack, rA.fH2D += pRes->fH2D, rA.fKernels += pRes->fKernels, rA.fD2H += pRes->fD2H, rA.fContainers += fContainers;
                rA.uiBatches++, rA.uiReads += hi - lo, rA.uiAlignedReads += pRes->uiAlignedReads;
                rA.vBatchSeconds.push_back( pRes->fPack + pRes->fH2D + pRes->fKernels + pRes->fD2H );
            };
            account( rT );
            if( pPerIndex )
                account( ( *pPerIndex )[ uiOfIndex ] );
        };
        // workers: uiInflight per replica, but no more than there are batches (replica-major order would leave the last
        // replicas without a worker when there are few batches: round k holds the k-th worker of every replica)
        std::vector<size_t> vIndexOfWorker;
        for( size_t k = 0; k < std::max<size_t>( uiInflight, 1 ); k++ )
            for( size_t g = 0; g < vIndices.size( ) && vIndexOfWorker.size( ) < std::max<size_t>( uiBatchesAll, 1 ); g++ )
                vIndexOfWorker.push_back( g );
        const size_t uiWorkers = vIndexOfWorker.size( );
        // Engines before admission: every worker's engine exists before the first worker starts, and an engine that has not
        // run a batch of this size yet runs its first one HERE, alone -- the first batch of an engine allocates GBs of device
        // pools and page-locks its staging arrays, which stalls every other stream of the process (0.3-0.8 s each; with four
        // new engines racing inside a timed leg the driver's run of round 3 measured 0.35 M reads/s instead of 19 M).
        std::vector<std::unique_ptr<detail::Engine>> vEngines;
        for( size_t k = 0; k < uiWorkers; k++ )
            vEngines.push_back( takeEngine( vIndices[ vIndexOfWorker[ k ] ] ) );
        try
        {
            for( size_t k = 0; k < uiWorkers; k++ )
                if( !vEngines[ k ]->primed( std::min( uiBatch, uiTo - uiFrom ) ) && uiNext < uiTo )
                {
                    const size_t lo = uiNext, hi = uiNext = std::min( uiTo, uiNext + uiBatch );
                    runBatch( *vEngines[ k ], vIndexOfWorker[ k ], lo, hi );
                }
        }
        catch( const std::exception& rE )
        {
            sFailure = rE.what( );
        }
        // every worker's FIRST batch is handed out here, in worker order: which replica takes a batch must not depend on which
        // thread the scheduler starts first (with few batches a late worker -- and with it a whole replica -- would get none)
        std::vector<std::pair<size_t, size_t>> vFirst( uiWorkers, std::make_pair( uiTo, uiTo ) );
        for( size_t k = 0; k < uiWorkers && uiNext < uiTo && sFailure.empty( ); k++ )
        {
            vFirst[ k ].first = uiNext;
            vFirst[ k ].second = uiNext = std::min( uiTo, uiNext + uiBatch );
        }
        auto worker = [ & ]( size_t uiMe ) {
            try
            {
                // bPinWorkers: threads this call starts itself run on the CPUs next to their replica's GPU (ma_host_bind_thread);
                // worker 0 is the caller's thread and stays where the caller put it
                int iDevice = 0;
                if( bPinWorkers && uiMe != 0 && ma_index_device( vIndices[ vIndexOfWorker[ uiMe ] ], &iDevice ) == 0 )
                    ma_host_bind_thread( iDevice, 0, nullptr );
                detail::Engine& xEngine = *vEngines[ uiMe ];
                if( vFirst[ uiMe ].first < vFirst[ uiMe ].second )
                    runBatch( xEngine, vIndexOfWorker[ uiMe ], vFirst[ uiMe ].first, vFirst[ uiMe ].second );
                for( ;; )
                {
                    size_t lo, hi;
                    {
                        std::lock_guard<std::mutex> xGuard( xNext );
                        if( uiNext >= uiTo || !sFailure.empty( ) )
                            break;
                        lo = uiNext;
                        hi = uiNext = std::min( uiTo, uiNext + uiBatch );
                    }
                    runBatch( xEngine, vIndexOfWorker[ uiMe ], lo, hi );
                }
            }
            catch( const std::exception& rE )
            {
                std::lock_guard<std::mutex> xGuard( xNext );
                if( sFailure.empty( ) )
                    sFailure = rE.what( );
            }
        };
        std::vector<std::thread> vWorkers;
        for( size_t k = 1; k < uiWorkers; k++ )
            vWorkers.emplace_back( worker, k );
        worker( 0 );
        for( auto& rW : vWorkers )
            rW.join( );
        for( size_t k = 0; k < uiWorkers; k++ )
            giveEngine( vIndices[ vIndexOfWorker[ k ] ], std::move( vEngines[ k ] ) );
        if( !sFailure.empty( ) )
            throw std::runtime_error( sFailure );
    }
    void alignRange( const ma_index* pIndex, const libMS::ContainerVector<std::shared_ptr<NucSeq>>& rQueries, size_t uiFrom,
                     size_t uiTo, TP_RESULT* pOut, AlignerTiming& rT, TP_FLAT* pFlat = nullptr,
                     std::shared_ptr<ReadVector> pReadsOfFlat = nullptr ) const
    {
        alignRangeOn( std::vector<const ma_index*>( 1, pIndex ), rQueries, uiFrom, uiTo, pOut, rT, pFlat, pReadsOfFlat );
    }

    // Creates the uiInflight engines of this aligner for pFM_index and runs one batch through each, one after the other
    // (results discarded): after it, execute / executeFlat on batches of up to that size allocate nothing.
    void warmUp( std::shared_ptr<FMIndex> pFM_index, std::shared_ptr<ReadVector> pSample )
    {
        if( pSample == nullptr || pSample->empty( ) )
            return;
        const size_t n = std::min( pSample->size( ), uiBatchReads );
        std::vector<detail::ReadRef> vReads;
        for( size_t i = 0; i < n; i++ )
            vReads.emplace_back( ( *pSample )[ i ]->xCodes );
        for( const ma_index* pIndex : pFM_index->pDev->all( ) ) // every replica of the index (one per GPU)
        {
            std::vector<std::unique_ptr<detail::Engine>> vEngines;
            for( size_t k = 0; k < std::max<size_t>( uiInflight, 1 ); k++ )
                vEngines.push_back( takeEngine( pIndex ) );
            for( auto& pEngine : vEngines )
                if( !pEngine->primed( n ) )
                    pEngine->run( vReads, false );
            for( auto& pEngine : vEngines )
                giveEngine( pIndex, std::move( pEngine ) );
        }
    }

    virtual std::shared_ptr<TP_RESULT> execute( std::shared_ptr<FMIndex> pFM_index,
                                                std::shared_ptr<libMS::ContainerVector<std::shared_ptr<NucSeq>>> pQueries ) override
    {
        auto pRet = std::make_shared<TP_RESULT>( );
        pRet->resize( pQueries->size( ) );
        xLast = AlignerTiming( );
        const auto t0 = std::chrono::steady_clock::now( );
        if( !pQueries->empty( ) ) // on every replica of the index (DeviceIndex::vReplicas: one per GPU)
            alignRangeOn( pFM_index->pDev->all( ), *pQueries, 0, pQueries->size( ), pRet.get( ), xLast, nullptr, nullptr, &vLastPerIndex );
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
        xLast.fWall = detail::secondsSince( t0 );
        return pRet;
    }

    // The same alignment run with the results left FLAT: one AlignedBatch per device batch, in input order (no Alignment
    // container is built; SmallInversions needs containers and is not applied here).
    std::shared_ptr<TP_FLAT> executeFlat( std::shared_ptr<FMIndex> pFM_index, std::shared_ptr<ReadVector> pQueries )
    {
        auto pRet = std::make_shared<TP_FLAT>( );
        xLast = AlignerTiming( );
        const auto t0 = std::chrono::steady_clock::now( );
        if( !pQueries->empty( ) )
            alignRangeOn( pFM_index->pDev->all( ), *pQueries, 0, pQueries->size( ), nullptr, xLast, pRet.get( ), pQueries, &vLastPerIndex );
        xLast.fWall = detail::secondsSince( t0 );
        return pRet;
    }

    // Paired mode (setUpCompGraphPaired, export.cpp:130-202) for a batch: vMates holds the mates of pair k at 2k and
    // 2k + 1; both mates of all pairs go through the device together, PairedReads then picks per pair on the host.
    std::shared_ptr<TP_RESULT> executePaired( std::shared_ptr<FMIndex> pFM_index,
                                              std::shared_ptr<libMS::ContainerVector<std::shared_ptr<NucSeq>>> vMates )
    {
        if( vMates->size( ) % 2 )
            throw std::runtime_error( "BatchAligner::executePaired: odd number of reads" );
        auto pPerRead = execute( pFM_index, vMates );
        auto pPack = std::make_shared<Pack>( );
        pPack->pDev = pFM_index->pDev;
        PairedReads xPairedReads( xParams );
        auto pRet = std::make_shared<TP_RESULT>( );
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

// Throughput API, several GPUs of one node: reads are independent, the index is replicated, the device batches
// of a read set rotate over the replicas (BatchAligner::alignRangeOn: uiInflight workers per replica, one shared batch
// counter), and the results land at the reads' input positions.  No collective, no inter-GPU traffic.  The aligner and its
// engines PERSIST: a second execute( ) / executeFlat( ) creates no engine, allocates nothing and page-locks nothing (round 4
// built a new BatchAligner per replica inside every execute( )).  Replicas may live on the same device (tests on a one-GPU
// box use "virtual shards" on device 0).
class MultiDeviceAligner
{
    ParameterSetManager xParams;
    std::vector<std::shared_ptr<FMIndex>> vReplicas;
    std::vector<const ma_index*> vIndices;
    BatchAligner xAligner;

    void configure( )
    {
        xAligner.uiBatchReads = uiBatchReads;
        xAligner.uiInflight = uiInflight;
    }

  public:
    size_t uiBatchReads = 1u << 18;
    size_t uiInflight = 2; // device batches in flight PER replica
    std::vector<AlignerTiming> vLast; // per replica, of the last execute()
    AlignerTiming xLast; // all replicas together

    MultiDeviceAligner( const ParameterSetManager& rParameters, const std::vector<std::shared_ptr<FMIndex>>& vReplicas )
        : xParams( rParameters ), vReplicas( vReplicas ), xAligner( rParameters )
    {
        if( vReplicas.empty( ) )
            throw std::runtime_error( "MultiDeviceAligner: no index replica" );
        for( const auto& pFM : vReplicas )
            for( const ma_index* pIndex : pFM->pDev->all( ) )
                vIndices.push_back( pIndex );
    }

    // One replica of pFM's index on every device of vDevices, each as an FMIndex of its own (a device that already holds
    // pFM's index reuses it).  (replicateIndex attaches replicas to ONE FMIndex instead: what the graph forms use.)
    static std::vector<std::shared_ptr<FMIndex>> replicate( const std::shared_ptr<FMIndex>& pFM, const std::vector<int>& vDevices,
                                                            int iDeviceOfOriginal = 0 )
    {
        std::vector<int> vOthers;
        bool bUsedOriginal = false;
        for( int iDev : vDevices )
        {
            if( iDev == iDeviceOfOriginal && !bUsedOriginal )
                bUsedOriginal = true;
            else
                vOthers.push_back( iDev );
        }
        // copies that are NOT attached to pFM: this aligner addresses every replica itself
        auto pScratch = std::make_shared<DeviceIndex>( );
        pScratch->p = pFM->pDev->p;
        std::vector<std::shared_ptr<DeviceIndex>> vCopies;
        try
        {
            vCopies = replicateIndex( pScratch, vOthers );
        }
        catch( ... )
        {
            pScratch->p = nullptr;
            throw;
        }
        pScratch->p = nullptr; // (borrowed)
        pScratch->vReplicas.clear( );
        std::vector<std::shared_ptr<FMIndex>> vRet;
        size_t uiCopy = 0;
        bUsedOriginal = false;
        for( int iDev : vDevices )
        {
            if( iDev == iDeviceOfOriginal && !bUsedOriginal )
            {
                vRet.push_back( pFM );
                bUsedOriginal = true;
                continue;
            }
            auto pCopy = std::make_shared<FMIndex>( );
            pCopy->pDev = vCopies[ uiCopy++ ];
            vRet.push_back( pCopy );
        }
        return vRet;
    }

    // engines of every replica created and primed with a batch of pSample (after it nothing is allocated inside execute)
    void warmUp( std::shared_ptr<ReadVector> pSample )
    {
        configure( );
        for( const auto& pFM : vReplicas )
            xAligner.warmUp( pFM, pSample );
    }

    std::shared_ptr<BatchAligner::TP_RESULT> execute( std::shared_ptr<libMS::ContainerVector<std::shared_ptr<NucSeq>>> pQueries )
    {
        configure( );
        auto pRet = std::make_shared<BatchAligner::TP_RESULT>( );
        pRet->resize( pQueries->size( ) );
        xLast = AlignerTiming( );
        vLast.assign( vIndices.size( ), AlignerTiming( ) );
        const auto t0 = std::chrono::steady_clock::now( );
        if( !pQueries->empty( ) )
            xAligner.alignRangeOn( vIndices, *pQueries, 0, pQueries->size( ), pRet.get( ), xLast, nullptr, nullptr, &vLast );
        xLast.fWall = detail::secondsSince( t0 );
        for( auto& rT : vLast )
            rT.fWall = xLast.fWall;
        return pRet;
    }

    // The same run with the results left FLAT: one AlignedBatch per device batch, in input order, whichever replica ran it.
    std::shared_ptr<BatchAligner::TP_FLAT> executeFlat( std::shared_ptr<ReadVector> pQueries )
    {
        configure( );
        auto pRet = std::make_shared<BatchAligner::TP_FLAT>( );
        xLast = AlignerTiming( );
        vLast.assign( vIndices.size( ), AlignerTiming( ) );
        const auto t0 = std::chrono::steady_clock::now( );
        if( !pQueries->empty( ) )
            xAligner.alignRangeOn( vIndices, *pQueries, 0, pQueries->size( ), nullptr, xLast, pRet.get( ), pQueries, &vLast );
        xLast.fWall = detail::secondsSince( t0 );
        for( auto& rT : vLast )
            rT.fWall = xLast.fWall;
        return pRet;
    }
};
} // namespace libMA
