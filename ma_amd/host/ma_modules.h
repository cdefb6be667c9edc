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
#include <cstdint>
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
        else
            throw std::runtime_error( "The presetting '" + sKey + "' can not be found." );
    }
    const ma_params* getSelected( ) const
    {
        return &xSelected;
    }
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

class SuffixArrayInterface : public libMS::Container // fMIndex.h:155-175
{
  public:
    std::shared_ptr<DeviceIndex> pDev;
};
class FMIndex : public SuffixArrayInterface
{
  public:
    uint64_t getRefSeqLength( ) const
    {
        uint64_t n = 0;
        maCheck( ma_index_sizes( pDev->p, nullptr, nullptr, &n, nullptr ) );
        return n;
    }
};
class Pack : public libMS::Container
{
  public:
    std::shared_ptr<DeviceIndex> pDev;
    std::vector<std::string> vNames;
    std::vector<uint64_t> vStarts, vLengths;
    uint64_t uiUnpackedSizeForwardPlusReverse( ) const
    {
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
    const uint64_t nWords = ( bwt.size( ) - 40 ) / 4, nSa = ( seqLen + 32 ) / 32;
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
        lens.push_back( c->length( ) );
        cat.insert( cat.end( ), c->xCodes.begin( ), c->xCodes.end( ) );
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
class Alignment : public libMS::Container // alignment.h:55-84
{
  public:
    std::vector<std::pair<MatchType, nucSeqIndex>> data;
    nucSeqIndex uiBeginOnRef = 0, uiEndOnRef = 0, uiBeginOnQuery = 0, uiEndOnQuery = 0;
    int64_t iScore = 0;
    double fMappingQuality = NAN;
    unsigned int index_of_strip = 0;
    bool bSecondary = false, bSupplementary = false;
    int64_t score( ) const
    {
        return iScore;
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

// Throughput API: a whole batch of reads through all stages in one go.
class BatchAligner
    : public libMS::Module<libMS::ContainerVector<std::shared_ptr<AlignmentVector>>, false, FMIndex,
                           libMS::ContainerVector<std::shared_ptr<NucSeq>>>
{
    ma_params xP;

  public:
    BatchAligner( const ParameterSetManager& rParameters ) : xP( *rParameters.getSelected( ) )
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
        return pRet;
    }
};
} // namespace libMA
