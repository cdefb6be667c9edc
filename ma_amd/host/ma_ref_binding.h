// ma_ref_binding.h -- the MI355X seed-and-extend path as graph nodes on the REAL types of ITBE-Lab/ma.
//
// This header includes only the reference's own headers (libs/{ms,ma,util}/inc, where they lie in the reference tree),
// the C ABI (include/ma_amd.h) and the reference-free batcher (ma_engine.h).  It is what a maintainer of MA compiles
// inside the reference tree: the classes below derive from the reference's libMS::Module<> with the template signatures
// of the modules they replace and exchange the reference's own containers, so each of them can stand in
// libMA::setUpCompGraph (libs/ma/src/util/export.cpp:84-126) next to the reference's FileReader / FileWriter / CPU
// modules -- all five together (the reads of all graph threads funnel into device batches, ma_engine.h) or any ONE of
// them between the reference's CPU modules (its input container is uploaded, its stage runs on the GPU).
//
//   ma_amd::BinarySeeding        : libMS::Module<libMA::SegmentVector, false, libMA::SuffixArrayInterface, libMA::NucSeq>
//                                                                           (binarySeeding.h:26, execute binarySeeding.cpp:86-178)
//   ma_amd::StripOfConsideration : libMS::Module<libMA::SoCPriorityQueue, false, libMA::SegmentVector, libMA::NucSeq,
//                                                libMA::Pack, libMA::FMIndex>   (stripOfConsideration.h:164, .cpp:162-173)
//   ma_amd::Harmonization        : libMS::Module<ContainerVector<shared_ptr<Seeds>>, false, libMA::SoCPriorityQueue,
//                                                libMA::NucSeq, libMA::FMIndex> (harmonization.h:34-35, .cpp:374-555)
//   ma_amd::NeedlemanWunsch      : libMS::Module<ContainerVector<shared_ptr<Alignment>>, false,
//                                                ContainerVector<shared_ptr<Seeds>>, libMA::NucSeq, libMA::Pack>
//                                                                           (needlemanWunsch.h:51-52,111-134)
//   ma_amd::MappingQuality       : libMS::Module<ContainerVector<shared_ptr<Alignment>>, false, libMA::NucSeq,
//                                                ContainerVector<shared_ptr<Alignment>>> (mappingQuality.h:22-23, .cpp:11-131)
//
// Nothing here computes: every stage runs behind include/ma_amd.h on the GPU; a non-zero status becomes
// std::runtime_error, which BasePledge::simultaneousGet collects and rethrows (module.h:339-377).  There is no CPU
// fallback: without a HIP device the first execute() throws.
//
// (ma_modules.h / ms_graph.h in this directory are the reference-FREE mirror of the same API -- look-alike containers in
// the reference's namespaces for builds that do not have the reference tree; the two cannot share a translation unit.)
#pragma once
#include "ma/container/alignment.h"
#include "ma/module/fileWriter.h"
#include "ma/container/fMIndex.h"
#include "ma/container/nucSeq.h"
#include "ma/container/pack.h"
#include "ma/container/seed.h"
#include "ma/container/segment.h"
#include "ma/container/soc.h"
#include "ms/container/container.h"
#include "ms/module/module.h"
#include "ms/util/parameter.h"

#include "ma_amd.h"
#include "ma_engine.h"

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace ma_amd
{
typedef libMS::ContainerVector<std::shared_ptr<libMA::Seeds>> SeedSets;
typedef libMS::ContainerVector<std::shared_ptr<libMA::Alignment>> Alignments;

inline void check( int rc )
{
    if( rc != 0 )
        throw std::runtime_error( ma_last_error( ) );
}

// Process-wide options of the binding (read when a module is constructed).
struct BindingOptions
{
    // RANSAC draws (sac_model_line.cpp:55,63 use libc's rand()): the device restates glibc's generator and starts every
    // read's Harmonization from the state srand( uiRansacSeed ) leaves -- the reference is only reproducible when its
    // caller does the same
    unsigned int uiRansacSeed = 1;
    engine::BatcherOptions xBatcher; // funnel of the per-read execute() calls (ma_engine.h)
    engine::PrefetchOptions xPrefetch; // PrefetchReader: reads pulled ahead of the graph threads, a device batch at a time
};
inline BindingOptions& options( )
{
    static BindingOptions xOptions;
    return xOptions;
}

// The parameters the path reads, out of the reference's ParameterSetManager (parameter.h:521-1060; it and pGlobalParams live in the global namespace).  Settings the device
// path does not implement are refused instead of being ignored.
inline ma_params paramsOf( const ::ParameterSetManager& rParameters )
{
    const auto pSel = rParameters.getSelected( );
    ma_params P;
    ma_params_default( &P );
    const std::string sTechnique = pSel->xSeedingTechnique->get( );
    if( sTechnique == "maxSpan" )
        P.seeding_technique = 0;
    else if( sTechnique == "SMEMs" )
        P.seeding_technique = 1;
    else if( sTechnique == "MEMs" )
        P.seeding_technique = 2;
    else
        throw std::runtime_error( "ma_amd: unknown seeding technique '" + sTechnique + "'" );
    // ("Skip ambiguous seeds" is not read: emplaceAllEachSeeds passes bSkip = true whatever the flag says, segment.h:365)
    if( !pSel->xRectangularSoc->get( ) )
        throw std::runtime_error( "ma_amd: 'Rectangular SoC' = false is not implemented on the device" );
    if( pSel->xDisableGapCostEstimationCutting->get( ) )
        throw std::runtime_error( "ma_amd: 'Pick Local Seed Set A' disabled is not implemented on the device" );
    if( !pSel->xOptimisticGapCostEstimation->get( ) )
        throw std::runtime_error( "ma_amd: pessimistic gap cost estimation is not implemented on the device" );
    P.min_seed_len = pSel->xMinSeedLength->get( );
    P.min_ambiguity = pSel->xMinimalSeedAmbiguity->get( );
    P.max_ambiguity = pSel->xMaximalSeedAmbiguity->get( );
    P.min_seed_size_drop = pSel->xMinimalSeedSizeDrop->get( );
    P.rel_min_seed_size_amount = pSel->xRelMinSeedSizeAmount->get( );
    P.max_num_soc = pSel->xMaxNumSoC->get( );
    P.min_num_soc = pSel->xMinNumSoC->get( );
    P.soc_width = pSel->xSoCWidth->get( );
    P.harm_score_min = pSel->xHarmScoreMin->get( );
    P.harm_score_min_rel = pSel->xHarmScoreMinRel->get( );
    P.soc_score_decrease_tol = pSel->xSoCScoreDecreaseTolerance->get( );
    P.score_diff_tol = pSel->xScoreDiffTolerance->get( );
    P.max_score_lookahead = pSel->xMaxScoreLookahead->get( );
    P.switch_qlen = pSel->xSwitchQlen->get( );
    P.max_delta_dist = pSel->xMaxDeltaDist->get( );
    P.min_delta_dist = pSel->xMinDeltaDist->get( );
    P.max_gap_area = pSel->xMaxGapArea->get( );
    P.genome_size_disable = (uint64_t)pSel->xGenomeSizeDisable->get( );
    P.disable_heuristics = pSel->xDisableHeuristics->get( ) ? 1 : 0;
    P.padding = pSel->xPadding->get( );
    P.bandwidth_ext = pSel->xBandwidthDPExtension->get( );
    P.min_bandwidth_gap = pSel->xMinBandwidthGapFilling->get( );
    P.zdrop = pSel->xZDrop->get( );
    P.min_alignment_score = pSel->xMinAlignmentScore->get( );
    P.report_n_best = pSel->xReportN->get( );
    P.max_supplementary = pSel->xMaxSupplementaryPerPrim->get( );
    P.max_overlap_supplementary = pSel->xMaxOverlapSupplementary->get( );
    P.search_inversions = pSel->xSearchInversions->get( ) ? 1 : 0;
    P.zdrop_inversion = pSel->xZDropInversion->get( );
    P.use_paired_reads = pSel->xUsePairedReads->get( ) ? 1 : 0;
    P.mean_paired_dist = pSel->xMeanPairedReadDistance->get( );
    P.std_paired_dist = pSel->xStdPairedReadDistance->get( );
    P.paired_bonus = pSel->xPairedBonus->get( );
    P.match = ::pGlobalParams->iMatch->get( );
    P.mismatch = ::pGlobalParams->iMissMatch->get( );
    P.gap = ::pGlobalParams->iGap->get( );
    P.extend = ::pGlobalParams->iExtend->get( );
    P.gap2 = ::pGlobalParams->iGap2->get( );
    P.extend2 = ::pGlobalParams->iExtend2->get( );
    P.sv_penalty = ::pGlobalParams->uiSVPenalty->get( );
    P.srand_seed = options( ).uiRansacSeed;
    return P;
}

// ---- the device-resident copy of a (Pack, FMIndex) pair ----------------------------------------------------------------
// vReplicas: copies of the same index on other devices of the node (replicateIndex); PrefetchReader rotates its device batches
// over p and the replicas, everything else uses p.
struct DeviceIndex
{
    ma_index* p = nullptr;
    std::vector<std::shared_ptr<DeviceIndex>> vReplicas;
    DeviceIndex( )
    {}
    DeviceIndex( const DeviceIndex& ) = delete;
    ~DeviceIndex( )
    {
        if( p != nullptr )
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

namespace detail
{
// FMIndex keeps its arrays protected (fMIndex.h:210-230); a pointer to member named through a derived class is the
// standard-conforming way to read them without touching the reference's header
struct FMIndexAccess : public libMA::FMIndex
{
    static const std::vector<uint32_t>& bwtOf( const libMA::FMIndex& r )
    {
        return r.*( &FMIndexAccess::bwt );
    }
    static const std::vector<int64_t>& saOf( const libMA::FMIndex& r )
    {
        return r.*( &FMIndexAccess::sa );
    }
    static int saIntervalOf( const libMA::FMIndex& r )
    {
        return r.*( &FMIndexAccess::sa_intv );
    }
};

// One registered container: the device copy it maps to and a weak reference to the container itself.  The registry is
// keyed by raw addresses; the weak reference tells a LIVE container from a new one that the allocator happened to put at
// the address of a genome that was unloaded without detachIndex (whose entry is then dropped instead of being served).
struct RegistryEntry
{
    std::shared_ptr<DeviceIndex> pDev;
    std::weak_ptr<const void> pOwner;
    uint64_t uiForwardLength = 0; // of the pair the device copy was made from (Pack: uiUnpackedSizeForwardStrand)
};
struct Registry
{
    std::mutex xMutex;
    std::map<const void*, RegistryEntry> xByObject; // keyed by the Pack's AND the FMIndex's address
    // entry of a live container, or nullptr (an entry whose container died is erased: its device copy goes with the last module
    // that still holds it)
    RegistryEntry* find( const void* pKey )
    {
        auto xIt = xByObject.find( pKey );
        if( xIt == xByObject.end( ) )
            return nullptr;
        if( xIt->second.pOwner.expired( ) )
        {
            xByObject.erase( xIt );
            return nullptr;
        }
        return &xIt->second;
    }
};
inline Registry& registry( )
{
    static Registry xRegistry;
    return xRegistry;
}
} // namespace detail

// Uploads the reference's index to the GPU (once per pair; replaces nothing in the reference: its Pack / FMIndex stay
// usable by its CPU modules).  Call it where the genome is loaded (GenomeManager::loadGenome, execution-context.h:60-93),
// or let the first module that sees both containers do it (StripOfConsideration::execute).  BinarySeeding sees the
// FMIndex only and therefore needs the pair attached beforehand.
inline std::shared_ptr<DeviceIndex> attachIndex( const std::shared_ptr<libMA::Pack>& pPack, const std::shared_ptr<libMA::FMIndex>& pFM )
{
    if( pPack == nullptr || pFM == nullptr )
        throw std::runtime_error( "ma_amd::attachIndex: null container" );
    const uint64_t uiN = pFM->getRefSeqLength( ), uiF = pPack->uiUnpackedSizeForwardStrand;
    if( uiN != 2 * uiF ) // on the cached path too: any Pack could be bound to an FMIndex that is registered already
        throw std::runtime_error( "ma_amd::attachIndex: Pack and FMIndex do not belong together" );
    auto& rReg = detail::registry( );
    std::lock_guard<std::mutex> xGuard( rReg.xMutex );
    if( detail::RegistryEntry* pHave = rReg.find( pFM.get( ) ) )
    {
        if( pHave->uiForwardLength != uiF )
            throw std::runtime_error( "ma_amd::attachIndex: this FMIndex is attached with another Pack" );
        detail::RegistryEntry xForPack = *pHave;
        xForPack.pOwner = std::shared_ptr<const void>( pPack, pPack.get( ) );
        rReg.xByObject[ pPack.get( ) ] = xForPack;
        return xForPack.pDev;
    }
    const std::vector<uint32_t>& rBwt = detail::FMIndexAccess::bwtOf( *pFM );
    const std::vector<int64_t>& rSa = detail::FMIndexAccess::saOf( *pFM );
    if( detail::FMIndexAccess::saIntervalOf( *pFM ) != 32 )
        throw std::runtime_error( "ma_amd::attachIndex: suffix array sampling interval must be 32" );
    uint64_t aL2[ 5 ];
    for( int i = 0; i < 5; i++ )
        aL2[ i ] = pFM->L2[ i ];
    // the forward strand, 2 bit per base MSB first (Pack::getNucleotideOnPos, pack.h:163-168: its vector is private)
    std::vector<uint8_t> vPac( ( uiF + 3 ) / 4, 0 );
    for( uint64_t i = 0; i < uiF; i++ )
        vPac[ i >> 2 ] |= (uint8_t)( pPack->getNucleotideOnPos( i ) << ( ( ~i & 3 ) << 1 ) );
    std::vector<uint64_t> vStarts, vLens;
    for( const auto& rSeq : pPack->xVectorOfSequenceDescriptors )
    {
        vStarts.push_back( rSeq.uiStartOffsetUnpacked );
        vLens.push_back( rSeq.uiLengthUnpacked );
    }
    auto pDev = std::make_shared<DeviceIndex>( );
    check( ma_index_create( rBwt.data( ), rBwt.size( ), rSa.data( ), rSa.size( ), aL2, pFM->primary, uiN, vPac.data( ),
                            (int32_t)vStarts.size( ), vStarts.data( ), vLens.data( ), &pDev->p ) );
    detail::RegistryEntry xEntry;
    xEntry.pDev = pDev;
    xEntry.uiForwardLength = uiF;
    xEntry.pOwner = std::shared_ptr<const void>( pFM, pFM.get( ) );
    rReg.xByObject[ pFM.get( ) ] = xEntry;
    xEntry.pOwner = std::shared_ptr<const void>( pPack, pPack.get( ) );
    rReg.xByObject[ pPack.get( ) ] = xEntry;
    return pDev;
}
// One more copy of an attached index on every device of vDevices (one process drives all GPUs of the node; the
// reference's graph copies use the whole node the same way, export.cpp:99-126, module.h:303-369).  Call it after attachIndex,
// before the PrefetchReader is constructed:
//     ma_amd::replicateIndex( ma_amd::attachIndex( pPack, pFMIndex ), { 1, 2, 3, 4, 5, 6, 7 } );
inline void replicateIndex( const std::shared_ptr<DeviceIndex>& pDev, const std::vector<int>& vDevices )
{
    for( ma_index* pCopy : engine::replicateOnDevices( pDev->p, vDevices ) )
    {
        auto pNew = std::make_shared<DeviceIndex>( );
        pNew->p = pCopy;
        pDev->vReplicas.push_back( pNew );
    }
}
// Frees the device copy once the last module that holds it lets go (call when the genome is unloaded; a genome that is
// unloaded without it is dropped from the registry the next time its address is looked up or reused).
inline void detachIndex( const std::shared_ptr<libMA::Pack>& pPack, const std::shared_ptr<libMA::FMIndex>& pFM )
{
    auto& rReg = detail::registry( );
    std::lock_guard<std::mutex> xGuard( rReg.xMutex );
    rReg.xByObject.erase( pPack.get( ) );
    rReg.xByObject.erase( pFM.get( ) );
}
inline std::shared_ptr<DeviceIndex> deviceIndexOf( const void* pPackOrFMIndex )
{
    auto& rReg = detail::registry( );
    std::lock_guard<std::mutex> xGuard( rReg.xMutex );
    detail::RegistryEntry* pHave = rReg.find( pPackOrFMIndex );
    if( pHave == nullptr )
        throw std::runtime_error( "ma_amd: this index was not uploaded; call ma_amd::attachIndex( pPack, pFMIndex ) where the "
                                  "genome is loaded" );
    return pHave->pDev;
}

// ---- containers: the reference's own, plus the ticket of the device batch the read went through -------------------------
// A ticket says "every stage of this read already ran on the GPU, as read uiRead of that batch": the next MI355X module
// only picks its records.  The containers ARE the reference's (filled completely when BatcherOptions::bStages is set), so
// a CPU module of the reference can consume them just as well.
struct TicketedSegments : public libMA::SegmentVector
{
    engine::Ticket xTicket;
};
// a query PrefetchReader hands out: the read went through all stages already, as read uiRead of that device batch
struct TicketedQuery : public libMA::NucSeq
{
    engine::Ticket xTicket;
    explicit TicketedQuery( const libMA::NucSeq& rOther )
    {
        sName = rOther.sName;
        iId = rOther.iId;
#if WITH_QUALITY
        if( rOther.bHasQuality( ) )
        {
            addQuality( );
            vAppend( rOther.pGetSequenceRef( ), rOther.pxQualityRef, rOther.length( ) );
        }
        else
#endif
            vAppend( rOther.pGetSequenceRef( ), rOther.length( ) );
    }
};
struct TicketedSoCs : public libMA::SoCPriorityQueue
{
    engine::Ticket xTicket;
    explicit TicketedSoCs( std::shared_ptr<libMA::Seeds> pSeeds ) : libMA::SoCPriorityQueue( pSeeds )
    {}
};
struct TicketedSeedSets : public SeedSets
{
    engine::Ticket xTicket;
};
struct TicketedAlignments : public Alignments
{
    engine::Ticket xTicket;
    // stage-at-a-time path: the MappingQuality selection of the same device run (computed together with the DP stage)
    std::shared_ptr<Alignments> pWithQuality;
};

namespace detail
{
inline libMA::Seed toSeed( const ma_seed& r )
{
    libMA::Seed xSeed( (libMA::nucSeqIndex)r.q_start, (libMA::nucSeqIndex)r.len, (libMA::nucSeqIndex)r.r_start, r.ambiguity,
                       r.on_forward != 0 );
    xSeed.uiDelta = (libMA::nucSeqIndex)r.delta;
    return xSeed;
}
inline ma_seed fromSeed( const libMA::Seed& rSeed )
{
    ma_seed r;
    r.q_start = (int64_t)rSeed.start( ), r.len = (int64_t)rSeed.size( ), r.r_start = (int64_t)rSeed.start_ref( );
    r.delta = (int64_t)rSeed.uiDelta, r.ambiguity = rSeed.uiAmbiguity, r.on_forward = rSeed.bOnForwStrand ? 1 : 0;
    return r;
}
// a device batch of exactly one read: the stage-at-a-time path
struct SingleRead
{
    ma_batch* p = nullptr;
    SingleRead( const ma_index* pIndex, const ma_params& rP, const libMA::NucSeq& rQuery )
    {
        check( ma_batch_create( pIndex, &rP, 1, rQuery.length( ) + 64, &p ) );
        const uint64_t aOff[ 2 ] = { 0, rQuery.length( ) };
        const uint8_t uiDummy = 0;
        if( ma_batch_set_reads( p, rQuery.length( ) ? rQuery.pGetSequenceRef( ) : &uiDummy, aOff, 1 ) != 0 )
        {
            ma_batch_destroy( p );
            throw std::runtime_error( ma_last_error( ) );
        }
    }
    SingleRead( const SingleRead& ) = delete;
    ~SingleRead( )
    {
        ma_batch_destroy( p );
    }
};
// SoCPriorityQueue (soc.h:96-420) out of the device's records: pSeeds as rectangularSoC leaves them, vMaxima as
// make_heap + rectangularSoC leave it -- the reference's own pop() then yields the reference's order
inline std::shared_ptr<TicketedSoCs> makeQueue( const ma_soc* pHeap, size_t uiStrips, const ma_seed* pSorted, size_t uiSeeds,
                                                const libMA::NucSeq& rQuery )
{
    auto pSeeds = std::make_shared<libMA::Seeds>( );
    pSeeds->xStats.sName = rQuery.sName; // ExtractSeeds::execute (stripOfConsideration.h:147)
    pSeeds->reserve( uiSeeds );
    for( size_t i = 0; i < uiSeeds; i++ )
        pSeeds->push_back( toSeed( pSorted[ i ] ) );
    auto pQueue = std::make_shared<TicketedSoCs>( pSeeds );
    pQueue->vMaxima.reserve( uiStrips );
    for( size_t k = 0; k < uiStrips; k++ )
    {
        libMA::SoCOrder xOrder;
        xOrder.uiAccumulativeLength = (libMA::nucSeqIndex)pHeap[ k ].acc_len;
        xOrder.uiSeedAmbiguity = pHeap[ k ].ambiguity;
        xOrder.uiSeedAmount = pHeap[ k ].n_seeds;
        pQueue->vMaxima.emplace_back( xOrder, pSeeds->begin( ) + pHeap[ k ].begin, pSeeds->begin( ) + pHeap[ k ].end );
    }
    return pQueue;
}
// alignment records [uiFrom, uiTo) of a download -> the reference's Alignment containers
inline void appendAlignments( const ma_alignment* vAlns, const uint64_t* vOps, uint64_t uiFrom, uint64_t uiTo,
                              bool bQuality, const libMA::NucSeq& rQuery, Alignments& rOut )
{
    for( uint64_t i = uiFrom; i < uiTo; i++ )
    {
        const ma_alignment& r = vAlns[ i ];
        auto pA = std::make_shared<libMA::Alignment>( (libMA::nucSeqIndex)r.begin_ref, (libMA::nucSeqIndex)r.begin_q,
                                                      (libMA::nucSeqIndex)r.end_ref, (libMA::nucSeqIndex)r.end_q );
        pA->iScore = r.score;
        pA->xStats.index_of_strip = r.soc_index;
        pA->xStats.sName = rQuery.sName; // NeedlemanWunsch::execute_one (needlemanWunsch.cpp:640-642)
        pA->bSecondary = r.secondary != 0;
        pA->bSupplementary = r.supplementary != 0;
        pA->fMappingQuality = bQuality ? r.mapq : NAN;
        pA->data.reserve( r.n_ops );
        libMA::nucSeqIndex uiLength = 0;
        for( uint32_t k = 0; k < r.n_ops; k++ )
        {
            pA->data.emplace_back( (libMA::MatchType)vOps[ 2 * ( r.ops_off + k ) ], (libMA::nucSeqIndex)vOps[ 2 * ( r.ops_off + k ) + 1 ] );
            uiLength += (libMA::nucSeqIndex)vOps[ 2 * ( r.ops_off + k ) + 1 ];
        }
        pA->uiLength = uiLength;
        rOut.push_back( pA );
    }
}
inline void downloadAlignments( ma_batch* pBatch, bool bQuality, const libMA::NucSeq& rQuery, Alignments& rOut )
{
    uint64_t nAln = 0, nOps = 0;
    check( ma_batch_counts( pBatch, nullptr, nullptr, nullptr, nullptr, &nAln, &nOps, nullptr ) );
    std::vector<uint64_t> vOff( 2 ), vOps( 2 * nOps + 2 );
    std::vector<ma_alignment> vAlns( nAln + 1 );
    check( ( bQuality ? ma_batch_get_mapq_alignments : ma_batch_get_alignments )( pBatch, vOff.data( ), vAlns.data( ), vOps.data( ) ) );
    appendAlignments( vAlns.data( ), vOps.data( ), 0, vOff[ 1 ], bQuality, rQuery, rOut );
}
inline void appendSeedSets( const std::vector<uint64_t>& vSeedOff, const std::vector<uint32_t>& vSoc, const std::vector<ma_seed>& vSeeds,
                            uint64_t uiFrom, uint64_t uiTo, const libMA::NucSeq& rQuery, SeedSets& rOut )
{
    for( uint64_t h = uiFrom; h < uiTo; h++ )
    {
        auto pSet = std::make_shared<libMA::Seeds>( );
        pSet->xStats.sName = rQuery.sName;
        pSet->xStats.index_of_strip = vSoc[ h ];
        pSet->bConsistent = true; // harmonization.cpp:329
        pSet->reserve( vSeedOff[ h + 1 ] - vSeedOff[ h ] );
        for( uint64_t i = vSeedOff[ h ]; i < vSeedOff[ h + 1 ]; i++ )
            pSet->push_back( toSeed( vSeeds[ i ] ) );
        rOut.push_back( pSet );
    }
}
} // namespace detail

// ---- BinarySeeding --------------------------------------------------------------------------------------------------
class BinarySeeding : public libMS::Module<libMA::SegmentVector, false, libMA::SuffixArrayInterface, libMA::NucSeq>
{
    const ma_params xP;
    const engine::BatcherOptions xBatcherOptions;
    std::mutex xBatcherMutex;
    std::shared_ptr<DeviceIndex> pBatcherIndex;
    std::shared_ptr<engine::DeviceBatcher> pBatcher;

    static engine::BatcherOptions withQueues( engine::BatcherOptions xOpt )
    {
        xOpt.bSocQueues = xOpt.bStages; // the reference's SoCPriorityQueue has to be filled eagerly (its pop() is not virtual)
        return xOpt;
    }
    std::shared_ptr<engine::DeviceBatcher> batcherFor( const std::shared_ptr<DeviceIndex>& pDev )
    {
        std::lock_guard<std::mutex> xGuard( xBatcherMutex );
        if( pBatcher == nullptr || pBatcherIndex != pDev )
        {
            pBatcher = std::make_shared<engine::DeviceBatcher>( pDev->p, xP, xBatcherOptions );
            pBatcherIndex = pDev;
        }
        return pBatcher;
    }

  public:
    BinarySeeding( const ::ParameterSetManager& rParameters )
        : xP( paramsOf( rParameters ) ), xBatcherOptions( withQueues( options( ).xBatcher ) )
    {}

    // binarySeeding.cpp:86-178.  Called concurrently and lock-free by all graph threads (export.cpp:84-126): the reads
    // that arrive together go through the device as ONE batch and through ALL stages at once; the returned container
    // carries the ticket the modules downstream pick their slices with.
    virtual std::shared_ptr<libMA::SegmentVector> execute( std::shared_ptr<libMA::SuffixArrayInterface> pFM_index,
                                                           std::shared_ptr<libMA::NucSeq> pQuerySeq ) override
    {
        auto pRet = std::make_shared<TicketedSegments>( );
        if( pQuerySeq == nullptr )
            return pRet;
        // a read that came through PrefetchReader has been aligned already; otherwise it joins the funnel and this thread waits
        if( const TicketedQuery* pAhead = dynamic_cast<const TicketedQuery*>( pQuerySeq.get( ) ) )
            pRet->xTicket = pAhead->xTicket;
        else
        {
            const auto pDev = deviceIndexOf( pFM_index.get( ) );
            pRet->xTicket = batcherFor( pDev )->align( engine::ReadRef( pQuerySeq->pGetSequenceRef( ), pQuerySeq->length( ) ) );
        }
        const engine::BatchResult& R = *pRet->xTicket.pResult;
        if( R.bStages )
        {
            const uint64_t uiFrom = R.vSegOff[ pRet->xTicket.uiRead ], uiTo = R.vSegOff[ pRet->xTicket.uiRead + 1 ];
            pRet->reserve( uiTo - uiFrom );
            for( uint64_t i = uiFrom; i < uiTo; i++ )
                pRet->emplace_back( (libMA::nucSeqIndex)R.vSegs[ i ].q_start, (libMA::nucSeqIndex)R.vSegs[ i ].q_size,
                                    libMA::SAInterval( R.vSegs[ i ].sa_start, R.vSegs[ i ].sa_start_rc, R.vSegs[ i ].sa_size ) );
        }
        return pRet;
    }

    // device batches run so far and the reads they carried (diagnostics)
    std::pair<uint64_t, uint64_t> batchStatistics( )
    {
        std::lock_guard<std::mutex> xGuard( xBatcherMutex );
        uint64_t uiBatches = 0, uiReads = 0;
        if( pBatcher != nullptr )
            pBatcher->stats( uiBatches, uiReads );
        return std::make_pair( uiBatches, uiReads );
    }
};

// ---- PrefetchReader: the funnel turned round ---------------------------------------------------------------------------
// A volatile source with the signature of the reader it wraps (fileReader.h:475: FileReader : Module<NucSeq, true, FileStream>;
// any Module<NucSeq, true, TP_ARGS...>).  It reads AHEAD: a device batch worth of reads is pulled from the wrapped reader
// (one caller at a time), goes through ALL stages on the GPU, and every execute( ) hands the calling graph thread one read of
// a FINISHED batch, as a TicketedQuery.  The five modules above find the ticket and only pick their slices: no graph thread
// waits for the GPU per read, so the graph of export.cpp:99-126 -- unchanged but for this one node --
//     auto pFileReader = std::make_shared<ma_amd::PrefetchReader<FileStream>>( rParameters, std::make_shared<FileReader>( rParameters ), pFMIndex );
// keeps the GPU busy with a few dozen threads (the per-read funnel behind BinarySeeding needs a thousand).
template <typename... TP_ARGS> class PrefetchReader : public libMS::Module<libMA::NucSeq, true, TP_ARGS...>
{
    typedef libMS::Module<libMA::NucSeq, true, TP_ARGS...> TP_SOURCE;
    std::shared_ptr<TP_SOURCE> pSource;
    std::shared_ptr<DeviceIndex> pDev; // keeps the device copy alive
    engine::PrefetchQueue<std::shared_ptr<libMA::NucSeq>> xQueue;

    static engine::PrefetchOptions withQueues( engine::PrefetchOptions xOpt )
    {
        xOpt.bSocQueues = xOpt.bStages; // as for the funnel: the reference's SoCPriorityQueue is filled eagerly
        return xOpt;
    }

  public:
    PrefetchReader( const ::ParameterSetManager& rParameters, std::shared_ptr<TP_SOURCE> pSource, std::shared_ptr<libMA::FMIndex> pFMIndex )
        : pSource( pSource ), pDev( deviceIndexOf( pFMIndex.get( ) ) ), xQueue( pDev->all( ), paramsOf( rParameters ), withQueues( options( ).xPrefetch ) )
    {}
    // nullptr = the wrapped reader is exhausted and every read it gave has been handed out (module.h:688-695)
    virtual std::shared_ptr<libMA::NucSeq> execute( std::shared_ptr<TP_ARGS>... pArgs ) override
    {
        std::shared_ptr<libMA::NucSeq> pQuery;
        engine::Ticket xTicket;
        if( !xQueue.next(
                pQuery, xTicket, [ & ]( ) { return pSource->execute( pArgs... ); },
                []( const std::shared_ptr<libMA::NucSeq>& pQ ) { return engine::ReadRef( pQ->pGetSequenceRef( ), pQ->length( ) ); } ) )
            return nullptr;
        auto pOut = std::make_shared<TicketedQuery>( *pQuery );
        pOut->xTicket = xTicket;
        return pOut;
    }
    std::pair<uint64_t, uint64_t> batchStatistics( )
    {
        uint64_t uiBatches = 0, uiReads = 0;
        double fRun = 0, fPull = 0;
        xQueue.stats( uiBatches, uiReads, fRun, fPull );
        return std::make_pair( uiBatches, uiReads );
    }
};

// ---- StripOfConsideration (ExtractSeeds + StripOfConsiderationSeeds) -----------------------------------------------
class StripOfConsideration
    : public libMS::Module<libMA::SoCPriorityQueue, false, libMA::SegmentVector, libMA::NucSeq, libMA::Pack, libMA::FMIndex>
{
    const ma_params xP;

  public:
    StripOfConsideration( const ::ParameterSetManager& rParameters ) : xP( paramsOf( rParameters ) )
    {}

    // stripOfConsideration.cpp:162-173
    virtual std::shared_ptr<libMA::SoCPriorityQueue> execute( std::shared_ptr<libMA::SegmentVector> pSegments,
                                                              std::shared_ptr<libMA::NucSeq> pQuery,
                                                              std::shared_ptr<libMA::Pack> pPack,
                                                              std::shared_ptr<libMA::FMIndex> pFM_index ) override
    {
        const auto pIn = std::dynamic_pointer_cast<TicketedSegments>( pSegments );
        if( pIn != nullptr && pIn->xTicket )
        {
            const engine::BatchResult& R = *pIn->xTicket.pResult;
            const size_t r = pIn->xTicket.uiRead;
            std::shared_ptr<TicketedSoCs> pRet;
            if( R.bStages )
                pRet = detail::makeQueue( R.vSocHeap.data( ) + R.vSocOff[ r ], R.vSocOff[ r + 1 ] - R.vSocOff[ r ],
                                          R.vSortedSeeds.data( ) + R.vSeedOff[ r ], R.vSeedOff[ r + 1 ] - R.vSeedOff[ r ], *pQuery );
            else
                pRet = detail::makeQueue( nullptr, 0, nullptr, 0, *pQuery ); // a shell that passes the ticket on
            pRet->xTicket = pIn->xTicket;
            return pRet;
        }
        // segments from elsewhere (the reference's BinarySeeding): upload them; extraction and the sweep run on the device
        const auto pDev = attachIndex( pPack, pFM_index );
        detail::SingleRead xBatch( pDev->p, xP, *pQuery );
        std::vector<ma_segment> vSegs;
        vSegs.reserve( pSegments->size( ) + 1 );
        for( const libMA::Segment& rSeg : *pSegments )
        {
            ma_segment r;
            r.q_start = (int64_t)rSeg.start( ), r.q_size = (int64_t)rSeg.size( );
            r.sa_start = rSeg.saInterval( ).start( ), r.sa_start_rc = rSeg.saInterval( ).startRevComp( ), r.sa_size = rSeg.saInterval( ).size( );
            vSegs.push_back( r );
        }
        const uint64_t aOff[ 2 ] = { 0, vSegs.size( ) };
        vSegs.push_back( ma_segment( ) );
        check( ma_batch_set_segments( xBatch.p, aOff, vSegs.data( ) ) );
        check( ma_extract_seeds_batch( xBatch.p ) );
        uint64_t nSeeds = 0, nStrips = 0;
        check( ma_batch_counts( xBatch.p, nullptr, &nSeeds, nullptr, nullptr, nullptr, nullptr, nullptr ) );
        check( ma_batch_get_soc_heap( xBatch.p, &nStrips, nullptr, nullptr, nullptr, nullptr ) );
        std::vector<ma_soc> vHeap( nStrips + 1 );
        std::vector<ma_seed> vSorted( nSeeds + 1 );
        uint64_t aSocOff[ 2 ], aSeedOff[ 2 ];
        check( ma_batch_get_soc_heap( xBatch.p, &nStrips, aSocOff, vHeap.data( ), aSeedOff, vSorted.data( ) ) );
        return detail::makeQueue( vHeap.data( ), nStrips, vSorted.data( ), nSeeds, *pQuery );
    }
};

// ---- Harmonization ---------------------------------------------------------------------------------------------------
class Harmonization : public libMS::Module<SeedSets, false, libMA::SoCPriorityQueue, libMA::NucSeq, libMA::FMIndex>
{
    const ma_params xP;

  public:
    Harmonization( const ::ParameterSetManager& rParameters ) : xP( paramsOf( rParameters ) )
    {}

    // harmonization.cpp:374-555
    virtual std::shared_ptr<SeedSets> execute( std::shared_ptr<libMA::SoCPriorityQueue> pSoCIn, std::shared_ptr<libMA::NucSeq> pQuery,
                                               std::shared_ptr<libMA::FMIndex> pFM_index ) override
    {
        auto pRet = std::make_shared<TicketedSeedSets>( );
        const auto pIn = std::dynamic_pointer_cast<TicketedSoCs>( pSoCIn );
        if( pIn != nullptr && pIn->xTicket )
        {
            pRet->xTicket = pIn->xTicket;
            const engine::BatchResult& R = *pRet->xTicket.pResult;
            if( R.bStages )
                detail::appendSeedSets( R.vHseedOff, R.vHsetSoc, R.vHseeds, R.vHsetOff[ pRet->xTicket.uiRead ],
                                        R.vHsetOff[ pRet->xTicket.uiRead + 1 ], *pQuery, *pRet );
            return pRet;
        }
        // a queue from elsewhere (the reference's StripOfConsideration): its state -- seeds as rectangularSoC left them and
        // the heap array -- is uploaded and the device pops and harmonizes from exactly there
        if( pSoCIn->pSeeds == nullptr )
        {
            if( pSoCIn->empty( ) )
                return pRet;
            throw std::runtime_error( "ma_amd::Harmonization: the SoC queue carries no seeds" );
        }
        const auto pDev = deviceIndexOf( pFM_index.get( ) );
        detail::SingleRead xBatch( pDev->p, xP, *pQuery );
        std::vector<ma_seed> vSeeds;
        vSeeds.reserve( pSoCIn->pSeeds->size( ) + 1 );
        for( const libMA::Seed& rSeed : *pSoCIn->pSeeds )
            vSeeds.push_back( detail::fromSeed( rSeed ) );
        std::vector<ma_soc> vHeap;
        vHeap.reserve( pSoCIn->vMaxima.size( ) + 1 );
        for( const auto& rStrip : pSoCIn->vMaxima )
        {
            ma_soc r;
            r.acc_len = std::get<0>( rStrip ).uiAccumulativeLength;
            r.ambiguity = std::get<0>( rStrip ).uiSeedAmbiguity;
            r.n_seeds = std::get<0>( rStrip ).uiSeedAmount;
            r.begin = (uint32_t)( std::get<1>( rStrip ) - pSoCIn->pSeeds->begin( ) );
            r.end = (uint32_t)( std::get<2>( rStrip ) - pSoCIn->pSeeds->begin( ) );
            vHeap.push_back( r );
        }
        const uint64_t aSeedOff[ 2 ] = { 0, vSeeds.size( ) }, aSocOff[ 2 ] = { 0, vHeap.size( ) };
        vSeeds.push_back( ma_seed( ) );
        vHeap.push_back( ma_soc( ) );
        check( ma_batch_set_soc_heap( xBatch.p, aSocOff, vHeap.data( ), aSeedOff, vSeeds.data( ) ) );
        check( ma_chain_batch( xBatch.p ) );
        uint64_t nSets = 0, nSeeds = 0;
        check( ma_batch_counts( xBatch.p, nullptr, nullptr, &nSets, &nSeeds, nullptr, nullptr, nullptr ) );
        std::vector<uint64_t> vSetOff( 2 ), vSeedOff( nSets + 1 );
        std::vector<uint32_t> vSoc( nSets + 1 );
        std::vector<ma_seed> vOut( nSeeds + 1 );
        check( ma_batch_get_hsets( xBatch.p, vSetOff.data( ), vSeedOff.data( ), vSoc.data( ), vOut.data( ) ) );
        detail::appendSeedSets( vSeedOff, vSoc, vOut, 0, nSets, *pQuery, *pRet );
        return pRet;
    }
};

// ---- NeedlemanWunsch -------------------------------------------------------------------------------------------------
class NeedlemanWunsch : public libMS::Module<Alignments, false, SeedSets, libMA::NucSeq, libMA::Pack>
{
    const ma_params xP;

  public:
    NeedlemanWunsch( const ::ParameterSetManager& rParameters ) : xP( paramsOf( rParameters ) )
    {}

    // needlemanWunsch.h:111-134
    virtual std::shared_ptr<Alignments> execute( std::shared_ptr<SeedSets> pSeedSets, std::shared_ptr<libMA::NucSeq> pQuery,
                                                 std::shared_ptr<libMA::Pack> pPack ) override
    {
        auto pRet = std::make_shared<TicketedAlignments>( );
        const auto pIn = std::dynamic_pointer_cast<TicketedSeedSets>( pSeedSets );
        if( pIn != nullptr && pIn->xTicket )
        {
            pRet->xTicket = pIn->xTicket;
            const engine::BatchResult& R = *pRet->xTicket.pResult;
            if( R.bStages )
                detail::appendAlignments( R.vAlns.data( ), R.vAlnOps.data( ), R.vAlnOff[ pRet->xTicket.uiRead ], R.vAlnOff[ pRet->xTicket.uiRead + 1 ],
                                          false, *pQuery, *pRet );
            return pRet;
        }
        // seed sets from elsewhere (the reference's Harmonization): upload, run the DP stage (the MappingQuality selection
        // of the same run is kept for a MappingQuality module that may follow)
        const auto pDev = deviceIndexOf( pPack.get( ) );
        detail::SingleRead xBatch( pDev->p, xP, *pQuery );
        std::vector<uint64_t> vSetOff{ 0, pSeedSets->size( ) }, vSeedOff{ 0 };
        std::vector<uint32_t> vSoc;
        std::vector<ma_seed> vSeeds;
        for( const auto& pSet : *pSeedSets )
        {
            vSoc.push_back( pSet->xStats.index_of_strip );
            for( const libMA::Seed& rSeed : *pSet )
                vSeeds.push_back( detail::fromSeed( rSeed ) );
            vSeedOff.push_back( vSeeds.size( ) );
        }
        vSoc.push_back( 0 );
        vSeeds.push_back( ma_seed( ) );
        check( ma_batch_set_hsets( xBatch.p, vSetOff.data( ), vSeedOff.data( ), vSoc.data( ), vSeeds.data( ) ) );
        check( ma_dp_batch( xBatch.p ) );
        detail::downloadAlignments( xBatch.p, false, *pQuery, *pRet );
        pRet->pWithQuality = std::make_shared<Alignments>( );
        detail::downloadAlignments( xBatch.p, true, *pQuery, *pRet->pWithQuality );
        return pRet;
    }
};

// ---- MappingQuality --------------------------------------------------------------------------------------------------
class MappingQuality : public libMS::Module<Alignments, false, libMA::NucSeq, Alignments>
{
    const ma_params xP;

  public:
    MappingQuality( const ::ParameterSetManager& rParameters ) : xP( paramsOf( rParameters ) )
    {}

    // mappingQuality.cpp:11-131
    virtual std::shared_ptr<Alignments> execute( std::shared_ptr<libMA::NucSeq> pQuery, std::shared_ptr<Alignments> pAlignments ) override
    {
        auto pRet = std::make_shared<TicketedAlignments>( );
        const auto pIn = std::dynamic_pointer_cast<TicketedAlignments>( pAlignments );
        if( pIn != nullptr && pIn->xTicket )
        {
            pRet->xTicket = pIn->xTicket;
            const engine::BatchResult& R = *pRet->xTicket.pResult;
            detail::appendAlignments( R.vMq.data( ), R.vMqOps.data( ), R.vMqOff[ pRet->xTicket.uiRead ], R.vMqOff[ pRet->xTicket.uiRead + 1 ], true, *pQuery,
                                      *pRet );
            return pRet;
        }
        if( pIn != nullptr && pIn->pWithQuality != nullptr )
        {
            for( const auto& pA : *pIn->pWithQuality )
                pRet->push_back( pA );
            return pRet;
        }
        // alignments from elsewhere (the reference's NeedlemanWunsch): upload them in the order it left them, the
        // MappingQuality kernel alone runs on the device (any device: it needs no index)
        if( pAlignments->empty( ) )
            return pRet;
        std::vector<ma_alignment> vAlns;
        std::vector<uint64_t> vOps;
        for( const auto& pA : *pAlignments )
        {
            ma_alignment r;
            r.begin_ref = (int64_t)pA->uiBeginOnRef, r.end_ref = (int64_t)pA->uiEndOnRef;
            r.begin_q = (int64_t)pA->uiBeginOnQuery, r.end_q = (int64_t)pA->uiEndOnQuery;
            r.score = pA->iScore;
            r.soc_index = pA->xStats.index_of_strip;
            r.n_ops = (uint32_t)pA->data.size( );
            r.ops_off = vOps.size( ) / 2;
            r.secondary = pA->bSecondary ? 1 : 0, r.supplementary = pA->bSupplementary ? 1 : 0;
            r.mapq = 0;
            for( const auto& rOp : pA->data )
            {
                vOps.push_back( (uint64_t)rOp.first );
                vOps.push_back( (uint64_t)rOp.second );
            }
            vAlns.push_back( r );
        }
        const auto pDev = anyDeviceIndex( );
        detail::SingleRead xBatch( pDev->p, xP, *pQuery );
        const uint64_t aOff[ 2 ] = { 0, vAlns.size( ) };
        vOps.push_back( 0 );
        check( ma_batch_set_alignments( xBatch.p, aOff, vAlns.data( ), vOps.data( ) ) );
        // the reference's MappingQuality re-flags and re-scores the very Alignment objects it was handed and returns the
        // kept ones; mirror that: the device returns flags, qualities and the selection as indices into the input order
        uint64_t nAln = 0, nOps = 0;
        check( ma_batch_counts( xBatch.p, nullptr, nullptr, nullptr, nullptr, &nAln, &nOps, nullptr ) );
        std::vector<uint64_t> vOff( 2 ), vOutOps( 2 * nOps + 2 );
        std::vector<ma_alignment> vOut( nAln + 1 );
        check( ma_batch_get_mapq_alignments( xBatch.p, vOff.data( ), vOut.data( ), vOutOps.data( ) ) );
        std::vector<bool> vTaken( vAlns.size( ), false );
        for( uint64_t i = 0; i < vOff[ 1 ]; i++ )
        {
            const ma_alignment& rO = vOut[ i ];
            size_t uiMatch = vAlns.size( );
            for( size_t k = 0; k < vAlns.size( ) && uiMatch == vAlns.size( ); k++ )
            {
                const ma_alignment& rI = vAlns[ k ];
                if( vTaken[ k ] || rI.begin_ref != rO.begin_ref || rI.end_ref != rO.end_ref || rI.begin_q != rO.begin_q ||
                    rI.end_q != rO.end_q || rI.score != rO.score || rI.soc_index != rO.soc_index || rI.n_ops != rO.n_ops )
                    continue;
                bool bSame = true;
                for( uint32_t j = 0; j < 2 * rI.n_ops && bSame; j++ )
                    bSame = vOps[ 2 * rI.ops_off + j ] == vOutOps[ 2 * rO.ops_off + j ];
                if( bSame )
                    uiMatch = k;
            }
            if( uiMatch == vAlns.size( ) )
                throw std::runtime_error( "ma_amd::MappingQuality: device result does not match an input alignment" );
            vTaken[ uiMatch ] = true;
            const auto& pA = ( *pAlignments )[ uiMatch ];
            pA->bSecondary = rO.secondary != 0;
            pA->bSupplementary = rO.supplementary != 0;
            pA->fMappingQuality = rO.mapq;
            pRet->push_back( pA );
        }
        return pRet;
    }

  private:
    static std::shared_ptr<DeviceIndex> anyDeviceIndex( )
    {
        auto& rReg = detail::registry( );
        std::lock_guard<std::mutex> xGuard( rReg.xMutex );
        for( auto& rKV : rReg.xByObject )
            if( !rKV.second.pOwner.expired( ) )
                return rKV.second.pDev;
        throw std::runtime_error( "ma_amd::MappingQuality: no index attached (ma_amd::attachIndex)" );
    }
};
// ---- BufferedFileWriter: the reference's FileWriter, its lock taken once per 64 KB instead of once per read ----------------
// FileWriter::execute formats a read's SAM records outside its lock and appends them under it (fileWriter.cpp:141-145), and
// its FileOutStream flushes on every append.  With the GPU modules in front of it that lock is what bounds the graph
// (profiles/r04_binding_graph_rate.txt: 0.8 M reads/s): a graph thread that is descheduled while it holds the lock stalls all
// the others, and every read pays a write( ).  This subclass keeps the reference's formatter -- every calling thread owns a
// PRIVATE libMA::FileWriter on an in-memory stream, so the bytes of a read's records are the reference's by construction --
// and forwards a thread's text to the shared stream under the shared lock when it has uiBufferBytes of it; flush( ) (or the
// destructor) writes what is left.  The records of one read stay together, the order between threads is as arbitrary as in
// the reference.  Same template signature and constructors as FileWriter: `std::make_shared<ma_amd::BufferedFileWriter>( rParameters,
// sFileName, pPack )` replaces `std::make_shared<FileWriter>( ... )` at export.cpp:109.  rParameters must outlive the writer.
class BufferedFileWriter : public libMA::FileWriter
{
    struct Collect : public libMA::OutStream
    {
        std::string sText;
        Collect& operator<<( std::string s ) override
        {
            sText += s;
            return *this;
        }
    };
    struct PerThread
    {
        std::shared_ptr<Collect> pText;
        std::unique_ptr<libMA::FileWriter> pFormatter;
    };
    const ::ParameterSetManager& rParameters;
    std::shared_ptr<libMA::Pack> pPackOfHeader;
    std::mutex xThreadsMutex;
    std::vector<std::unique_ptr<PerThread>> vThreads; // one per thread that ever called execute( ) on this writer
    const uint64_t uiId = nextId( );
    static uint64_t nextId( )
    {
        static std::atomic<uint64_t> uiNext{ 1 };
        return uiNext++;
    }
    PerThread& mine( )
    {
        static thread_local std::vector<std::pair<uint64_t, PerThread*>> vMine; // by writer id (ids are never reused)
        for( auto& rEntry : vMine )
            if( rEntry.first == uiId )
                return *rEntry.second;
        std::unique_ptr<PerThread> pNew( new PerThread( ) );
        pNew->pText = std::make_shared<Collect>( );
        pNew->pFormatter.reset( new libMA::FileWriter( rParameters, std::static_pointer_cast<libMA::OutStream>( pNew->pText ), pPackOfHeader ) );
        pNew->pText->sText.clear( ); // the private writer's header: the shared stream has the one the base class wrote
        std::lock_guard<std::mutex> xGuard( xThreadsMutex );
        vThreads.push_back( std::move( pNew ) );
        vMine.emplace_back( uiId, vThreads.back( ).get( ) );
        return *vThreads.back( );
    }
    void forward( std::string& rText )
    {
        if( rText.empty( ) )
            return;
        {
            std::lock_guard<std::mutex> xGuard( *pLock );
            *pOut << rText;
        }
        rText.clear( );
    }

  public:
    size_t uiBufferBytes = 1u << 16;

    BufferedFileWriter( const ::ParameterSetManager& rParameters, std::string sFileName, std::shared_ptr<libMA::Pack> pPackContainer )
        : libMA::FileWriter( rParameters, sFileName, pPackContainer ), rParameters( rParameters ), pPackOfHeader( pPackContainer )
    {}
    BufferedFileWriter( const ::ParameterSetManager& rParameters, std::shared_ptr<libMA::OutStream> pOut,
                        std::shared_ptr<libMA::Pack> pPackContainer )
        : libMA::FileWriter( rParameters, pOut, pPackContainer ), rParameters( rParameters ), pPackOfHeader( pPackContainer )
    {}
    ~BufferedFileWriter( )
    {
        flush( );
    }
    virtual std::shared_ptr<libMS::Container> execute( std::shared_ptr<libMA::NucSeq> pQuery, std::shared_ptr<Alignments> pAlignments,
                                                       std::shared_ptr<libMA::Pack> pPack ) override
    {
        PerThread& rMine = mine( );
        auto pRet = rMine.pFormatter->execute( pQuery, pAlignments, pPack ); // the reference's formatting, into this thread's text
        if( rMine.pText->sText.size( ) >= uiBufferBytes )
            forward( rMine.pText->sText );
        return pRet;
    }
    // writes what the threads' buffers still hold; call it (from one thread) when the graph threads are done and before the
    // stream is read or closed
    void flush( )
    {
        std::lock_guard<std::mutex> xGuard( xThreadsMutex );
        for( auto& pThread : vThreads )
            forward( pThread->pText->sText );
    }
};
} // namespace ma_amd
