// ma_batch_nodes.h -- the THROUGHPUT form of the path as graph nodes (adaptor (i)): modules whose containers are
// whole batches, so that reader -> aligner -> writer runs under promiseMe / BasePledge::simultaneousGet like the per-read
// chain of libMA::setUpCompGraph (export.cpp:84-126) does, with a handful of graph threads instead of thousands:
//
//   BatchFileReader : Module<ReadVector, true,  FileStream>                 up to uiBatchReads reads per call, nullptr = EoF
//   BatchAlign      : Module<AlignedBatch, false, FMIndex, ReadVector>      one device batch through ALL stages; the result
//                                                                           stays FLAT (one header + one ops array)
//   BatchFileWriter : Module<Container, false, ReadVector, AlignedBatch, Pack>   SAM text of the batch formatted from the
//                                                                           flat view into byte arenas, ONE write per batch
//
// ReadVector = ContainerVector<shared_ptr<NucSeq>> -- the reference's own way of passing many containers at once
// (needlemanWunsch.h:51-52 takes ContainerVector<shared_ptr<Seeds>>; BinarySeeding::seed takes a vector of queries,
// binarySeeding.h:575-584).  Every graph thread that calls BatchAlign::execute gets its own engine (stream + device batch),
// so the number of graph threads IS the number of device batches in flight.
#pragma once
#include "ma_flat_sam.h"
#include "ma_sam.h"
#include <atomic>
#include <thread>

namespace libMA
{
class BatchAlign : public libMS::Module<AlignedBatch, false, FMIndex, ReadVector>
{
    const ma_params xP;
    std::mutex xMutex;
    // engines whose graph thread is between two batches, with the index they are bound to (an engine's ma_batch belongs to ONE
    // ma_index: reusing it for another genome would align against the first one)
    std::vector<std::pair<const ma_index*, std::unique_ptr<detail::Engine>>> vIdle;
    std::mutex xPrimeMutex; // first batches of new engines run one at a time (they allocate GBs and page-lock memory)
    std::atomic<size_t> uiTurn{ 0 }; // which replica of the index takes the next batch

  public:
    // seconds summed over all batches (phases of different batches overlap when several graph threads are at work)
    std::atomic<uint64_t> uiBatches{ 0 }, uiReads{ 0 }, uiAligned{ 0 };
    double fPack = 0, fH2D = 0, fKernels = 0, fD2H = 0;

    BatchAlign( const ParameterSetManager& rParameters ) : xP( *rParameters.getSelected( ) )
    {}

    virtual std::shared_ptr<AlignedBatch> execute( std::shared_ptr<FMIndex> pFM_index, std::shared_ptr<ReadVector> pReads ) override
    {
        auto pRet = std::make_shared<AlignedBatch>( );
        pRet->pReads = pReads;
        pRet->uiFirst = 0;
        // device batches rotate over the replicas of the index (DeviceIndex::vReplicas: one per GPU of the node):
        // the graph of reader -> BatchAlign -> writer stays as it is and feeds all of them
        const ma_index* pIndex = pFM_index->pDev->p;
        if( !pFM_index->pDev->vReplicas.empty( ) )
        {
            const size_t k = uiTurn.fetch_add( 1 ) % ( pFM_index->pDev->vReplicas.size( ) + 1 );
            if( k > 0 )
                pIndex = pFM_index->pDev->vReplicas[ k - 1 ]->p;
        }
        std::unique_ptr<detail::Engine> pEngine;
        {
            std::lock_guard<std::mutex> xGuard( xMutex );
            for( size_t k = 0; k < vIdle.size( ); k++ )
                if( vIdle[ k ].first == pIndex )
                {
                    pEngine = std::move( vIdle[ k ].second );
                    vIdle.erase( vIdle.begin( ) + k );
                    break;
                }
        }
        std::vector<detail::ReadRef> vRefs;
        vRefs.reserve( pReads->size( ) );
        for( const auto& pQ : *pReads )
            vRefs.emplace_back( pQ->xCodes );
        try
        {
            if( pEngine == nullptr || !pEngine->primed( vRefs.size( ) ) )
            {
                // engines before admission: the graph threads all arrive at once at the start of a run; their engines are
                // created and run their first batch one after the other instead of racing for the allocator and the page-locker
                std::lock_guard<std::mutex> xPrime( xPrimeMutex );
                if( pEngine == nullptr )
                    pEngine.reset( new detail::Engine( pIndex, xP ) );
                pRet->pResult = pEngine->run( vRefs, false );
            }
            else
                pRet->pResult = pEngine->run( vRefs, false );
        }
        catch( ... )
        {
            if( pEngine != nullptr )
            {
                std::lock_guard<std::mutex> xGuard( xMutex );
                vIdle.emplace_back( pIndex, std::move( pEngine ) );
            }
            throw;
        }
        std::lock_guard<std::mutex> xGuard( xMutex );
        vIdle.emplace_back( pIndex, std::move( pEngine ) );
        uiBatches++, uiReads += pReads->size( ), uiAligned += pRet->pResult->uiAlignedReads;
        fPack += pRet->pResult->fPack, fH2D += pRet->pResult->fH2D, fKernels += pRet->pResult->fKernels, fD2H += pRet->pResult->fD2H;
        return pRet;
    }
};

// Volatile source of batches out of a FASTA / FASTQ stream: the records are parsed exactly like FileReader does it
// (fileReader.cpp:37-203), up to uiBatchReads of them under ONE acquisition of the stream's lock.
class BatchFileReader : public libMS::Module<ReadVector, true, FileStream>
{
    std::atomic<size_t> uiLastBatchBytes{ 0 }; // text of the previous batch: the next one reserves that much up front

  public:
    size_t uiBatchReads = 1u << 18;
    BatchFileReader( const ParameterSetManager& )
    {}
    virtual std::shared_ptr<ReadVector> execute( std::shared_ptr<FileStream> pStream ) override
    {
        auto pRet = std::make_shared<ReadVector>( );
        std::string sRecords; // the text of the batch, cut out of the stream under its lock ...
        sRecords.reserve( uiLastBatchBytes + uiLastBatchBytes / 8 );
        size_t uiRecords = 0;
        {
            std::lock_guard<std::mutex> xLock( pStream->xMutex );
            const bool bCut = pStream->capture( &sRecords );
            try
            {
                while( pRet->size( ) + uiRecords < uiBatchReads )
                {
                    if( bCut )
                    {
                        if( !FileReader::skipRecord( *pStream ) )
                            break;
                        uiRecords++;
                    }
                    else if( auto pQ = FileReader::parseRecord( *pStream ) ) // a stream that cannot be cut: parse in place
                        pRet->push_back( pQ );
                    else
                        break;
                }
            }
            catch( ... )
            {
                pStream->capture( nullptr );
                throw;
            }
            pStream->capture( nullptr );
        }
        // ... and turned into reads outside of it, while the next graph thread cuts its batch
        if( uiRecords > 0 )
        {
            uiLastBatchBytes = sRecords.size( );
            StringStream xText( std::move( sRecords ) );
            pRet->reserve( uiRecords );
            while( auto pQ = FileReader::parseRecord( xText ) )
                pRet->push_back( pQ );
        }
        if( pRet->empty( ) )
            return nullptr; // end of file (module.h:688-695)
        return pRet;
    }
};

// Volatile source over reads that already are in memory (benchmarks, callers that hold their reads): consecutive slices.
class BatchSource : public libMS::Module<ReadVector, true>
{
    const std::shared_ptr<ReadVector> pAll;
    std::atomic<size_t> uiNext{ 0 };

  public:
    size_t uiBatchReads = 1u << 18;
    BatchSource( std::shared_ptr<ReadVector> pAll ) : pAll( pAll )
    {}
    virtual std::shared_ptr<ReadVector> execute( ) override
    {
        const size_t lo = uiNext.fetch_add( uiBatchReads );
        if( lo >= pAll->size( ) )
            return nullptr;
        return std::make_shared<ReadVector>( pAll->begin( ) + lo, pAll->begin( ) + std::min( pAll->size( ), lo + uiBatchReads ) );
    }
};

// SAM records of a whole batch (fileWriter.cpp:11-158 per read), formatted from the flat view by uiFormatThreads threads
// into one arena each and written arena by arena -- in input order -- under one acquisition of the writer's lock.
class BatchFileWriter : public libMS::Module<libMS::Container, false, ReadVector, AlignedBatch, Pack>
{
    std::shared_ptr<FileWriter> pPerRead; // owns stream + lock (and serves the options the flat formatter does not)
    ma_amd::flat::SamFormat xFormat;
    ma_amd::flat::Contigs xContigs;

  public:
    size_t uiFormatThreads = 8;
    std::atomic<uint64_t> uiBytes{ 0 }, uiReads{ 0 };

    BatchFileWriter( const ParameterSetManager& rParameters, std::shared_ptr<OutStream> pOut, std::shared_ptr<Pack> pPack )
        : pPerRead( std::make_shared<FileWriter>( rParameters, pOut, pPack ) )
    {
        const SamOptions& rO = rParameters.xSam;
        xFormat.bNoSecondary = rO.bNoSecondary, xFormat.bNoSupplementary = rO.bNoSupplementary;
        xFormat.bOutputMCigar = rO.bOutputMCigar, xFormat.bCGTag = rO.bCGTag, xFormat.bSoftClip = rO.bSoftClip;
        xContigs.vNames = pPack->vNames, xContigs.vStarts = pPack->vStarts, xContigs.vLengths = pPack->vLengths;
    }

    virtual std::shared_ptr<libMS::Container> execute( std::shared_ptr<ReadVector> pReads, std::shared_ptr<AlignedBatch> pAligned,
                                                       std::shared_ptr<Pack> pPack ) override
    {
        const size_t n = pReads->size( );
        if( pAligned->size( ) != n )
            throw std::runtime_error( "BatchFileWriter: the batch of alignments does not belong to these reads" );
        if( pPerRead->xOptions.bEmulateNgmlrTags ) // needs Alignment objects and reference bases: the per-read writer
        {
            for( size_t i = 0; i < n; i++ )
                pPerRead->execute( ( *pReads )[ i ], pAligned->alignmentsOf( i ), pPack );
            uiReads += n;
            return std::make_shared<libMS::Container>( );
        }
        const size_t uiThreads = std::max<size_t>( 1, std::min<size_t>( uiFormatThreads, n / 2048 + 1 ) );
        std::vector<ma_amd::flat::Arena> vArenas( uiThreads );
        std::string sFailure;
        std::mutex xFailure;
        auto format = [ & ]( size_t t ) {
            try
            {
                ma_amd::flat::Arena& rOut = vArenas[ t ];
                const uint64_t* pOff = pAligned->offsets( );
                for( size_t i = n * t / uiThreads; i < n * ( t + 1 ) / uiThreads; i++ )
                {
                    const NucSeq& rQ = *( *pReads )[ i ];
                    ma_amd::flat::ReadView xQ;
                    xQ.sName = rQ.sName.data( ), xQ.uiNameLen = rQ.sName.size( );
                    xQ.pCodes = rQ.xCodes.data( ), xQ.uiLength = rQ.xCodes.size( );
                    xQ.pQuality = rQ.xQuality.empty( ) ? nullptr : rQ.xQuality.data( );
                    ma_amd::flat::formatRead( rOut, xFormat, xContigs, xQ, pAligned->alignments( ) + pOff[ i ], pOff[ i + 1 ] - pOff[ i ],
                                              pAligned->ops( ) );
                }
            }
            catch( const std::exception& rE )
            {
                std::lock_guard<std::mutex> xGuard( xFailure );
                if( sFailure.empty( ) )
                    sFailure = rE.what( );
            }
        };
        std::vector<std::thread> vT;
        for( size_t t = 1; t < uiThreads; t++ )
            vT.emplace_back( format, t );
        format( 0 );
        for( auto& rT : vT )
            rT.join( );
        if( !sFailure.empty( ) )
            throw std::runtime_error( sFailure );
        uint64_t uiTotal = 0;
        {
            std::lock_guard<std::mutex> xGuard( *pPerRead->pLock );
            for( const auto& rArena : vArenas )
            {
                pPerRead->pOut->write( rArena.data( ), rArena.size( ) );
                uiTotal += rArena.size( );
            }
        }
        uiBytes += uiTotal, uiReads += n;
        return std::make_shared<libMS::Container>( );
    }
    virtual bool requiresLock( ) const
    {
        return false; // serialises its own output
    }
};
} // namespace libMA
