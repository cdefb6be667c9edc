// ms_graph.h -- minimal libMS-compatible computational-graph surface (Module / Pledge / Container /
// ContainerVector / promiseMe / makePledge / simultaneousGet) so that the MI355X modules in
// ma_modules.h can be wired exactly like libMA::setUpCompGraph wires the reference's
// (libs/ma/src/util/export.cpp:99-126).  Semantics follow libs/ms/inc/ms/module/module.h:
//   Module::execute / executeTup (87-112), requiresLock (114), Pledge::get (674-721: EOF = nullptr from
//   a volatile source, a non-volatile module must not return nullptr), promiseMe (735-741),
//   makePledge (755-760), BasePledge::simultaneousGet (268-378: one worker per graph sink, first
//   exception kept and rethrown after the join).
// This is our own code against the reference's interface; when integrating into MA itself the
// reference's module.h is used instead and only ma_modules.h is added.
#pragma once
#include <algorithm>
#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

namespace libMS
{
class Container // libs/ms/inc/ms/container/container.h:36-65
{
  public:
    virtual ~Container( )
    {}
    virtual std::string getTypeName( ) const
    {
        return "Container";
    }
};

template <typename T> class ContainerVector : public Container, public std::vector<T> // container.h:67-211
{
  public:
    using std::vector<T>::vector;
    std::string getTypeName( ) const override
    {
        return "ContainerVector";
    }
};

template <typename TP_RETURN_, bool IS_VOLATILE_, typename... TP_ARGUMENTS> class Module
{
  public:
    typedef TP_RETURN_ TP_RETURN;
    static constexpr bool IS_VOLATILE = IS_VOLATILE_;
    typedef std::tuple<std::shared_ptr<TP_ARGUMENTS>...> TP_TUPLE_ARGS;

    virtual std::shared_ptr<TP_RETURN> execute( std::shared_ptr<TP_ARGUMENTS>... )
    {
        throw std::runtime_error( "module did not implement execute" );
    }
    virtual std::shared_ptr<TP_RETURN> executeTup( const TP_TUPLE_ARGS& t )
    {
        return call( t, std::index_sequence_for<TP_ARGUMENTS...>( ) );
    }
    virtual bool requiresLock( ) const
    {
        return false;
    }
    virtual ~Module( )
    {}

  private:
    template <size_t... I> std::shared_ptr<TP_RETURN> call( const TP_TUPLE_ARGS& t, std::index_sequence<I...> )
    {
        return this->execute( std::get<I>( t )... );
    }
};

class BasePledge
{
  public:
    virtual ~BasePledge( )
    {}
    virtual std::shared_ptr<Container> getAsBaseType( ) = 0;
    virtual bool hasVolatile( ) const = 0;
    virtual void reset( ) = 0; // forget the cached content of this pledge and of everything downstream
    virtual void addSuccessor( BasePledge* ) = 0;

    // Drives all sinks of a graph to exhaustion (interface and semantics of BasePledge::simultaneousGet,
    // module.h:268-378): every sink is pulled until its volatile source runs dry, sink 0 reports progress through the
    // callback (false = stop everybody), and the first failure of any worker stops the others and reaches the caller as
    // std::runtime_error once all workers are back.  numThreads workers share the sinks (0 = one worker per sink).
    static inline void simultaneousGet(
        std::vector<std::shared_ptr<BasePledge>> vPledges, std::function<bool( )> callback = []( ) { return true; },
        unsigned int numThreads = 0 )
    {
        struct Run
        {
            std::atomic<size_t> uiNextSink{ 0 };
            std::atomic<bool> bGo{ true };
            std::once_flag xFirstFailure;
            std::string sFailure;
            bool bFailed = false;
        } xRun;
        auto drain = [ & ]( size_t uiSink ) {
            BasePledge& rSink = *vPledges[ uiSink ];
            const bool bRepeats = rSink.hasVolatile( );
            for( ;; )
            {
                const bool bMore = rSink.getAsBaseType( ) != nullptr && bRepeats;
                if( uiSink == 0 && !callback( ) )
                    xRun.bGo = false;
                if( !bMore || !xRun.bGo )
                    return;
            }
        };
        auto work = [ & ]( ) {
            for( size_t uiSink = xRun.uiNextSink++; uiSink < vPledges.size( ); uiSink = xRun.uiNextSink++ )
                try
                {
                    drain( uiSink );
                }
                catch( const std::exception& rFailure )
                {
                    std::call_once( xRun.xFirstFailure, [ & ]( ) {
                        xRun.sFailure = rFailure.what( );
                        xRun.bFailed = true;
                    } );
                    xRun.bGo = false;
                    return;
                }
        };
        const size_t uiWorkers = numThreads == 0 ? vPledges.size( ) : std::min<size_t>( numThreads, vPledges.size( ) );
        std::vector<std::thread> vWorkers;
        for( size_t k = 0; k < uiWorkers; k++ )
            vWorkers.emplace_back( work );
        for( auto& rWorker : vWorkers )
            rWorker.join( );
        if( xRun.bFailed )
            throw std::runtime_error( xRun.sFailure );
    }
};

template <class TP_TYPE, bool IS_VOLATILE = false, typename... TP_DEPENDENCIES> class Pledge : public BasePledge
{
  public:
    typedef TP_TYPE TP_CONTENT;
    typedef Module<TP_CONTENT, IS_VOLATILE, typename TP_DEPENDENCIES::TP_CONTENT...> TP_PLEDGER;
    typedef std::tuple<std::shared_ptr<TP_DEPENDENCIES>...> TP_PREDECESSORS;
    typedef std::tuple<std::shared_ptr<typename TP_DEPENDENCIES::TP_CONTENT>...> TP_INPUT;

    Pledge( )
    {}
    Pledge( std::shared_ptr<TP_PLEDGER> pPledger, std::shared_ptr<TP_DEPENDENCIES>... tPredecessors )
        : pPledger( pPledger ), tPredecessors( tPredecessors... )
    {
        (void)std::initializer_list<int>{ ( tPredecessors->addSuccessor( this ), 0 )... };
    }

    void addSuccessor( BasePledge* pX ) override
    {
        vSuccessors.push_back( pX );
    }

    // module.h:600-625: clears the content (unless this is a constant) and all successors
    void reset( ) override
    {
        if( pContent == nullptr || pPledger == nullptr )
            return; // already reset / a constant made with makePledge keeps its content
        pContent = nullptr;
        for( BasePledge* pS : vSuccessors )
            pS->reset( );
    }

    void set( std::shared_ptr<TP_CONTENT> pC )
    {
        pContent = pC;
    }

    bool hasVolatile( ) const override
    {
        if( IS_VOLATILE )
            return true;
        return anyVolatile( std::index_sequence_for<TP_DEPENDENCIES...>( ) );
    }

    virtual std::shared_ptr<TP_CONTENT> get( )
    {
        if( pPledger == nullptr && pContent == nullptr )
            throw std::runtime_error( "No pledger known for unfulfilled pledge" );
        if( pPledger == nullptr )
            return pContent;
        if( !IS_VOLATILE && pContent != nullptr )
            return pContent; // module.h:685-687: no need to execute again until someone reset()s us
        TP_INPUT tInput;
        if( !fill( tInput, std::index_sequence_for<TP_DEPENDENCIES...>( ) ) )
            return std::shared_ptr<TP_CONTENT>( nullptr ); // EOF of a volatile source upstream
        std::shared_ptr<TP_CONTENT> pRet;
        if( pPledger->requiresLock( ) )
        {
            std::lock_guard<std::mutex> xGuard( xMutex );
            pRet = pPledger->executeTup( tInput );
        }
        else
            pRet = pPledger->executeTup( tInput );
        pContent = pRet;
        if( pRet == nullptr && !IS_VOLATILE )
            throw std::runtime_error( "An non-volatile module is not allowed to return nullpointers in execute; "
                                      "throw an exception instead or return an empty container!" );
        return pRet;
    }

    std::shared_ptr<Container> getAsBaseType( ) override
    {
        return std::dynamic_pointer_cast<Container>( this->get( ) );
    }

  private:
    template <size_t... I> bool fill( TP_INPUT& tInput, std::index_sequence<I...> )
    {
        bool ok = true;
        // evaluated left to right like the reference's TemplateLoop; stops at the first EOF
        (void)std::initializer_list<int>{
            ( ok = ok && ( ( std::get<I>( tInput ) = std::get<I>( tPredecessors )->get( ) ) != nullptr ), 0 )... };
        return ok;
    }
    template <size_t... I> bool anyVolatile( std::index_sequence<I...> ) const
    {
        bool v = false;
        (void)std::initializer_list<int>{ ( v = v || std::get<I>( tPredecessors )->hasVolatile( ), 0 )... };
        return v;
    }
    std::shared_ptr<TP_PLEDGER> pPledger;
    std::shared_ptr<TP_CONTENT> pContent;
    TP_PREDECESSORS tPredecessors;
    std::vector<BasePledge*> vSuccessors;
    std::mutex xMutex;
};

// Lock / UnLock (libs/ms/inc/ms/module/splitter.h:29-120): Lock pins the element a volatile source
// delivered so that every consumer of this graph iteration sees the same one; UnLock, placed at the
// sink, resets the lock pledge (and with it every cached pledge downstream) for the next iteration.
template <typename TP_CONTAINER> class Lock : public Module<TP_CONTAINER, false, TP_CONTAINER>
{
  public:
    std::shared_ptr<TP_CONTAINER> execute( std::shared_ptr<TP_CONTAINER> pIn ) override
    {
        return pIn;
    }
};
template <typename TP_CONTAINER> class UnLock : public Module<TP_CONTAINER, true, TP_CONTAINER> // volatile (splitter.h:62)
{
  public:
    std::shared_ptr<BasePledge> pLockPledge;
    UnLock( std::shared_ptr<BasePledge> pLockPledge ) : pLockPledge( pLockPledge )
    {}
    std::shared_ptr<TP_CONTAINER> execute( std::shared_ptr<TP_CONTAINER> pIn ) override
    {
        pLockPledge->reset( );
        return pIn;
    }
};

// TupleGet (splitter.h:83-94): element IDX of a container of shared pointers (e.g. one mate of a PairedReadsContainer)
template <typename TP_TUPLE, size_t IDX>
class TupleGet : public Module<typename TP_TUPLE::value_type::element_type, false, TP_TUPLE>
{
  public:
    typename TP_TUPLE::value_type execute( std::shared_ptr<TP_TUPLE> pIn ) override
    {
        return ( *pIn )[ IDX ];
    }
};

template <class TP_MODULE, class... TP_PLEDGES>
std::shared_ptr<Pledge<typename TP_MODULE::TP_RETURN, TP_MODULE::IS_VOLATILE, TP_PLEDGES...>>
promiseMe( std::shared_ptr<TP_MODULE> pModule, std::shared_ptr<TP_PLEDGES>... pPledges )
{
    return std::make_shared<Pledge<typename TP_MODULE::TP_RETURN, TP_MODULE::IS_VOLATILE, TP_PLEDGES...>>( pModule,
                                                                                                            pPledges... );
}

template <class TP_CONTAINER, class... TP_ARGS> std::shared_ptr<Pledge<TP_CONTAINER>> makePledge( TP_ARGS&&... args )
{
    auto pRet = std::make_shared<Pledge<TP_CONTAINER>>( );
    pRet->set( std::make_shared<TP_CONTAINER>( args... ) );
    return pRet;
}
} // namespace libMS
