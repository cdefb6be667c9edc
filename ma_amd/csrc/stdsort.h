// stdsort.h -- device-callable restatement of libstdc++'s (GCC 11) std::sort / std::make_heap /
// std::pop_heap / std::lower_bound so that tie orders match the reference, whose results depend on
// the permutation produced by these unstable algorithms (stripOfConsideration.cpp:33, soc.h:213,280,
// harmonization.cpp:187,331, needlemanWunsch.h:131, mappingQuality.cpp:14,104).
// libstdc++ is not part of /root/reference; the algorithm restated here is bits/stl_algo.h
// (__introsort_loop, __unguarded_partition_pivot, __move_median_to_first, __final_insertion_sort,
// threshold 16) and bits/stl_heap.h (__adjust_heap, __push_heap, __make_heap, __pop_heap) of
// GCC 11.4; tests/test_stdsort.py pins it against the real std::sort on inputs full of ties.
#pragma once
#include "ma_common.h"

namespace ma
{
namespace ss
{
template <typename T, typename C> MA_HD void push_heap_( T* first, i64 hole, i64 top, T value, C comp )
{
    i64 parent = ( hole - 1 ) / 2;
    while( hole > top && comp( first[ parent ], value ) )
    {
        first[ hole ] = first[ parent ];
        hole = parent;
        parent = ( hole - 1 ) / 2;
    }
    first[ hole ] = value;
}

template <typename T, typename C> MA_HD void adjust_heap( T* first, i64 hole, i64 len, T value, C comp )
{
    const i64 top = hole;
    i64 second = hole;
    while( second < ( len - 1 ) / 2 )
    {
        second = 2 * ( second + 1 );
        if( comp( first[ second ], first[ second - 1 ] ) )
            second--;
        first[ hole ] = first[ second ];
        hole = second;
    }
    if( ( len & 1 ) == 0 && second == ( len - 2 ) / 2 )
    {
        second = 2 * ( second + 1 );
        first[ hole ] = first[ second - 1 ];
        hole = second - 1;
    }
    push_heap_( first, hole, top, value, comp );
}

template <typename T, typename C> MA_HD void make_heap( T* first, i64 len, C comp )
{
    if( len < 2 )
        return;
    i64 parent = ( len - 2 ) / 2;
    while( true )
    {
        T value = first[ parent ];
        adjust_heap( first, parent, len, value, comp );
        if( parent == 0 )
            return;
        parent--;
    }
}

// __pop_heap(first, last, result): heap is [first,last), *result receives the top
template <typename T, typename C> MA_HD void pop_heap_to( T* first, i64 last, i64 result, C comp )
{
    T value = first[ result ];
    first[ result ] = first[ 0 ];
    adjust_heap( first, (i64)0, last, value, comp );
}

// std::pop_heap(first, first+len): moves the top to first[len-1]
template <typename T, typename C> MA_HD void pop_heap( T* first, i64 len, C comp )
{
    if( len > 1 )
        pop_heap_to( first, len - 1, len - 1, comp );
}

template <typename T, typename C> MA_HD void unguarded_linear_insert( T* a, i64 last, C comp )
{
    T val = a[ last ];
    i64 next = last - 1;
    while( comp( val, a[ next ] ) )
    {
        a[ last ] = a[ next ];
        last = next;
        --next;
    }
    a[ last ] = val;
}

template <typename T, typename C> MA_HD void insertion_sort( T* a, i64 first, i64 last, C comp )
{
    if( first == last )
        return;
    for( i64 i = first + 1; i != last; ++i )
    {
        if( comp( a[ i ], a[ first ] ) )
        {
            T val = a[ i ];
            for( i64 k = i; k > first; --k ) // move_backward(first, i, i+1)
                a[ k ] = a[ k - 1 ];
            a[ first ] = val;
        }
        else
            unguarded_linear_insert( a, i, comp );
    }
}

template <typename T, typename C> MA_HD void move_median_to_first( T* a, i64 result, i64 ia, i64 ib, i64 ic, C comp )
{
    if( comp( a[ ia ], a[ ib ] ) )
    {
        if( comp( a[ ib ], a[ ic ] ) )
            mswap( a[ result ], a[ ib ] );
        else if( comp( a[ ia ], a[ ic ] ) )
            mswap( a[ result ], a[ ic ] );
        else
            mswap( a[ result ], a[ ia ] );
    }
    else if( comp( a[ ia ], a[ ic ] ) )
        mswap( a[ result ], a[ ia ] );
    else if( comp( a[ ib ], a[ ic ] ) )
        mswap( a[ result ], a[ ic ] );
    else
        mswap( a[ result ], a[ ib ] );
}

template <typename T, typename C> MA_HD i64 unguarded_partition( T* a, i64 first, i64 last, i64 pivot, C comp )
{
    while( true )
    {
        while( comp( a[ first ], a[ pivot ] ) )
            ++first;
        --last;
        while( comp( a[ pivot ], a[ last ] ) )
            --last;
        if( !( first < last ) )
            return first;
        mswap( a[ first ], a[ last ] );
        ++first;
    }
}

// __partial_sort(first, last, last) == heap sort of the whole range
template <typename T, typename C> MA_HD void heap_sort_range( T* a, i64 first, i64 last, C comp )
{
    T* base = a + first;
    i64 len = last - first;
    make_heap( base, len, comp );
    while( len > 1 )
    {
        --len;
        pop_heap_to( base, len, len, comp );
    }
}

// std::sort(a, a+n, comp)
template <typename T, typename C> MA_HD_OUTLINE void sort( T* a, i64 n, C comp )
{
    if( n <= 0 )
        return;
    // __introsort_loop with an explicit stack for the right-hand recursion
    i64 stF[ 64 ], stL[ 64 ], stD[ 64 ];
    int sp = 0;
    i64 first = 0, last = n;
    i64 depth = 0;
    {
        u64 m = (u64)n; // __lg(n) * 2
        while( m >>= 1 )
            depth++;
        depth *= 2;
    }
    while( true )
    {
        while( last - first > 16 )
        {
            if( depth == 0 )
            {
                heap_sort_range( a, first, last, comp );
                break;
            }
            --depth;
            const i64 mid = first + ( last - first ) / 2;
            move_median_to_first( a, first, first + 1, mid, last - 1, comp );
            const i64 cut = unguarded_partition( a, first + 1, last, first, comp );
            // recurse on [cut,last) first (libstdc++ does so before looping on [first,cut)); the two
            // ranges are disjoint, so deferring it on a stack yields the same permutation
            stF[ sp ] = cut, stL[ sp ] = last, stD[ sp ] = depth;
            sp++;
            last = cut;
        }
        if( sp == 0 )
            break;
        sp--;
        first = stF[ sp ], last = stL[ sp ], depth = stD[ sp ];
    }
    // __final_insertion_sort
    if( n > 16 )
    {
        insertion_sort( a, (i64)0, (i64)16, comp );
        for( i64 i = 16; i != n; ++i )
            unguarded_linear_insert( a, i, comp );
    }
    else
        insertion_sort( a, (i64)0, n, comp );
}

// The part of std::sort that is left of a range [first, last) once __introsort_loop has come down to it with `depth`
// levels of its budget left -- the rest of the loop on that range, then the final insertion sort's share of it (the blocks
// the loop leaves are sorted in place, see wave_sort.h) -- by one thread.
template <typename T, typename C> MA_HD void finish_range( T* a, i64 first, i64 last, i64 depth, C comp )
{
    const i64 rangeFirst = first, rangeLast = last;
    int stF[ 40 ], stL[ 40 ], stD[ 40 ];
    int sp = 0;
    while( true )
    {
        while( last - first > 16 )
        {
            if( depth == 0 || sp == 40 )
            {
                heap_sort_range( a, first, last, comp ); // (sp == 40 cannot happen: the budget is 2 * lg n <= 40 levels)
                break;
            }
            --depth;
            const i64 mid = first + ( last - first ) / 2;
            move_median_to_first( a, first, first + 1, mid, last - 1, comp );
            const i64 cut = unguarded_partition( a, first + 1, last, first, comp );
            stF[ sp ] = (int)cut, stL[ sp ] = (int)last, stD[ sp ] = (int)depth;
            sp++;
            last = cut;
        }
        if( sp == 0 )
            break;
        sp--;
        first = stF[ sp ], last = stL[ sp ], depth = stD[ sp ];
    }
    insertion_sort( a, rangeFirst, rangeLast, comp );
}

// std::lower_bound
template <typename T, typename V, typename C> MA_HD i64 lower_bound( const T* a, i64 n, const V& val, C comp )
{
    i64 first = 0, len = n;
    while( len > 0 )
    {
        i64 half = len >> 1;
        i64 middle = first + half;
        if( comp( a[ middle ], val ) )
        {
            first = middle + 1;
            len = len - half - 1;
        }
        else
            len = half;
    }
    return first;
}
} // namespace ss
} // namespace ma
