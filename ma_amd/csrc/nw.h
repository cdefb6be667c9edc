// nw.h -- NeedlemanWunsch::execute_one / dynPrg / ksw / ksw_dual_ext (needlemanWunsch.cpp:82-877) split
// for the GPU into (1) DP job enumeration per harmonized seed set, (2) the batched ksw kernel and
// (3) a stitch pass that replays the same walk and consumes the ksw results in call order.
// The inputs of every kswcpp call depend only on the seeds and the reference window -- never on
// another call's result -- which is what makes the split legal.  Alignment::append /
// removeDangeling / larger / overlap and MappingQuality::execute follow alignment.cpp:10-98,240-296,
// alignment.h:659-735,819-845 and mappingQuality.cpp:11-131.
#pragma once
#include <type_traits>
#include "fm_device.h"
#include "stdsort.h"

namespace ma
{
struct NwParams
{
    u32 max_gap_area, padding, bandwidth_ext, min_bandwidth_gap, zdrop, sv_penalty;
    u32 match, mismatch, gap, extend;
    i32 kq, ke; // KswCppParam q / e as int8 (needlemanWunsch.cpp:400-416)
    u32 min_alignment_score, report_n_best, max_supplementary;
    double max_overlap_supplementary;
};

enum : u32
{
    MT_SEED = 0,
    MT_MATCH = 1,
    MT_MISS = 2,
    MT_INS = 3,
    MT_DEL = 4
};

// One DP job = one kswcpp_dispatch call. Sequences are addressed relative to the read and to the
// extracted reference window [win_begin, ...) on the doubled text; `rev` = both reversed in place
// (needlemanWunsch.cpp:252-259, 558-571).
struct DpJob
{
    u64 win_begin; // absolute position of window offset 0 on T.revcomp(T)
    u64 read_off; // offset of the read in the batch's base array
    u32 q_from, q_to, r_from, r_to; // [from,to) on query / window
    i32 w, zdrop, flag;
    u32 rev;
};

struct AlnHeader // per harmonized set
{
    u64 begin_ref, end_ref, begin_q, end_q;
    i64 score;
    u64 length;
    u64 ops_off; // into the ops pool
    u32 n_ops, ops_cap;
    u32 soc_index;
    u32 secondary, supplementary;
    double mapq;
};

// packed op: type in the top 4 bits, length below
MA_HD u64 op_pack( u32 type, u64 len )
{
    return ( (u64)type << 60 ) | len;
}
MA_HD u32 op_type( u64 o )
{
    return (u32)( o >> 60 );
}
MA_HD u64 op_len( u64 o )
{
    return o & 0x0fffffffffffffffull;
}

struct AlnBuilder
{
    AlnHeader* h;
    u64* ops; // this alignment's region
    u32* err;
    // the LAST entry (ops[ n_ops - 1 ]) lives here until another one follows it or aln_flush is called: every append looks at
    // it and most appends change it, and as a read-modify-write of global memory that was a memory round trip per append in
    // the middle of the walk's dependency chain
    u64 last = 0;
};
MA_HD void aln_flush( AlnBuilder& A )
{
    if( A.h->n_ops != 0 )
        A.ops[ A.h->n_ops - 1 ] = A.last;
}

MA_HD u64 indel_cost( const NwParams& P, u64 len )
{
    const u64 c = (u64)P.extend * len + (u64)P.gap;
    return c < (u64)P.sv_penalty ? c : (u64)P.sv_penalty;
}

// Alignment::append (alignment.cpp:10-98)
MA_HD void aln_append( const NwParams& P, AlnBuilder& A, u32 type, u64 size )
{
    if( size == 0 )
        return;
    AlnHeader& h = *A.h;
    if( type == MT_SEED || type == MT_MATCH )
    {
        h.score += (i64)( (u64)P.match * size );
        h.end_ref += size;
        h.end_q += size;
    }
    else if( type == MT_MISS )
    {
        h.score -= (i64)( (u64)P.mismatch * size );
        h.end_ref += size;
        h.end_q += size;
    }
    else
    {
        if( type == MT_INS )
            h.end_q += size;
        else
            h.end_ref += size;
        if( h.n_ops != 0 && op_type( A.last ) == type )
        {
            // the reference takes the entry off, adds its length to the new one and puts that back (the entry before it is of
            // another type: equal neighbours never exist): same thing in place
            const u64 prev = op_len( A.last );
            size += prev;
            h.score += (i64)indel_cost( P, prev );
            h.score -= (i64)indel_cost( P, size );
            A.last = op_pack( type, size );
            h.length += size - prev;
            return;
        }
        h.score -= (i64)indel_cost( P, size );
    }
    if( h.n_ops != 0 && op_type( A.last ) == type )
        A.last = op_pack( type, op_len( A.last ) + size );
    else if( h.n_ops < h.ops_cap )
    {
        if( h.n_ops != 0 )
            A.ops[ h.n_ops - 1 ] = A.last;
        A.last = op_pack( type, size );
        h.n_ops++;
    }
    else
        *A.err |= MA_ERR_OPS_OVERFLOW;
    h.length += size;
}

// Alignment::removeDangeling (alignment.cpp:240-296); returns the number of ops dropped at the front
MA_HD u32 aln_remove_dangling( const NwParams& P, AlnBuilder& A )
{
    AlnHeader& h = *A.h;
    if( h.n_ops == 0 )
        return 0;
    u32 front = 0;
    while( front < h.n_ops && ( op_type( A.ops[ front ] ) == MT_DEL || op_type( A.ops[ front ] ) == MT_INS ) )
    {
        const u64 len = op_len( A.ops[ front ] );
        if( op_type( A.ops[ front ] ) == MT_DEL )
            h.begin_ref += len;
        else
            h.begin_q += len;
        h.score += (i64)indel_cost( P, len );
        h.length -= len;
        front++;
    }
    while( h.n_ops > front &&
           ( op_type( A.ops[ h.n_ops - 1 ] ) == MT_DEL || op_type( A.ops[ h.n_ops - 1 ] ) == MT_INS ) )
    {
        const u64 len = op_len( A.ops[ h.n_ops - 1 ] );
        if( op_type( A.ops[ h.n_ops - 1 ] ) == MT_DEL )
            h.end_ref -= len;
        else
            h.end_q -= len;
        h.score += (i64)indel_cost( P, len );
        h.length -= len;
        h.n_ops--;
    }
    return front;
}

// ksw result as seen by the stitch pass
struct KswResult
{
    i32 max_q, max_t;
    const u32* cigar;
    u32 n_cigar;
    // the one-alignment-per-wavefront kernel hands over the first four entries with the result (most gap fills of a long read
    // have one to three): read with the job record, 64 jobs at a time, instead of in a memory round trip of their own
    u32 first[ 4 ];
    bool cached = false;
    MA_HD u32 at( u32 i ) const
    {
        if( cached && i < 4 )
            return i < 2 ? ( i == 0 ? first[ 0 ] : first[ 1 ] ) : ( i == 2 ? first[ 2 ] : first[ 3 ] );
        return cigar[ i ];
    }
};

// ---- the walk ---------------------------------------------------------------------------------
// Sink interface (duck-typed):
//   void job( q_from,q_to,r_from,r_to, w, zdrop, flag, rev )        -- enumeration
//   KswResult next()                                                -- stitch: result of the next job
//   static const bool STITCH
struct NwWindow
{
    u64 begin_ref, end_ref; // extracted window [begin_ref, end_ref)
    bool valid;
};

// bounding box, bridging test, padding and contig clamp (needlemanWunsch.cpp:654-733), bLocal == false
MA_HD NwWindow nw_window( const IndexView& X, const NwParams& P, const ma_seed* S, u32 n )
{
    NwWindow W;
    W.valid = false;
    W.begin_ref = W.end_ref = 0;
    if( n == 0 )
        return W;
    u64 beginRef = (u64)S[ 0 ].r_start, endRef = (u64)S[ n - 1 ].r_start + (u64)S[ n - 1 ].len;
    for( u32 i = 0; i < n; i++ )
    {
        const u64 er = (u64)S[ i ].r_start + (u64)S[ i ].len;
        if( endRef < er )
            endRef = er;
        if( beginRef > (u64)S[ i ].r_start )
            beginRef = (u64)S[ i ].r_start;
    }
    if( beginRef >= endRef || bridging( X, beginRef, endRef - beginRef + 1 ) )
        return W;
    const i64 oldContig = seq_id_or_rev( X, beginRef );
    beginRef -= (u64)P.padding;
    if( beginRef > endRef )
        beginRef = 0;
    endRef += (u64)P.padding;
    if( endRef >= X.n )
        endRef = X.n - 1;
    if( seq_id_or_rev( X, beginRef ) != oldContig )
        beginRef = start_of_seq_or_rev( X, oldContig );
    if( seq_id_or_rev( X, endRef ) != oldContig )
        endRef = end_of_seq_or_rev( X, oldContig ) - 1;
    W.begin_ref = beginRef;
    W.end_ref = endRef;
    W.valid = true;
    return W;
}

// SINK::WAVE (device only): ONE alignment per wavefront -- all lanes walk the seeds and cigars in step (the same loads, the
// same appends), and the base comparisons of an M run, the bulk of a long read's walk, are spread over the lanes (match_run)
template <typename S, typename = void> struct sink_is_wave
{
    static const bool value = false;
};
template <typename S> struct sink_is_wave<S, decltype( (void)S::WAVE )>
{
    static const bool value = S::WAVE;
};
template <typename SINK> struct NwWalk
{
    const IndexView& X;
    const NwParams& P;
    SINK& sink;
    const uint8_t* Q; // read codes
    u64 winBegin; // absolute position of window offset 0
    AlnBuilder A; // only used when SINK::STITCH
    u64 qLenTotal = 0; // length of the read (set by run)

    MA_HD u32 qb( u64 i ) const
    {
        return Q[ i ];
    }
    MA_HD u32 rb( u64 i ) const
    {
        return text_base( X, winBegin + i );
    }
    // An M run of a cigar as match / missmatch entries (needlemanWunsch.cpp:573-620 appends base by base; Alignment::append
    // merges equal neighbours, so appending a whole run of equal bases at once gives the same entries and the same score).
    // The header and the last op live in global memory: one append per RUN instead of per base, the read through an aligned
    // 8-byte window and the reference through the pac byte that holds four bases (50 kb reads: this loop was most of k_stitch).
    MA_HD void match_run( u64 qPos, u64 rPos, u32 amount )
    {
#if defined( __HIP_DEVICE_COMPILE__ )
        if( sink_is_wave<SINK>::value )
        {
            // 64 bases per trip: one coalesced load of the read and of the packed reference, the mismatches as a ballot, the
            // runs of equal bits appended one by one (appends of equal neighbours merge: the entries of the serial loop)
            const u32 lane = threadIdx.x & 63;
            for( u32 k0 = 0; k0 < amount; k0 += 64 )
            {
                const u32 k = k0 + lane, nv = amount - k0 < 64u ? amount - k0 : 64u;
                // (the bases out of 1 KB windows of the read and the reference in LDS instead of these two loads: measured, slower
                // -- 34 -> 42 ms for 20 k alignments of 50 kb: the kernel is bound by the instructions of the walk, not by these loads)
                bool miss = false;
                if( k < amount )
                    miss = qb( qPos + k ) != rb( rPos + k );
                const unsigned long long mm = __ballot( miss );
                u32 pos = 0;
                while( pos < nv )
                {
                    const u32 bit = (u32)( mm >> pos ) & 1u;
                    const unsigned long long other = ( bit ? ~mm : mm ) >> pos; // set where the type changes
                    u32 len = other ? (u32)__builtin_ctzll( other ) : 64u - pos;
                    if( len > nv - pos )
                        len = nv - pos;
                    aln_append( P, A, bit ? MT_MISS : MT_MATCH, len );
                    pos += len;
                }
            }
            return;
        }
#endif
        u32 curType = MT_MATCH;
        u64 curLen = 0;
        uintptr_t qWordAt = ~(uintptr_t)0;
        u64 qWord = 0, pacAt = ~0ull;
        u32 pacByte = 0;
        for( u32 k = 0; k < amount; k++ )
        {
            const uintptr_t qAddr = (uintptr_t)( Q + qPos + k );
            if( ( qAddr & ~(uintptr_t)7 ) != qWordAt )
            {
                qWordAt = qAddr & ~(uintptr_t)7;
                // the window never starts before the allocation of the reads (allocations are at least 256-byte aligned) but its
                // end may lie behind the last base of a caller-owned array: then it is put together byte by byte
                if( qWordAt + 8 <= (uintptr_t)( Q + qLenTotal ) )
                    qWord = *(const u64*)qWordAt;
                else
                {
                    qWord = 0;
                    for( uintptr_t a = qAddr; a < (uintptr_t)( Q + qLenTotal ); a++ )
                        qWord |= (u64)( *(const uint8_t*)a ) << ( 8 * ( a & 7 ) );
                }
            }
            const u32 q = (u32)( qWord >> ( 8 * ( qAddr & 7 ) ) ) & 0xffu;
            const u64 p = winBegin + rPos + k;
            const bool comp = p >= X.F;
            const u64 f = comp ? X.n - 1 - p : p;
            if( ( f >> 2 ) != pacAt )
            {
                pacAt = f >> 2;
                pacByte = X.pac[ pacAt ];
            }
            u32 r = ( pacByte >> ( ( ~(u32)f & 3 ) << 1 ) ) & 3;
            if( comp )
                r = 3u - r;
            const u32 type = q == r ? MT_MATCH : MT_MISS;
            if( type != curType && curLen != 0 )
            {
                aln_append( P, A, curType, curLen );
                curLen = 0;
            }
            curType = type;
            curLen++;
        }
        aln_append( P, A, curType, curLen );
    }
    MA_HD void app( u32 type, u64 size )
    {
        if( SINK::STITCH )
            aln_append( P, A, type, size );
    }

    // NeedlemanWunsch::ksw (needlemanWunsch.cpp:82-169)
    MA_HD void gap_small( u64 fromQ, u64 toQ, u64 fromR, u64 toR )
    {
        const i32 qlen = (i32)( toQ - fromQ ), tlen = (i32)( toR - fromR );
        i32 w = (i32)P.min_bandwidth_gap;
        const i32 d = tlen - qlen < 0 ? qlen - tlen : tlen - qlen;
        if( d + 10 > w )
            w = d + 10;
        if( !SINK::STITCH )
        {
            sink.job( (u32)fromQ, (u32)toQ, (u32)fromR, (u32)toR, w, -1, 0, 0 );
            return;
        }
        const KswResult R = sink.next( );
        u64 qPos = fromQ, rPos = fromR;
        for( u32 i = 0; i < R.n_cigar; i++ )
        {
            const u32 sym = R.at( i ) & 0xf, amount = R.at( i ) >> 4;
            if( sym == 0 )
            {
                match_run( qPos, rPos, amount );
                qPos += amount;
                rPos += amount;
            }
            else if( sym == 1 )
            {
                app( MT_INS, amount );
                qPos += amount;
            }
            else if( sym == 2 )
            {
                app( MT_DEL, amount );
                rPos += amount;
            }
        }
        app( MT_DEL, toQ - qPos ); // (sic) needlemanWunsch.cpp:167-168
        app( MT_INS, toR - rPos );
    }

    // NeedlemanWunsch::ksw_dual_ext (needlemanWunsch.cpp:239-497)
    MA_HD void gap_dual( u64 fromQ, u64 toQ, u64 fromR, u64 toR )
    {
        if( !SINK::STITCH )
        {
            sink.job( (u32)fromQ, (u32)toQ, (u32)fromR, (u32)toR, (i32)P.bandwidth_ext, (i32)P.zdrop, 0x40, 0 );
            sink.job( (u32)fromQ, (u32)toQ, (u32)fromR, (u32)toR, (i32)P.bandwidth_ext, (i32)P.zdrop,
                      0x40 | 0x02 | 0x80, 1 );
            return;
        }
        const KswResult Lz = sink.next( );
        const KswResult Rz = sink.next( );
        u64 qCenter = ( fromQ + (u64)(i64)Lz.max_q + ( toQ - (u64)(i64)Rz.max_q - 1 ) ) / 2;
        qCenter = mmax( fromQ, mmin( toQ, qCenter ) );
        u64 rCenter = ( fromR + (u64)(i64)Lz.max_t + ( toR - (u64)(i64)Rz.max_t - 1 ) ) / 2;
        rCenter = mmax( fromR, mmin( toR, rCenter ) );
        u64 qPos = fromQ, rPos = fromR;
        if( rPos != rCenter && qPos != qCenter )
            for( u32 i = 0; i < Lz.n_cigar; ++i )
            {
                const u32 sym = Lz.at( i ) & 0xf;
                u32 amount = Lz.at( i ) >> 4;
                if( sym == 0 )
                {
                    if( qPos + amount > qCenter )
                        amount = (u32)( qCenter - qPos );
                    if( rPos + amount > rCenter )
                        amount = (u32)( rCenter - rPos );
                    match_run( qPos, rPos, amount );
                    qPos += amount;
                    rPos += amount;
                }
                else if( sym == 1 )
                {
                    if( qPos + amount > qCenter )
                        amount = (u32)( qCenter - qPos );
                    app( MT_INS, amount );
                    qPos += amount;
                }
                else if( sym == 2 )
                {
                    if( rPos + amount > rCenter )
                        amount = (u32)( rCenter - rPos );
                    app( MT_DEL, amount );
                    rPos += amount;
                }
                if( rPos == rCenter )
                    break;
                if( qPos == qCenter )
                    break;
            }
        u64 rPosRight = toR - (u64)(i64)Rz.max_t - 1;
        u64 qPosRight = toQ - (u64)(i64)Rz.max_q - 1;
        u32 notUnrolled = 0;
        u32 lastType = MT_SEED;
        u32 i = 0;
        for( ; i < Rz.n_cigar; ++i )
        {
            if( rPosRight >= rCenter && qPosRight >= qCenter )
                break;
            const u32 sym = Rz.at( i ) & 0xf;
            u32 amount = Rz.at( i ) >> 4;
            if( sym == 0 )
            {
                if( rPosRight + amount >= rCenter && qPosRight + amount >= qCenter )
                {
                    if( rPosRight < rCenter && ( qPosRight >= qCenter || rCenter - rPosRight > qCenter - qPosRight ) )
                    {
                        notUnrolled = amount - (u32)( rCenter - rPosRight );
                        amount = (u32)( rCenter - rPosRight );
                    }
                    else
                    {
                        notUnrolled = amount - (u32)( qCenter - qPosRight );
                        amount = (u32)( qCenter - qPosRight );
                    }
                }
                qPosRight += amount;
                rPosRight += amount;
                lastType = MT_MATCH;
            }
            else if( sym == 1 )
            {
                if( qPosRight + amount > qCenter && rPosRight >= rCenter )
                {
                    notUnrolled = amount - (u32)( qCenter - qPosRight );
                    amount = (u32)( qCenter - qPosRight );
                }
                qPosRight += amount;
                lastType = MT_INS;
            }
            else if( sym == 2 )
            {
                if( rPosRight + amount > rCenter && qPosRight >= qCenter )
                {
                    notUnrolled = amount - (u32)( rCenter - rPosRight );
                    amount = (u32)( rCenter - rPosRight );
                }
                rPosRight += amount;
                lastType = MT_DEL;
            }
        }
        const u64 kq = (u64)(i64)P.kq, ke = (u64)(i64)P.ke;
        u64 mmPen = ( qPosRight - qPos ) >= ( rPosRight - rPos ) ? ( qPosRight - qPos ) - ( rPosRight - rPos )
                                                               : ( rPosRight - rPos ) - ( qPosRight - qPos );
        mmPen *= (u64)P.mismatch;
        const u64 uiM = mmin( qPosRight - qPos, rPosRight - rPos );
        if( uiM > 0 )
            mmPen += kq + ke * uiM;
        u64 gapPen = 0;
        if( qPosRight - qPos > 0 )
            gapPen += kq + ke * qPosRight - qPos; // (sic) operator precedence, needlemanWunsch.cpp:413-416
        if( rPosRight - rPos > 0 )
            gapPen += kq + ke * rPosRight - rPos;
        if( mmPen < gapPen && qPos < qPosRight && rPos < rPosRight )
        {
            // base by base in the reference (needlemanWunsch.cpp:420-427); as runs the entries are the same
            const u64 m = mmin( qPosRight - qPos, rPosRight - rPos );
            if( SINK::STITCH )
                match_run( qPos, rPos, (u32)m );
            qPos += m;
            rPos += m;
        }
        app( MT_INS, qPosRight - qPos );
        app( MT_DEL, rPosRight - rPos );
        if( lastType == MT_MATCH )
            match_run( qPosRight, rPosRight, notUnrolled );
        else
            app( lastType, notUnrolled );
        if( lastType == MT_MATCH )
        {
            qPosRight += notUnrolled;
            rPosRight += notUnrolled;
        }
        else if( lastType == MT_INS )
            qPosRight += notUnrolled;
        else if( lastType == MT_DEL )
            rPosRight += notUnrolled;
        for( ; i < Rz.n_cigar; ++i )
        {
            const u32 sym = Rz.at( i ) & 0xf, amount = Rz.at( i ) >> 4;
            if( sym == 0 )
            {
                match_run( qPosRight, rPosRight, amount );
                qPosRight += amount;
                rPosRight += amount;
            }
            else if( sym == 1 )
            {
                app( MT_INS, amount );
                qPosRight += amount;
            }
            else if( sym == 2 )
            {
                app( MT_DEL, amount );
                rPosRight += amount;
            }
        }
    }

    // NeedlemanWunsch::dynPrg (needlemanWunsch.cpp:499-622)
    MA_HD void dyn_prg( u64 fromQ, u64 toQ, u64 fromR, u64 toR, bool bLocalBeginning, bool bLocalEnd )
    {
        if( toR <= fromR )
            if( toQ <= fromQ )
                return;
        if( toQ <= fromQ )
        {
            app( MT_DEL, toR - fromR );
            return;
        }
        if( toR <= fromR )
        {
            app( MT_INS, toQ - fromQ );
            return;
        }
        if( !bLocalBeginning && !bLocalEnd )
        {
            if( toQ - fromQ > (u64)P.max_gap_area || toR - fromR > (u64)P.max_gap_area )
                gap_dual( fromQ, toQ, fromR, toR );
            else
                gap_small( fromQ, toQ, fromR, toR );
            return;
        }
        const bool bReverse = bLocalBeginning;
        if( !SINK::STITCH )
        {
            sink.job( (u32)fromQ, (u32)toQ, (u32)fromR, (u32)toR, (i32)P.bandwidth_ext, (i32)P.zdrop,
                      bReverse ? ( 0x40 | 0x02 | 0x80 ) : 0x40, bReverse ? 1 : 0 );
            return;
        }
        const KswResult R = sink.next( );
        u64 qPos = fromQ, rPos = fromR;
        if( bReverse )
        {
            rPos = toR - (u64)(i64)R.max_t - 1;
            qPos = toQ - (u64)(i64)R.max_q - 1;
        }
        for( u32 i = 0; i < R.n_cigar; i++ )
        {
            const u32 sym = R.at( i ) & 0xf, amount = R.at( i ) >> 4;
            if( sym == 0 )
            {
                match_run( qPos, rPos, amount );
                qPos += amount;
                rPos += amount;
            }
            else if( sym == 1 )
            {
                app( MT_INS, amount );
                qPos += amount;
            }
            else if( sym == 2 )
            {
                app( MT_DEL, amount );
                rPos += amount;
            }
        }
        if( bReverse )
        {
            const u64 byR = toR - (u64)(i64)R.max_t - 1, byQ = toQ - (u64)(i64)R.max_q - 1;
            A.h->begin_ref += byR;
            A.h->end_ref += byR;
            A.h->begin_q += byQ;
            A.h->end_q += byQ;
        }
    }

    // NeedlemanWunsch::execute_one (needlemanWunsch.cpp:625-877) after the window was fixed
    // seed k of the set: (q_start, r_start, len)
    template <typename T = SINK> MA_HD typename std::enable_if<sink_is_wave<T>::value>::type sd( const ma_seed* S, u32 n, u32 k, u64& q, u64& r, u64& l )
    {
        sink.seed( S, n, k, q, r, l );
    }
    template <typename T = SINK> MA_HD typename std::enable_if<!sink_is_wave<T>::value>::type sd( const ma_seed* S, u32, u32 k, u64& q, u64& r, u64& l )
    {
        q = (u64)S[ k ].q_start, r = (u64)S[ k ].r_start, l = (u64)S[ k ].len;
    }
    MA_HD void run( const ma_seed* S, u32 n, u64 qlen, const NwWindow& W )
    {
        qLenTotal = qlen;
        const u64 beginRef = W.begin_ref, endRef = W.end_ref;
        u64 q0, r0, l0;
        sd( S, n, 0, q0, r0, l0 );
        dyn_prg( 0, q0, 0, r0 - beginRef, true, false );
        u64 endLastQ = q0 + l0;
        u64 endLastR = r0 + l0 - beginRef;
        app( MT_SEED, l0 );
        for( u32 k = 1; k < n; k++ )
        {
            u64 sq, sr, sl;
            sd( S, n, k, sq, sr, sl );
            if( sl == 0 )
                continue;
            u64 ovQ = endLastQ - sq;
            if( sq > endLastQ )
                ovQ = 0;
            u64 ovR = endLastR - ( sr - beginRef );
            if( sr > endLastR + beginRef )
                ovR = 0;
            const u64 overlap = mmax( ovQ, ovR );
            if( sl > overlap )
            {
                dyn_prg( endLastQ, sq, endLastR, sr - beginRef, false, false );
                if( ovQ > ovR )
                    app( MT_DEL, ovQ - ovR );
                if( ovR > ovQ )
                    app( MT_INS, ovR - ovQ );
                app( MT_SEED, sl - overlap );
                if( sq + sl > endLastQ )
                    endLastQ = sq + sl;
                if( sr + sl > endLastR + beginRef )
                    endLastR = sr + sl - beginRef;
            }
        }
        dyn_prg( endLastQ, qlen - 1, endLastR, endRef - beginRef - 1, false, true );
        if( SINK::STITCH )
        {
            aln_flush( A );
            const u32 front = aln_remove_dangling( P, A );
            if( front > 0 )
            {
                for( u32 i = front; i < A.h->n_ops; i++ )
                    A.ops[ i - front ] = A.ops[ i ];
                A.h->n_ops -= front;
            }
        }
    }
};

// Alignment::larger (alignment.h:819-845)
MA_HD bool aln_larger( const AlnHeader& a, const AlnHeader& b )
{
    u32 uiA = 0, uiB = 0;
    if( b.secondary )
        uiB = 2;
    if( b.supplementary )
        uiB = 1;
    if( a.secondary )
        uiA = 2;
    if( a.supplementary )
        uiA = 1;
    if( uiA != uiB )
        return uiA < uiB;
    if( a.score == b.score )
        return a.soc_index < b.soc_index;
    return a.score > b.score;
}

// Alignment::overlap (alignment.h:659-735)
MA_HD double aln_overlap( const AlnHeader& a, const u64* ao, const AlnHeader& o, const u64* oo )
{
    const u64 uiS = mmax( a.begin_q, o.begin_q ), uiE = mmin( a.end_q, o.end_q );
    if( uiS >= uiE )
        return 0;
    u64 ov = 0;
    u32 i = 0, j = 0;
    u64 qp = a.begin_q, qo = o.begin_q;
    while( qp + op_len( ao[ i ] ) < uiS )
    {
        if( op_type( ao[ i ] ) != MT_DEL )
            qp += op_len( ao[ i ] );
        i++;
    }
    while( qo + op_len( oo[ j ] ) < uiS )
    {
        if( op_type( oo[ j ] ) != MT_DEL )
            qo += op_len( oo[ j ] );
        j++;
    }
    while( qp < uiE && qo < uiE && i < a.n_ops && j < o.n_ops )
    {
        u64 ql = 0, qlo = 0;
        if( op_type( ao[ i ] ) != MT_DEL )
            ql = op_len( ao[ i ] );
        if( op_type( oo[ j ] ) != MT_DEL )
            qlo = op_len( oo[ j ] );
        const u64 s_in = mmax( mmax( qp, qo ), uiS );
        const u64 e_in = mmin( mmin( qp + ql, qo + qlo ), uiE );
        u64 cur = 0;
        if( s_in < e_in )
            cur = e_in - s_in;
        if( op_type( ao[ i ] ) != MT_INS && op_type( oo[ j ] ) != MT_INS )
            ov += cur;
        if( qp + ql < qo + qlo )
        {
            qp += ql;
            i++;
        }
        else
        {
            qo += qlo;
            j++;
        }
    }
    const u64 sz = mmin( a.end_q - a.begin_q, o.end_q - o.begin_q );
    return (double)ov / (double)sz;
}

// Per read: sort the alignments of NeedlemanWunsch::execute (needlemanWunsch.h:131-132), then
// MappingQuality::execute (mappingQuality.cpp:11-131). `order` (n) receives the NW order as indices
// into hdr[]; `mq_order` the MappingQuality output order, returns its length.
struct ByLarger
{
    const AlnHeader* h;
    MA_HD bool operator( )( u32 a, u32 b ) const
    {
        return aln_larger( h[ a ], h[ b ] );
    }
};
struct ByScore
{
    const AlnHeader* h;
    MA_HD bool operator( )( u32 a, u32 b ) const
    {
        return h[ a ].score > h[ b ].score;
    }
};

// nw_sort = false: the alignments were handed in (ma_batch_set_alignments) in the order their NeedlemanWunsch left them
MA_HD u32 finish_read( const NwParams& P, AlnHeader* hdr, const u64* opsPool, u32 n, u64 qlen, u32* order, u32* mq_order,
                       bool nw_sort = true )
{
    for( u32 i = 0; i < n; i++ )
        order[ i ] = i;
    if( nw_sort )
        ss::sort( order, (i64)n, ByLarger{ hdr } );
    if( n == 0 )
        return 0;
    for( u32 i = 0; i < n; i++ )
        mq_order[ i ] = order[ i ];
    ss::sort( mq_order, (i64)n, ByScore{ hdr } );
    AlnHeader& first = hdr[ mq_order[ 0 ] ];
    first.secondary = 0;
    u32 nSupp = 0;
    for( u32 i = 1; i < n; i++ )
    {
        AlnHeader& a = hdr[ mq_order[ i ] ];
        a.mapq = 0.0;
        if( nSupp < P.max_supplementary &&
            aln_overlap( a, opsPool + a.ops_off, first, opsPool + first.ops_off ) < P.max_overlap_supplementary )
        {
            a.supplementary = 1;
            a.secondary = 0;
            nSupp++;
        }
        else
        {
            a.supplementary = 0;
            a.secondary = 1;
        }
    }
    if( n - nSupp >= 2 )
    {
        u32 i = 1;
        const AlnHeader* second = nullptr;
        while( second == nullptr || second->supplementary )
        {
            second = &hdr[ mq_order[ i ] ];
            i++;
        }
        if( first.score == 0 )
            first.mapq = 0;
        else
            first.mapq = (double)( first.score - second->score ) / (double)first.score;
    }
    else
        first.mapq = (double)first.score / (double)( (u64)P.match * qlen );
    u32 nSeeds = 0;
    for( u32 k = 0; k < first.n_ops; k++ )
        if( op_type( opsPool[ first.ops_off + k ] ) == MT_SEED )
            nSeeds++;
    if( nSeeds <= 1 )
        first.mapq /= 2;
    if( (double)first.score >= (double)( (u64)P.match * qlen ) * 0.8 && n >= 3 )
        first.mapq *= 2;
    if( first.mapq > 1 )
        first.mapq = 1;
    if( nSupp > 0 )
    {
        for( u32 i = 1; i < n; i++ )
            if( hdr[ mq_order[ i ] ].supplementary )
                hdr[ mq_order[ i ] ].mapq = first.mapq;
        ss::sort( mq_order, (i64)n, ByLarger{ hdr } );
    }
    u32 m = n;
    if( P.report_n_best != 0 && m > P.report_n_best + nSupp )
        m = P.report_n_best + nSupp;
    u32 w = 0;
    for( u32 i = 0; i < m; i++ )
        if( !( hdr[ mq_order[ i ] ].score < (i64)P.min_alignment_score ) )
            mq_order[ w++ ] = mq_order[ i ];
    return w;
}
} // namespace ma
