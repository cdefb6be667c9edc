// seeding.h -- BinarySeeding (binarySeeding.h:55-452, binarySeeding.cpp:32-178) as a per-read state
// machine that performs exactly one FMIndex::extend_backward per step, so that the 64 reads of a
// wavefront advance in lockstep through one shared "load two occ blocks + popcount" code path
// regardless of which extension phase each read is in.
#pragma once
#include "fm_device.h"

namespace ma
{
struct SeedParams
{
    u32 technique; // 0 maxSpan, 1 SMEMs, 2 MEMs (mems_from below; the state machine serves 0 and 1)
    u32 min_seed_len; // MEMs only: xMinSeedLength (binarySeeding.h:568)
    u32 min_amb, max_amb;
    u32 min_seed_size_drop;
    u32 disable_heuristics;
    double rel_min_seed_size_amount;
    u64 genome_size_disable;
    const uint8_t* window_begin = nullptr; // device kernels whose reads stay in HBM: the reads array, for seed_qbyte's 16-byte window
    const uint8_t* window_end = nullptr; // (null: off)
    u32 smem_compact = 0; // SMEM pending lists hold 16-byte entries (smem_put / smem_get): reads < 2^22 bases, text < 2^35
    u32 smem_merge = 0; // SMEM backward phase: an entry whose interval equals that of the entry pushed before it is not kept (seed_apply)
};

// Scratch a read needs while it is being seeded (lives in HBM, one slot per resident lane).
struct SeedScratch
{
    ma_segment* stage; // staged output segments of the read in flight (capacity seg_cap)
    u32 seg_cap;
    ma_segment* smem_a; // SMEM pending lists (capacity smem_cap each)
    ma_segment* smem_b;
    u32 smem_cap;
    u32 drop_div; // SeedParams::min_seed_size_drop (seed_emit keeps the running sum seed_finish needs)
    u32* stack; // 2 * MA_SEED_STACK words: interval stack of procesInterval (kept out of SeedLane so that the lane
                // state stays in registers; a dynamically indexed member would put the whole struct in scratch memory)
    // Round 6: the HEADS of the two SMEM lists in LDS (16-byte entries only; kernels whose reads stay in HBM).  A backward list holds one
    // to three entries once equal intervals are merged, the forward list about a dozen: with the first lds_n entries of each list in
    // LDS a backward step neither reads nor writes a list entry in HBM (50 kb x 10 k reads, SMEM tasks: 1.42 TB of fabric traffic per
    // step for 0.35 TB of occ blocks before).  Entry e of list w (0 = smem_a, 1 = smem_b) of this lane: lds[ 2 (w lds_n + e) lds_stride ].
    // lds_n <= smem_cap: an entry past the capacity is never written (the batch goes to the fallback), so none may be read from LDS.
    u64* lds = nullptr;
    u32 lds_n = 0, lds_stride = 0;
};

enum SeedPhase : u32
{
    PH_NEW_CENTER = 0,
    PH_P1_RIGHT = 1,
    PH_P1_LEFT = 2,
    PH_P2_LEFT = 3,
    PH_P2_RIGHT = 4,
    PH_SMEM_FWD = 5,
    PH_SMEM_BWD = 6,
    PH_DONE = 7
};

#define MA_SEED_STACK 40

struct SeedLane
{
    // read
    const uint8_t* q;
    u32 qlen;
    // interval stack of procesInterval (binarySeeding.cpp:32-84): left parts recurse, right parts iterate
    // (entries live in SeedScratch::stack)
    u32 sp;
    u32 aS, aN; // area currently processed
    // extension state
    u32 phase;
    u32 center, i, start, end;
    u32 s1_start, s1_end; // first (right-then-left) segment of the center, for the duplicate test
    i64 ik[ 3 ];
    // K-mer table entry of the centre's LEFT run (second pass), looked up together with the right run's (seed_center): the
    // two gathers share one memory round trip.  eL1 == 0: none
    u64 eL0, eL1;
    // reads in HBM: the bases around the centre the transitions need -- q[c-K], q[c-1], q[c], q[c+1], q[c+K], 3 bits each,
    // bit 15 = valid -- out of the block seed_center loaded (otherwise every transition is a memory round trip of its own)
    u32 cb;
    // SMEM state
    u32 nPrev, nCurr, jPrev; // list sizes / cursor
    u32 curQStart, curQSize; // prev[jPrev] of the extension in flight (its interval is in ik): read once, in seed_prepare
    u32 bHaveOne, retS, retE;
    u32 flip; // which of smem_a/smem_b is "prev"
    i64 lastK, lastS; // interval of the entry pushed last onto curr: the twin test of smem_merge
    u64 first0, first1; // the entry pushed FIRST onto curr (16-byte form): what the next position starts with (seed_try)
    u64 nxt0, nxt1; // prev[ jPrev ] of the NEXT backward step (16-byte entries only), requested one step ahead: the entry
    u32 nxtOk; //      arrives while this step's occ blocks are on their way, instead of in a round trip of its own
    // output
    u32 nseg;
    u32 err;
    u64 drop_sum; // sum over emitted segments of q_size / min_seed_size_drop (numSeedsLarger, segment.h:278-289)
    // counters
    u32 steps, blocks;
    // task mode (one lane per AREA instead of per read, pipeline.hip k_seed_tasks): after the area's centre has been
    // extended the lane stops and reports the two remaining areas instead of walking them itself
    u32 task_mode;
    u32 childS[ 2 ], childN[ 2 ]; // [0] left of the covered interval, [1] right of it; N == 0: none
    // 16 read bytes in registers for reads that stay in HBM (seed_qbyte): a lane walks its read base by base, and one
    // byte load per step is one 64-byte fabric request per step once the line has left L2 -- a third of the requests
    // k_seed issued on 10 kb reads (431 GB of fetches for 299 GB of occ blocks)
    u32 qwinLo;
    u32 qw0, qw1, qw2, qw3;
};

// base i of the lane's read.  P.window_begin / window_end (wave-uniform; null = off: the read is in LDS, or host code)
// delimit the reads array: the 16 bytes are taken aligned, or flush with an end of the array where an aligned block
// would leave it.
MA_HD void seed_qwin_load( SeedLane& L, const SeedParams& P, u32 i )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    const uintptr_t a = (uintptr_t)( L.q + i );
    uintptr_t base = a & ~(uintptr_t)15;
    uint4 v;
    if( base + 16 > (uintptr_t)P.window_end || base < (uintptr_t)P.window_begin )
    {
        // first / last bytes of the reads array: flush with its end, byte loads
        if( base + 16 > (uintptr_t)P.window_end )
            base = (uintptr_t)P.window_end - 16;
        if( base < (uintptr_t)P.window_begin )
            base = (uintptr_t)P.window_begin;
        __builtin_memcpy( &v, (const void*)base, 16 );
    }
    else
        v = *(const uint4*)base; // one global_load_dwordx4
    L.qw0 = v.x, L.qw1 = v.y, L.qw2 = v.z, L.qw3 = v.w;
    L.qwinLo = i - (u32)( a - base );
#endif
}
template <bool WIN> MA_HD u32 seed_qbyte( SeedLane& L, const SeedParams& P, u32 i )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    if( WIN && P.window_end )
    {
        if( i - L.qwinLo >= 16u )
            seed_qwin_load( L, P, i );
        const u32 d = i - L.qwinLo;
        // selects of VALUES: without the empty asm the compiler folds them into one load through a selected address, and that address
        // keeps the whole lane state of k_seed_tasks* in scratch memory (90 scratch stores in its loop, 0.76 TB of writes per step)
        u32 w0 = L.qw0, w1 = L.qw1, w2 = L.qw2, w3 = L.qw3;
        asm( "" : "+v"( w0 ), "+v"( w1 ), "+v"( w2 ), "+v"( w3 ) );
        const u32 lo = d & 8u ? w2 : w0, hi = d & 8u ? w3 : w1;
        return ( ( d & 4u ? hi : lo ) >> ( 8u * ( d & 3u ) ) ) & 0xffu;
    }
#endif
    return L.q[ i ];
}

MA_HD u32 comp_base( u32 c ) // NucSeq::nucleotideComplement (nucSeq.h:524-532)
{
    return c < 4 ? 3 - c : 5;
}

// Entries of the SMEM pending lists (binarySeeding.h:296-433).  The lists live in HBM, one pair per resident lane, and
// their traffic is as large as that of the occ blocks (Illumina preset: 99 GB of reads and 94 GB of writes per 1 M reads
// beside 105 GB of occ blocks).  A 40-byte record straddles 64-byte lines; packed into 16 bytes -- three 35-bit interval
// fields and ONE 22-bit query field -- an entry is one aligned dwordx4 and four of them share a line.
// The start of the match is not stored: all entries of one list share it.  The forward phase pushes (center, ...)
// (binarySeeding.h:296-337), a backward position i pushes (i, ...) (:380-413), so the entries a backward position i reads were
// pushed at i + 1 (or by the forward phase with center = i + 1) and the caller names the start when it unpacks (`qs`);
// the 40-byte form keeps its own copy, and the host emulation checks that both agree.  (Round 4 stored two 11-bit query
// fields, which kept reads of 2048 bases and more -- the Nanopore preset's -- on the 40-byte records.)
#define MA_SMEM_QZ_BITS 22
MA_HD void smem_pack( u32 qz, i64 a, i64 b, i64 c, u64& w0, u64& w1 )
{
    w0 = (u64)a | ( ( (u64)c & 0x1fffffffull ) << 35 );
    w1 = (u64)b | ( ( (u64)c >> 29 ) << 35 ) | ( (u64)qz << 41 );
}
MA_HD ma_segment smem_unpack( u64 w0, u64 w1, u32 qs )
{
    ma_segment s;
    s.sa_start = (i64)( w0 & 0x7ffffffffull );
    s.sa_start_rc = (i64)( w1 & 0x7ffffffffull );
    s.sa_size = (i64)( ( w0 >> 35 ) | ( ( ( w1 >> 35 ) & 0x3full ) << 29 ) );
    s.q_start = (i64)qs;
    s.q_size = (i64)( ( w1 >> 41 ) & ( ( 1ull << MA_SMEM_QZ_BITS ) - 1 ) );
    return s;
}
MA_HD void smem_put( const SeedParams& P, ma_segment* list, u32 idx, u32 qs, u32 qz, i64 a, i64 b, i64 c )
{
    if( P.smem_compact )
    {
        u64* w = (u64*)list + 2 * (u64)idx;
        u64 w0, w1;
        smem_pack( qz, a, b, c, w0, w1 );
#if defined( __HIP_DEVICE_COMPILE__ )
        *(ulonglong2*)w = make_ulonglong2( w0, w1 );
#else
        w[ 0 ] = w0, w[ 1 ] = w1;
#endif
    }
    else
    {
        ma_segment s;
        s.q_start = qs, s.q_size = qz, s.sa_start = a, s.sa_start_rc = b, s.sa_size = c;
        list[ idx ] = s;
    }
}
// qs: the start all entries of this list share (see above)
MA_HD ma_segment smem_get( const SeedParams& P, const ma_segment* list, u32 idx, u32 qs )
{
    if( P.smem_compact )
    {
        const u64* w = (const u64*)list + 2 * (u64)idx;
#if defined( __HIP_DEVICE_COMPILE__ )
        const ulonglong2 v = *(const ulonglong2*)w;
        const u64 w0 = v.x, w1 = v.y;
#else
        const u64 w0 = w[ 0 ], w1 = w[ 1 ];
#endif
        return smem_unpack( w0, w1, qs );
    }
#if !defined( __HIP_DEVICE_COMPILE__ )
    if( (u64)list[ idx ].q_start != (u64)qs )
        abort( ); // host emulation only: the shared start the compact form relies on
#endif
    return list[ idx ];
}

// request entry idx of a list of 16-byte entries into the lane's prefetch registers
MA_HD void smem_prefetch( SeedLane& L, const ma_segment* list, u32 idx );
// the same three by list NUMBER (0 = smem_a, 1 = smem_b): the first S.lds_n entries of a list live in LDS (SeedScratch::lds)
MA_HD void smem_put_w( const SeedParams& P, const SeedScratch& S, u32 w, u32 idx, u32 qs, u32 qz, i64 a, i64 b, i64 c )
{
    if( P.smem_compact && idx < S.lds_n )
    {
        u64 w0, w1;
        smem_pack( qz, a, b, c, w0, w1 );
        u64* p = S.lds + 2 * (u64)( ( w * S.lds_n + idx ) * S.lds_stride );
#if defined( __HIP_DEVICE_COMPILE__ )
        *(ulonglong2*)p = make_ulonglong2( w0, w1 );
#else
        p[ 0 ] = w0, p[ 1 ] = w1;
#endif
        return;
    }
    smem_put( P, w ? S.smem_b : S.smem_a, idx, qs, qz, a, b, c );
}
MA_HD ma_segment smem_get_w( const SeedParams& P, const SeedScratch& S, u32 w, u32 idx, u32 qs )
{
    if( P.smem_compact && idx < S.lds_n )
    {
        const u64* p = S.lds + 2 * (u64)( ( w * S.lds_n + idx ) * S.lds_stride );
#if defined( __HIP_DEVICE_COMPILE__ )
        const ulonglong2 v = *(const ulonglong2*)p;
        return smem_unpack( v.x, v.y, qs );
#else
        return smem_unpack( p[ 0 ], p[ 1 ], qs );
#endif
    }
    return smem_get( P, w ? S.smem_b : S.smem_a, idx, qs );
}
MA_HD void seed_emit( SeedLane& L, const SeedScratch& S, u32 start, u32 size, i64 a, i64 b, i64 c )
{
    if( L.nseg < S.seg_cap )
    {
        ma_segment s;
        s.q_start = start;
        s.q_size = size;
        s.sa_start = a;
        s.sa_start_rc = b;
        s.sa_size = c;
        S.stage[ L.nseg ] = s;
        if( S.drop_div )
            L.drop_sum += (u64)size / (u64)S.drop_div;
    }
    else
        L.err |= MA_ERR_SEG_OVERFLOW;
    L.nseg++;
}

MA_HD void smem_prefetch( SeedLane& L, const ma_segment* list, u32 idx )
{
    const u64* w = (const u64*)list + 2 * (u64)idx;
#if defined( __HIP_DEVICE_COMPILE__ )
    const ulonglong2 v = *(const ulonglong2*)w;
    L.nxt0 = v.x, L.nxt1 = v.y;
#else
    L.nxt0 = w[ 0 ], L.nxt1 = w[ 1 ];
#endif
    L.nxtOk = 1;
}
MA_HD void smem_prefetch_w( SeedLane& L, const SeedScratch& S, u32 w, u32 idx ) // (16-byte entries only, like smem_prefetch)
{
    if( idx < S.lds_n )
    {
        const u64* p = S.lds + 2 * (u64)( ( w * S.lds_n + idx ) * S.lds_stride );
#if defined( __HIP_DEVICE_COMPILE__ )
        const ulonglong2 v = *(const ulonglong2*)p;
        L.nxt0 = v.x, L.nxt1 = v.y;
#else
        L.nxt0 = p[ 0 ], L.nxt1 = p[ 1 ];
#endif
        L.nxtOk = 1;
        return;
    }
    smem_prefetch( L, w ? S.smem_b : S.smem_a, idx );
}
MA_HD void seed_begin_read( SeedLane& L, const uint8_t* q, u32 qlen )
{
    L.nxtOk = 0;
    L.q = q;
    L.qlen = qlen;
    L.qwinLo = 0x80000000u; // empty
    L.sp = 0;
    L.aS = 0;
    L.aN = qlen;
    L.nseg = 0;
    L.drop_sum = 0;
    L.err = 0;
    L.steps = L.blocks = 0;
    L.task_mode = 0;
    L.phase = qlen == 0 ? PH_DONE : PH_NEW_CENTER;
}
// task mode: the lane extends the centre of [aS, aS + aN) of the read and stops
MA_HD void seed_begin_area( SeedLane& L, const uint8_t* q, u32 qlen, u32 aS, u32 aN )
{
    seed_begin_read( L, q, qlen );
    L.task_mode = 1;
    L.aS = aS;
    L.aN = aN;
    L.childN[ 0 ] = L.childN[ 1 ] = 0;
    L.childS[ 0 ] = L.childS[ 1 ] = 0;
    L.phase = aN == 0 ? PH_DONE : PH_NEW_CENTER;
}

// After an extension around `center` covered [cS, cS+cN] (reference convention), split the area
// (binarySeeding.cpp:58-82): recurse left first, continue right afterwards.
MA_HD void seed_after_center( SeedLane& L, const SeedScratch& S, u32 cS, u32 cN )
{
    const u32 cE = cS + cN, aE = L.aS + L.aN;
    const bool hasLeft = cS != 0 && L.aS + 1 < cS;
    const bool hasRight = aE > cE + 1;
    if( L.task_mode )
    {
        L.childS[ 0 ] = L.aS, L.childN[ 0 ] = hasLeft ? cS - L.aS : 0;
        L.childS[ 1 ] = cE, L.childN[ 1 ] = hasRight ? aE - cE : 0;
        L.phase = PH_DONE;
        return;
    }
    if( hasLeft )
    {
        if( hasRight )
        {
            if( L.sp < MA_SEED_STACK )
            {
                S.stack[ 2 * L.sp ] = cE;
                S.stack[ 2 * L.sp + 1 ] = aE - cE;
                L.sp++;
            }
            else
                L.err |= MA_ERR_STACK_OVERFLOW;
        }
        L.aN = cS - L.aS; // [aS, cS)
        L.phase = PH_NEW_CENTER;
    }
    else if( hasRight )
    {
        L.aS = cE;
        L.aN = aE - cE;
        L.phase = PH_NEW_CENTER;
    }
    else if( L.sp > 0 )
    {
        L.sp--;
        L.aS = S.stack[ 2 * L.sp ];
        L.aN = S.stack[ 2 * L.sp + 1 ];
        L.phase = PH_NEW_CENTER;
    }
    else
        L.phase = PH_DONE;
}

// The first K-1 extension steps of a run that starts at the centre with a single-base interval come from the K-mer table of
// the index: the right run (PH_P1_RIGHT) of the bases q[centre .. centre+K-1] complemented, the left run (PH_P2_LEFT) of
// q[centre], q[centre-1], ..  An entry is valid for a run when its interval is still larger than min_amb: interval sizes only
// shrink along a run, so none of the skipped steps would have met the stop rule (binarySeeding.h:109-112).
MA_HD bool kmer_decode( const SeedParams& P, u64 x, u64 y, i64 ik[ 3 ] )
{
    const i64 size = (i64)( ( x >> 35 ) | ( ( ( y >> 35 ) & 0x3full ) << 29 ) );
    if( size <= (i64)P.min_amb )
        return false;
    ik[ 0 ] = (i64)( x & 0x7ffffffffull );
    ik[ 1 ] = (i64)( y & 0x7ffffffffull );
    ik[ 2 ] = size;
    return true;
}
// 2-bit codes of up to 16 bases held one per byte (x: bases 0..7, y: 8..15) -> base j at bits 2j, 2j+1
MA_HD u32 kmer_gather( u64 x, u64 y )
{
    auto g = []( u64 v ) -> u32 {
        v = ( v | ( v >> 6 ) ) & 0x000f000f000f000full;
        v = ( v | ( v >> 12 ) ) & 0x000000ff000000ffull;
        return (u32)( ( v | ( v >> 24 ) ) & 0xffffull );
    };
    return g( x ) | g( y ) << 16;
}
MA_HD u32 kmer_key_right( u32 le, u32 K ) // first base of the span most significant, complemented
{
    u32 r = 0;
#if defined( __HIP_DEVICE_COMPILE__ )
    r = __brev( le ) >> ( 32u - 2u * K );
#else
    for( u32 j = 0; j < 32; j++ )
        r |= ( ( le >> j ) & 1u ) << ( 31u - j );
    r >>= 32u - 2u * K;
#endif
    r = ( ( r & 0xaaaaaaaau ) >> 1 ) | ( ( r & 0x55555555u ) << 1 );
    return r ^ (u32)( ( 1ull << ( 2u * K ) ) - 1ull );
}

// Start of a centre: its base (returned), and -- when the index has a K-mer table -- the table entries of BOTH runs that
// start at the centre: eR (returned through eRx / eRy, 0 = none) and the lane's eL.  Reads in HBM (WIN): ONE load of the 48
// bytes around the centre serves the base, both keys and the bases the later transitions of this centre start with (L.cb);
// step by step the same costs a window load for the centre, K byte loads per key and a window load per transition, each a
// memory round trip of its own that the whole wavefront waits for.
template <bool WIN, bool JUMP, bool SM = true, bool MS = true> MA_HD u32 seed_center( SeedLane& L, const SeedParams& P, const IndexView& X, u64& eRx, u64& eRy )
{
    const u32 K = JUMP && MS && ( !SM || P.technique == 0 ) ? X.kmer_k : 0, c = L.center;
    eRx = eRy = 0;
    if( MS )
    {
        L.eL0 = L.eL1 = 0;
        L.cb = 0;
    }
    bool vR = K != 0 && c + K <= L.qlen, vL = K != 0 && c + 1 >= K;
    u32 keyR = 0, keyL = 0, qc;
#if defined( __HIP_DEVICE_COMPILE__ )
    const uintptr_t a0 = (uintptr_t)( L.q + c ) - K; // q[c-K] (possibly in front of the read)
    const uintptr_t base = a0 & ~(uintptr_t)15;
    if( WIN && K != 0 && P.window_end && base >= (uintptr_t)P.window_begin && base + 48 <= (uintptr_t)P.window_end )
    {
        const uint4 b0 = ( (const uint4*)base )[ 0 ], b1 = ( (const uint4*)base )[ 1 ], b2 = ( (const uint4*)base )[ 2 ];
        const u64 w0 = b0.x | (u64)b0.y << 32, w1 = b0.z | (u64)b0.w << 32, w2 = b1.x | (u64)b1.y << 32, w3 = b1.z | (u64)b1.w << 32,
                  w4 = b2.x | (u64)b2.y << 32, w5 = b2.z | (u64)b2.w << 32;
        const u32 d0 = (u32)( a0 - base ), sh = 8u * ( d0 & 7u );
        const bool t = ( d0 & 8u ) != 0;
        auto fun = [ & ]( u64 lo, u64 hi ) -> u64 { return sh ? ( lo >> sh ) | ( hi << ( 64u - sh ) ) : lo; };
        // v[j]: bytes 8j .. 8j+7 of the view that starts at q[c-K]
        const u64 v0 = fun( t ? w1 : w0, t ? w2 : w1 ), v1 = fun( t ? w2 : w1, t ? w3 : w2 ), v2 = fun( t ? w3 : w2, t ? w4 : w3 ),
                  v3 = fun( t ? w4 : w3, t ? w5 : w4 );
        auto at = [ & ]( u32 j ) -> u32 { // j is wave-uniform
            const u64 v = j < 8 ? v0 : j < 16 ? v1 : j < 24 ? v2 : v3;
            return (u32)( v >> ( 8u * ( j & 7u ) ) ) & 0xffu;
        };
        auto span = [ & ]( u32 j, u64& x, u64& y ) { // K bytes from byte j on
            const u32 s8 = 8u * ( j & 7u );
            const u64 p = j < 8 ? v0 : j < 16 ? v1 : v2, q = j < 8 ? v1 : j < 16 ? v2 : v3, r = j < 8 ? v2 : j < 16 ? v3 : 0ull;
            x = s8 ? ( p >> s8 ) | ( q << ( 64u - s8 ) ) : p;
            y = s8 ? ( q >> s8 ) | ( r << ( 64u - s8 ) ) : q;
            if( K < 8 )
                x &= ( 1ull << ( 8u * K ) ) - 1ull;
            y = K > 8 ? y & ( ( 1ull << ( 8u * ( K - 8u ) ) ) - 1ull ) : 0ull; // K <= 14
        };
        qc = at( K );
        const u32 qm1 = at( K - 1 ), qp1 = at( K + 1 ), qmK = at( 0 ), qpK = at( 2 * K );
        if( ( ( qc | qm1 | qp1 | qmK | qpK ) & ~7u ) == 0 )
            L.cb = qmK | qm1 << 3 | qc << 6 | qp1 << 9 | qpK << 12 | 0x8000u;
        u64 x, y;
        span( K, x, y );
        vR = vR && ( ( x | y ) & 0xfcfcfcfcfcfcfcfcull ) == 0;
        keyR = kmer_key_right( kmer_gather( x, y ), K );
        span( 1, x, y );
        vL = vL && ( ( x | y ) & 0xfcfcfcfcfcfcfcfcull ) == 0;
        keyL = kmer_gather( x, y ); // q[c] = last base of the span most significant
    }
    else
#endif
    {
        qc = seed_qbyte<WIN>( L, P, c );
        if( qc >= 4 )
            return qc;
        u32 bad = 0;
        for( u32 j = 0; vR && j < K; j++ )
        {
            const u32 b = L.q[ c + j ];
            bad |= b >> 2;
            keyR = ( keyR << 2 ) | ( ( 3u - b ) & 3u );
        }
        vR = vR && !bad;
        bad = 0;
        for( u32 j = 0; vL && j < K; j++ )
        {
            const u32 b = L.q[ c - j ];
            bad |= b >> 2;
            keyL = ( keyL << 2 ) | ( b & 3u );
        }
        vL = vL && !bad;
    }
    if( qc >= 4 )
        return qc;
#if defined( __HIP_DEVICE_COMPILE__ )
    if( K != 0 )
    {
        // both gathers in flight before either entry is looked at
        ulonglong2 eR = make_ulonglong2( 0, 0 ), eL = make_ulonglong2( 0, 0 );
        if( vR )
            eR = ( (const ulonglong2*)X.kmer_tab )[ keyR ];
        if( vL )
            eL = ( (const ulonglong2*)X.kmer_tab )[ keyL ];
        eRx = eR.x, eRy = eR.y;
        L.eL0 = eL.x, L.eL1 = eL.y;
    }
#endif
    return qc;
}
// base i of the read out of the centre's cached bases when it is one of them
template <bool WIN> MA_HD u32 seed_qbyte_c( SeedLane& L, const SeedParams& P, const IndexView& X, u32 i )
{
    if( WIN && ( L.cb & 0x8000u ) )
    {
        const u32 K = X.kmer_k;
        const i32 d = (i32)( i - L.center );
        if( d == -1 )
            return ( L.cb >> 3 ) & 7u;
        if( d == 1 )
            return ( L.cb >> 9 ) & 7u;
        if( d == 0 )
            return ( L.cb >> 6 ) & 7u;
        if( d == -(i32)K )
            return L.cb & 7u;
        if( d == (i32)K )
            return ( L.cb >> 12 ) & 7u;
    }
    return seed_qbyte<WIN>( L, P, i );
}

MA_HD bool seed_stop( const SeedParams& P, const i64 ok[ 3 ], const i64 ik[ 3 ] ) // binarySeeding.h:109-112
{
    if( ok[ 2 ] <= 0 )
        return true;
    return ok[ 2 ] <= (i64)P.min_amb && ik[ 2 ] <= (i64)P.max_amb;
}

// Can the lane extend right away, i.e. without any phase transition?  (The kernel batches the transitions of a
// wavefront: a lane whose transition is due idles for a few steps until enough lanes wait, so that the divergent
// bookkeeping of seed_prepare is executed once for many lanes instead of on every step for one or two.)
// SM = false: the kernel serves maxSpan only (P.technique == 0): the SMEM states and their lane registers are compiled out;
// MS = false: SMEMs only (P.technique == 1), the same for the maxSpan states
template <bool WIN = false, bool SM = true, bool MS = true> MA_HD bool seed_try( SeedLane& L, const SeedParams& P, u32& c, const SeedScratch* S = nullptr )
{
    if( SM && L.phase == PH_SMEM_BWD )
    {
        // the next entry of the list was requested a step ago (seed_prepare / here): take it, request the one after it
        if( S != nullptr && L.nxtOk && L.jPrev < L.nPrev )
        {
            const ma_segment s = smem_unpack( L.nxt0, L.nxt1, L.i + 1 );
            L.ik[ 0 ] = s.sa_start, L.ik[ 1 ] = s.sa_start_rc, L.ik[ 2 ] = s.sa_size;
            L.curQStart = (u32)s.q_start, L.curQSize = (u32)s.q_size;
            if( L.jPrev + 1 < L.nPrev )
                smem_prefetch_w( L, *S, L.flip, L.jPrev + 1 );
            else
                L.nxtOk = 0;
            c = seed_qbyte<WIN>( L, P, L.i );
            return true;
        }
        // End of a backward position (the bookkeeping of seed_prepare's PH_SMEM_BWD) without leaving the fast path: the next
        // position starts with the entry that was pushed FIRST onto the list just written, which is still in registers -- no
        // list read, no batching with other lanes' transitions; the second entry is requested for the step after this one.
        // (16-byte entries only.)  The entry was pushed at position L.i, which is the start it carries.
        if( S != nullptr && P.smem_compact && L.jPrev == L.nPrev && L.nCurr >= 1 && L.i != 0 )
        {
            L.flip ^= 1;
            L.nPrev = L.nCurr;
            L.nCurr = 0;
            L.jPrev = 0;
            L.bHaveOne = 0;
            L.retS = L.i;
            const ma_segment s = smem_unpack( L.first0, L.first1, L.i );
            L.ik[ 0 ] = s.sa_start, L.ik[ 1 ] = s.sa_start_rc, L.ik[ 2 ] = s.sa_size;
            L.curQStart = (u32)s.q_start, L.curQSize = (u32)s.q_size;
            if( L.nPrev > 1 )
                smem_prefetch_w( L, *S, L.flip, 1 );
            else
                L.nxtOk = 0;
            L.i--;
            c = seed_qbyte<WIN>( L, P, L.i );
            return true;
        }
        return false;
    }
    const bool right = ( MS && ( L.phase == PH_P1_RIGHT || L.phase == PH_P2_RIGHT ) ) || ( SM && L.phase == PH_SMEM_FWD );
    const bool left = MS && ( L.phase == PH_P1_LEFT || L.phase == PH_P2_LEFT );
    const bool ok = right ? L.i < L.qlen : ( left && L.i != 0xffffffffu );
    if( ok )
    {
        const u32 b = seed_qbyte<WIN>( L, P, L.i );
        c = right ? comp_base( b ) : b;
    }
    return ok;
}

// The base of THIS step has been taken out of the 16-byte window (seed_try / seed_prepare returned true); when the run's next
// base lies outside of it, the next block is requested now, ahead of this step's occ blocks: it arrives while the lane waits
// for those anyway.  (Loaded on demand at the top of the next step, some lane of the wavefront needs a block on nearly every
// trip, and every trip became two memory round trips in a row: 7.2 k of 13.7 k cycles per trip of the 10 kb workload.)
template <bool WIN> MA_HD void seed_prefetch( SeedLane& L, const SeedParams& P )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    if( WIN && P.window_end )
    {
        const bool right = L.phase == PH_P1_RIGHT || L.phase == PH_P2_RIGHT || L.phase == PH_SMEM_FWD;
        const bool left = L.phase == PH_P1_LEFT || L.phase == PH_P2_LEFT;
        const u32 nxt = right ? L.i + 1 : L.i - 1; // L.i == 0 going left: wraps, and fails the test below
        if( ( right || left ) && nxt < L.qlen && nxt - L.qwinLo >= 16u )
            seed_qwin_load( L, P, nxt );
    }
#endif
}

// ---- transitions (no index access) ----------------------------------------------------------
// Runs cheap bookkeeping until the lane either needs an extension (returns true and sets c) or is done.
// JUMP: use the K-mer table of the index (only the kernel whose register budget has room for it: inlined into the
// read-per-lane kernel it costs a wave of occupancy, 117 -> 138 VGPRs, which eats the gain)
template <bool WIN = false, bool JUMP = false, bool SM = true, bool MS = true> MA_HD bool seed_prepare( SeedLane& L, const SeedParams& P, const SeedScratch& S, const IndexView& X, u32& c )
{
    while( true )
    {
        switch( L.phase )
        {
            case PH_DONE:
                return false;
            case PH_NEW_CENTER:
            {
                L.center = L.aS + L.aN / 2;
                u64 eRx, eRy;
                const u32 qc = seed_center<WIN, JUMP, SM, MS>( L, P, X, eRx, eRy );
                if( qc >= 4 )
                { // N covers one position (binarySeeding.h:70-72 / 275-277)
                    seed_after_center( L, S, L.center, 1 );
                    break;
                }
                init_interval( X, 3 - qc, L.ik );
                if( MS && ( !SM || P.technique == 0 ) )
                {
                    if( L.ik[ 2 ] == 0 )
                    {
                        seed_after_center( L, S, L.center, 1 );
                        break;
                    }
                    L.end = L.center;
                    L.i = L.center + 1;
                    if( JUMP && kmer_decode( P, eRx, eRy, L.ik ) )
                    {
                        L.end = L.center + X.kmer_k - 1;
                        L.i = L.center + X.kmer_k;
                    }
                    L.phase = PH_P1_RIGHT;
                }
                else if( SM )
                {
                    L.nCurr = 0;
                    L.retS = L.retE = L.center;
                    L.i = L.center + 1;
                    L.flip = 0;
                    L.phase = PH_SMEM_FWD;
                }
                else
                    L.phase = PH_DONE;
                break;
            }
            case PH_P1_RIGHT:
                if( !MS )
                    return false;
                if( L.i < L.qlen )
                {
                    c = comp_base( seed_qbyte_c<WIN>( L, P, X, L.i ) );
                    return true;
                }
                // end of query: switch direction (binarySeeding.h:118-120)
                mswap( L.ik[ 0 ], L.ik[ 1 ] );
                L.start = L.center;
                if( L.center > 0 )
                {
                    L.i = L.center - 1;
                    L.phase = PH_P1_LEFT;
                    break;
                }
                L.phase = PH_P1_LEFT;
                L.i = 0xffffffffu; // sentinel: left loop skipped
                break;
            case PH_P1_LEFT:
                if( !MS )
                    return false;
                if( L.i != 0xffffffffu )
                {
                    c = seed_qbyte_c<WIN>( L, P, X, L.i );
                    return true;
                }
                // record first segment and start the second pass (binarySeeding.h:152-163)
                seed_emit( L, S, L.start, L.end - L.start, L.ik[ 0 ], L.ik[ 1 ], L.ik[ 2 ] );
                L.s1_start = L.start;
                L.s1_end = L.end;
                init_interval( X, seed_qbyte_c<WIN>( L, P, X, L.center ), L.ik );
                L.start = L.center;
                L.phase = PH_P2_LEFT;
                L.i = L.center > 0 ? L.center - 1 : 0xffffffffu;
                if( JUMP && kmer_decode( P, L.eL0, L.eL1, L.ik ) )
                {
                    L.start = L.center - ( X.kmer_k - 1 );
                    L.i = L.start > 0 ? L.start - 1 : 0xffffffffu;
                }
                break;
            case PH_P2_LEFT:
                if( !MS )
                    return false;
                if( L.i != 0xffffffffu )
                {
                    c = seed_qbyte_c<WIN>( L, P, X, L.i );
                    return true;
                }
                mswap( L.ik[ 0 ], L.ik[ 1 ] );
                L.end = L.center;
                L.i = L.center + 1;
                L.phase = PH_P2_RIGHT;
                break;
            case PH_P2_RIGHT:
                if( !MS )
                    return false;
                if( L.i < L.qlen )
                {
                    c = comp_base( seed_qbyte_c<WIN>( L, P, X, L.i ) );
                    return true;
                }
                {
                    // finish the center (binarySeeding.h:226-251)
                    if( L.s1_start == L.start && L.s1_end == L.end )
                        seed_after_center( L, S, L.s1_start, L.s1_end - L.s1_start );
                    else
                    {
                        seed_emit( L, S, L.start, L.end - L.start, L.ik[ 1 ], L.ik[ 0 ], L.ik[ 2 ] );
                        const u32 s = L.start < L.s1_start ? L.start : L.s1_start;
                        const u32 e = L.end > L.s1_end ? L.end : L.s1_end;
                        seed_after_center( L, S, s, e - s );
                    }
                }
                break;
            case PH_SMEM_FWD:
                if( !SM )
                    return false;
                if( L.i < L.qlen )
                {
                    c = comp_base( seed_qbyte<WIN>( L, P, L.i ) );
                    return true;
                }
                // forward phase over: reverse the list (binarySeeding.h:343) and go backwards
                {
                    for( u32 a = 0, b = L.nCurr; a + 1 < b; a++, b-- )
                    {
                        const ma_segment t = smem_get_w( P, S, 0, a, L.center ), u = smem_get_w( P, S, 0, b - 1, L.center );
                        smem_put_w( P, S, 0, a, (u32)u.q_start, (u32)u.q_size, u.sa_start, u.sa_start_rc, u.sa_size );
                        smem_put_w( P, S, 0, b - 1, (u32)t.q_start, (u32)t.q_size, t.sa_start, t.sa_start_rc, t.sa_size );
                    }
                    L.nPrev = L.nCurr;
                    L.nCurr = 0;
                    L.flip = 0; // prev = smem_a, curr = smem_b
                    L.jPrev = 0;
                    L.nxtOk = 0;
                    L.bHaveOne = 0;
                    if( L.center != 0 && L.nPrev > 0 )
                    {
                        L.i = L.center - 1;
                        L.phase = PH_SMEM_BWD;
                    }
                    else
                    {
                        // cannot extend backwards at all (binarySeeding.h:354, 437-448)
                        if( L.nPrev > 0 )
                        {
                            const ma_segment f = smem_get_w( P, S, 0, 0, L.center );
                            seed_emit( L, S, (u32)f.q_start, (u32)f.q_size, f.sa_start, f.sa_start_rc, f.sa_size );
                        }
                        seed_after_center( L, S, L.retS, L.retE - L.retS );
                    }
                }
                break;
            case PH_SMEM_BWD:
            {
                if( !SM )
                    return false;
                if( L.jPrev < L.nPrev )
                {
                    const ma_segment s = L.nxtOk ? smem_unpack( L.nxt0, L.nxt1, L.i + 1 ) : smem_get_w( P, S, L.flip, L.jPrev, L.i + 1 );
                    if( P.smem_compact && L.jPrev + 1 < L.nPrev )
                        smem_prefetch_w( L, S, L.flip, L.jPrev + 1 );
                    else
                        L.nxtOk = 0;
                    L.ik[ 0 ] = s.sa_start;
                    L.ik[ 1 ] = s.sa_start_rc;
                    L.ik[ 2 ] = s.sa_size;
                    L.curQStart = (u32)s.q_start;
                    L.curQSize = (u32)s.q_size;
                    c = seed_qbyte<WIN>( L, P, L.i );
                    return true;
                }
                // end of one backward position: swap lists (binarySeeding.h:416-433)
                L.nxtOk = 0;
                L.flip ^= 1;
                L.nPrev = L.nCurr;
                L.nCurr = 0;
                L.jPrev = 0;
                L.bHaveOne = 0;
                bool fin = false;
                if( L.nPrev == 0 )
                    fin = true;
                else
                {
                    L.retS = L.i;
                    if( L.i == 0 )
                        fin = true;
                    else
                        L.i--;
                }
                if( fin )
                {
                    if( L.nPrev > 0 )
                    {
                        const ma_segment f = smem_get_w( P, S, L.flip, 0, L.i );
                        seed_emit( L, S, (u32)f.q_start, (u32)f.q_size, f.sa_start, f.sa_start_rc, f.sa_size );
                    }
                    seed_after_center( L, S, L.retS, L.retE - L.retS );
                }
                break;
            }
            default:
                return false;
        }
    }
}

// ---- apply the result of the extension requested by seed_prepare --------------------------------
template <bool SM = true, bool MS = true> MA_HD void seed_apply( SeedLane& L, const SeedParams& P, const SeedScratch& S, const i64 ok[ 3 ] )
{
    switch( L.phase )
    {
        case PH_P1_RIGHT:
        case PH_P2_RIGHT:
            if( !MS )
                break;
            if( seed_stop( P, ok, L.ik ) )
                L.i = L.qlen; // leave the loop; seed_prepare performs the transition
            else
            {
                L.end = L.i;
                L.ik[ 0 ] = ok[ 0 ], L.ik[ 1 ] = ok[ 1 ], L.ik[ 2 ] = ok[ 2 ];
                L.i++;
            }
            break;
        case PH_P1_LEFT:
        case PH_P2_LEFT:
            if( !MS )
                break;
            if( seed_stop( P, ok, L.ik ) )
                L.i = 0xffffffffu;
            else
            {
                L.start = L.i;
                L.ik[ 0 ] = ok[ 0 ], L.ik[ 1 ] = ok[ 1 ], L.ik[ 2 ] = ok[ 2 ];
                L.i = L.i == 0 ? 0xffffffffu : L.i - 1;
            }
            break;
        case PH_SMEM_FWD:
        { // binarySeeding.h:296-337
            if( !SM )
                break;
            auto push = [ & ]( u32 st, u32 sz, i64 a, i64 b, i64 c ) {
                if( L.nCurr < S.smem_cap )
                    smem_put_w( P, S, 0, L.nCurr, st, sz, a, b, c );
                else
                    L.err |= MA_ERR_SMEM_OVERFLOW;
                L.nCurr++;
            };
            if( ok[ 2 ] != L.ik[ 2 ] )
                push( L.center, L.i - L.center - 1, L.ik[ 1 ], L.ik[ 0 ], L.ik[ 2 ] );
            if( L.i == L.qlen - 1 && ok[ 2 ] != 0 )
                push( L.center, L.i - L.center, ok[ 1 ], ok[ 0 ], ok[ 2 ] );
            if( ok[ 2 ] == 0 || ( ok[ 2 ] <= (i64)P.min_amb && L.ik[ 2 ] <= (i64)P.max_amb ) )
                L.i = L.qlen; // break
            else
            {
                L.ik[ 0 ] = ok[ 0 ], L.ik[ 1 ] = ok[ 1 ], L.ik[ 2 ] = ok[ 2 ];
                L.retE = L.i;
                L.i++;
            }
            break;
        }
        case PH_SMEM_BWD:
        { // binarySeeding.h:380-413
            if( !SM )
                break;
            ma_segment s; // = prev[ L.jPrev ], kept in the lane state by seed_prepare (saves a memory round trip per step)
            s.q_start = L.curQStart, s.q_size = L.curQSize, s.sa_start = L.ik[ 0 ], s.sa_start_rc = L.ik[ 1 ], s.sa_size = L.ik[ 2 ];
            if( ok[ 2 ] <= (i64)P.min_amb && !L.bHaveOne )
            {
                seed_emit( L, S, (u32)s.q_start, (u32)s.q_size, s.sa_start, s.sa_start_rc, s.sa_size );
                L.bHaveOne = 1;
            }
            else if( ok[ 2 ] > (i64)P.min_amb || ( ok[ 2 ] > 0 && (u64)s.q_size >= (u64)P.max_amb ) )
            {
                // Twins.  The list is ordered by decreasing match length; after a few backward steps most of its entries
                // have shrunk to the SAME interval (for a unique read: all ~13 of them, the one locus) and differ only in
                // length.  extend_backward reads start and size only, so twins fail or survive together on every later
                // position, stay neighbours, and the later one is never emitted: when they fail the earlier one has set
                // bHaveOne (or found it set), and with uiMinAmbiguity == 0 a failed entry has size 0 and is not pushed
                // either (binarySeeding.h:380-413); at the start of the query only the front of the list is emitted
                // (:437-448).  Dropping the later twin here changes no emitted segment and no list's emptiness -- and
                // takes the backward phase of a 150 bp read from ~13 extensions per position to ~1 (a run-time switch:
                // with uiMinAmbiguity > 0 a failed twin of sufficient length IS pushed, so the lists stay as they are).
                const bool twin = P.smem_merge && L.nCurr > 0 && L.lastK == ok[ 0 ] && L.lastS == ok[ 2 ];
                if( !twin )
                {
                    if( L.nCurr < S.smem_cap )
                        smem_put_w( P, S, L.flip ^ 1u, L.nCurr, L.i, (u32)s.q_size + 1, ok[ 0 ], ok[ 1 ], ok[ 2 ] );
                    else
                        L.err |= MA_ERR_SMEM_OVERFLOW;
                    if( L.nCurr == 0 )
                        smem_pack( (u32)s.q_size + 1, ok[ 0 ], ok[ 1 ], ok[ 2 ], L.first0, L.first1 );
                    L.nCurr++;
                    L.lastK = ok[ 0 ], L.lastS = ok[ 2 ];
                }
            }
            L.jPrev++;
            break;
        }
        default:
            break;
    }
}

// BinarySeeding::execute's post filter (binarySeeding.cpp:172-175, numSeedsLarger segment.h:278-289):
// returns the number of segments to keep (0 = drop all)
MA_HD u32 seed_finish( const SeedLane& L, const SeedParams& P, const SeedScratch& S, const IndexView& X )
{
    u32 n = L.nseg < S.seg_cap ? L.nseg : S.seg_cap;
    if( !P.disable_heuristics && P.min_seed_size_drop != 0 )
    {
        const u64 sum = L.drop_sum; // == sum of stage[k].q_size / min_seed_size_drop over the n kept segments
        if( (double)sum < P.rel_min_seed_size_amount * (double)L.qlen && P.genome_size_disable < X.n )
            n = 0;
    }
    return n;
}

// memExtension (binarySeeding.h:460-537) for ONE start position i of a read: extend rightwards from q[i]; whenever rows
// drop out of the interval (and the match is longer than the minimal seed length and not too ambiguous), the dropped rows
// are maximal to the right, and those of them that cannot be extended to the left by q[i-1] are reported -- per row when
// only some of them can.  SINK::emit( q_start, q_size, sa_start, sa_size ) receives the segments in the reference's order;
// the intervals of dropped rows carry -1 as start of the reverse-complement interval (fMIndex.h:123-150), which
// extend_backward's start and size do not depend on.  Every position is independent: one lane per (read, position).
template <typename SINK>
MA_HD void mems_from( const IndexView& X, const SeedParams& P, const uint8_t* q, u32 qlen, u32 i, SINK& sink, u64& steps, u64& blocks )
{
    if( q[ i ] >= 4 )
        return;
    const i64 minAmb = (i64)P.min_amb, maxAmb = (i64)P.max_amb;
    i64 ik[ 3 ];
    init_interval( X, 3u - q[ i ], ik );
    u32 nb;
    for( u32 j = i + 1; j <= qlen && ik[ 2 ] > minAmb; j++ )
    {
        i64 ok[ 3 ] = { 0, -1, 0 };
        if( j < qlen && q[ j ] < 4 )
        {
            extend_backward( X, ik, 3u - q[ j ], ok, nb );
            steps++, blocks += nb;
        }
        if( j - i - 1 > P.min_seed_len && ok[ 2 ] < ik[ 2 ] && ik[ 2 ] < maxAmb )
        {
            // ik.revComp( ) minus ok.revComp( ): at most two runs of rows (do_for_difference)
            const i64 aS = ik[ 1 ], aE = ik[ 1 ] + ik[ 2 ], bS = ok[ 1 ], bE = ok[ 1 ] + ok[ 2 ];
            const i64 uiY = mmin( bS, aE ), uiX = mmax( bE, aS );
            for( int part = 0; part < 2; part++ )
            {
                const i64 dS = part == 0 ? aS : uiX, dN = part == 0 ? uiY - aS : aE - uiX;
                if( dN <= 0 )
                    continue; // start( ) < uiY / uiX < end( ) do not hold
                i64 xe[ 3 ] = { 0, -1, 0 };
                if( i > 0 )
                {
                    const i64 xd[ 3 ] = { dS, -1, dN };
                    extend_backward( X, xd, q[ i - 1 ], xe, nb );
                    steps++, blocks += nb;
                }
                if( xe[ 2 ] == 0 )
                    sink.emit( i, j - i - 1, dS, dN );
                else if( xe[ 2 ] < dN )
                {
                    i64 kLast = dS;
                    for( i64 k = dS; k <= dS + dN; k++ )
                    {
                        bool cut = k == dS + dN;
                        if( !cut )
                        {
                            const i64 xr[ 3 ] = { k, -1, 1 };
                            i64 xo[ 3 ];
                            extend_backward( X, xr, q[ i - 1 ], xo, nb );
                            steps++, blocks += nb;
                            cut = xo[ 2 ] != 0;
                        }
                        if( cut )
                        {
                            if( k > kLast )
                                sink.emit( i, j - i - 1, kLast, k - kLast );
                            kLast = k + 1;
                        }
                    }
                }
            }
        }
        ik[ 0 ] = ok[ 0 ], ik[ 1 ] = ok[ 1 ], ik[ 2 ] = ok[ 2 ];
    }
}

// Whole read on one lane (used by the host emulation and as the loop body of the kernel).
MA_HD void seed_read_serial( SeedLane& L, const SeedParams& P, const SeedScratch& S, const IndexView& X )
{
    u32 c;
    while( seed_prepare( L, P, S, X, c ) )
    {
        i64 ok[ 3 ];
        u32 nb;
        extend_backward( X, L.ik, c, ok, nb );
        L.steps++;
        L.blocks += nb;
        seed_apply( L, P, S, ok );
    }
}
} // namespace ma
