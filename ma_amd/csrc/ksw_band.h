// ksw_band.h -- FOUR extension jobs of up to 254 query bases per wavefront: a narrow band that is PROVEN, job by job, to give the
// wide band's answer (round 5).
//
// kswcpp computes every cell of the rectangle a band of 512 leaves (kswcpp_core.h:541-879), and so do ksw_ext.h / ksw_grp.h: an
// extension of q bases is q cells on each of ~2 q diagonals, although a good alignment stays within a few cells of the main
// diagonal.  What the callers read -- max, max_q, max_t and the cigar traced back from there (needlemanWunsch.cpp:239-622) -- is
// decided by those few cells, and that can be CHECKED after the fact.  This kernel computes only the cells with |t - j| <= B
// (B = 24: 25 cells per diagonal, a ring of 32 query rows in the 16 lanes of a quarter wavefront, layout of ksw_grp.h) and hands a
// job back to the exact kernels unless the checks below hold.
//
// The band.  Cell (t, j) lives in half (j & 1) of lane (j & 31) >> 1 of its group while -B <= t - j <= B; a lane whose two rows
// have left the band takes the rows 32 further on.  A cell whose upper neighbour (t-1, j) lies outside the band takes
// v = -(q+e), x = -(q+e), x2 = -(q2+e2) for it, one whose left neighbour (t, j-1) lies outside takes u = -(q+e), y, y2 alike:
// the missing cell is treated as reached by opening a gap from the cell beside it, which is a path of the full matrix.  Hence
//   (V) every value of the band matrix is the score of a real path: band H <= true H, cell by cell.
// Let a = match, f(L) = min(q + L e, q2 + L e2) the cost of a gap of L, G = f(B + 1).  A path that visits a cell outside the band has
// gaps of at least B + 1 in one direction more than in the other (f is concave: several gaps cost at least f of their sum), so
//   (U) a cell (t, j) outside the band, and any cell reached THROUGH one, has true H <= a (min(t, j) + 1) - G, and on diagonal r
//       the cells outside the band have min(t, j) <= (r - B - 1) / 2:   UB(r) = a ((r - B - 1) / 2 + 1) - G.
//   (L) the same UB(r) bounds a BAND cell of diagonal r that is reached through a cell outside the band: a path that visits offset
//       +-(B + 1) and ends at (t, j) with t + j = r has at least B + 1 gap bases in one direction before it returns, so it matches
//       at most (r + 2 - (B + 1)) / 2 = (r - B - 1) / 2 + 1 pairs (rounded down), and its gaps cost at least G.
//   (E) a band cell whose band H exceeds UB of its diagonal is EXACT (true H >= band H > UB: by (L) no optimal path to it leaves the
//       band, and the band matrix holds every path that stays inside), and so is every cell on its optimal paths, with every
//       candidate of the cell update that reaches the maximum: the direction byte of such a cell is the wide matrix's.
// The checks (all on the job's own numbers, after its last diagonal):
//   1. ez.max > a qlen - G: no cell outside the band, on any diagonal, reaches ez.max -- the wide run's maximum is the band's,
//      raised on the same diagonal r*, and the early stop (ksw_reg.h) evaluated on band cells holds for the wide matrix.
//   2. calcMaxScore's position (kswcpp_core.h:156-299) is the largest chunk base over the EIGHT classes (t - st0) mod 8 of the
//      first maximum of each class that beats H[en0]: it depends on every class, not on the maximum alone.  With st0, en0 of
//      the WIDE band: every class has a band cell on r*, and every class's band maximum exceeds UB(r*).  Then by (U), (E) the
//      class maxima, the cells that hold them and their order are the wide matrix's, and H[en0] (outside the band) beats none.
//      (A diagonal with fewer than 8 cells below en0 has no classes: the position is H[en0]'s or that of the first larger cell
//      behind it, and the winner's band H must exceed UB(r*).)
//   3. the cell the back-trace starts from, (max_t, r* - max_t), has band H > UB(r*): by (E) every direction byte on its path is
//      the wide matrix's, and the path stays inside the band (the walk checks it anyway).
//   4. no z-drop in the wide run: its ez.max is at most a (r / 2 + 1) on diagonal r and its diagonal maximum at least the
//      band's, which after a raise to m0 on r0 is at least m0 - (r - r0)(q + e) (ksw_grp.h); the difference stays <= zdrop.
//      The diagonals BEFORE the first raise count too (ez.max = 0, max_t = max_q = -1: the test of ksw_apply_zdrop is armed from
//      diagonal 0; a first-base mismatch z-drops at r = 0 when zdrop < |mismatch|): the first raise on r1 > 0 needs
//      (r1 + 1)(q + e) <= zdrop, the diagonal maximum being at least -(r + 2)(q + e) (round 6).
// A job that fails a check, leaves the regime (r > w) or outgrows its cigar buffer is appended to the list of the extension
// kernel it would have gone to without this one (k_ksw_ext<1> / <2>, which run after it).  tests/test_gpu_round5.py::test_banded_extensions_are_the_wide_bands_or_handed_back compares every
// proved job with the oracle's kswcpp at the full band.
//
// Round 6: the same proof for the LONG extension jobs (G = 1: one job per wavefront, B = 120, a ring of 128 query rows in the 64
// lanes).  A 10 kb read of the reverse strand keeps about half of its seeds in Harmonization and the rest
// of the read -- ~2 500 bases -- is ONE extension against the 1 000 padded reference bases behind the last seed
// (needlemanWunsch.cpp:708-716, 781-782), band 512: 99 % of the DP cells of a 10 kb batch (profiles/r05_dp_job_histogram.txt).  The
// read does continue there: the alignment follows the main diagonal for min(qlen, tlen) bases and loses ~70 points against a perfect
// one (tools/band_long_experiment.py: B = 120 proves 100 % of such jobs, B = 96 98 %, B = 64 78 %).  What differs from the short jobs:
//   * the wide band CUTS the rectangle (w < qlen) and kswcpp's cells at the wide band's edge read stale values.  (U) does not need a
//     clean matrix: H(t, j) <= H(t-1, j-1) + a holds for every cell of the wide band whatever its edge cells read (ksw_reg.h), the
//     chain of a cell at offset d = |t - j| starts at the boundary cell of that offset, so W(t, j) <= a (min(t, j) + 1) - f(d) for
//     the wide run's matrix W.  The cells at offsets <= B + 1 and their neighbours are interior cells of the wide band
//     (ksw_bandl_ok: w >= 2 B + 34, the SSE blocks' garbage lanes included), their recurrences are the clean ones, so restricted to
//     |t - j| <= B the wide matrix IS the band's DP with other inputs at offset +-(B + 1): W >= N cell by cell, and a path that
//     enters through offset +-(B + 1) is bounded by (L).  Hence W(c) <= max(N(c), UB(r)) on the band and (E) holds as before.
//   * check 1 asks less than "a perfect alignment minus G": a cell outside the band with t - j >= B + 1 has j <= tlen - B - 2, one with
//     j - t >= B + 1 has t <= qlen - B - 2, so by (U) none of them exceeds a NB - G with
//     NB = max( min(tlen - B - 1, qlen), min(qlen - B - 1, tlen) ).  For an end extension (qlen >> tlen) that is a tlen - G as
//     before; for the two extensions into a large gap between two seeds (ksw_dual_ext, needlemanWunsch.cpp:239-260: qlen ~ tlen ~
//     thousands) it is a (N - B - 1) - G: an alignment of 4 000 bases may lose 387 points instead of 145 and still be proved.
//   * kswcpp tracks H as int32 when max(qlen, tlen) * 24 leaves int16 (kswcpp.h:101-115): calcMaxScore then has FOUR classes
//     (t - st0) mod 4 instead of eight (band_exact_max_lds, NC).
//   * the early stop: a band cell's diagonal chain stays in the band and starts on a diagonal <= B + 1, so from r >= B + 3 on
//     later band H <= max(B_r, B_{r-1}) over the BAND's cells; the cells outside it stay below a min(qlen, tlen) - G < ez.max.
//   * check 4 uses what the proof gives: up to diagonal r the wide run's ez.max is at most max(band ez.max, UB(r)); a z-drop after
//     the LAST raise changes nothing the callers read (the back-trace starts at the same cell, kswcpp_core.h:796-835).
//   * a job that fails goes to the list of jobs handed back to the exact kernels (k_ksw_pk<S>, second pass of ksw_run_all).
// tests/test_gpu_round6.py::test_long_extensions_on_the_proven_band compares every job with the oracle's kswcpp at the full band.
#pragma once
#include "ksw_grp.h"

#if defined( __HIPCC__ )
namespace ma
{
#define KSW_BAND_B 24 // cells on either side of the main diagonal
#define KSW_BAND_QMAX 254
#define KSW_BANDL_B 120 // long jobs: one per wavefront, 121 cells per diagonal in a ring of 128 query rows
#define KSW_BANDL_NMAX 7900 // min(qlen, tlen) of a long job: bounds the diagonals (2 N + B) and keeps H + a * cells-left in 16 bits
#define KSW_BANDL_ROWS( N ) ( 2 * ( N ) + KSW_BANDL_B + 24 ) // direction rows (128 B) of a wavefront's scratch, N = the largest min(qlen, tlen) of the launch
#define KSW_BAND_LDS ( KSW_GRP_STAGE_ROWS * 128 + KSW_GRP_CIG_WORDS * 4 + 4 * 64 + 4 * 256 + 64 )
// jobs tried, proved, failed check 1 / 2 / 3 / 4, handed back for another reason (regime, cigar buffer), diagonals (ma_debug_band_stats)
static __device__ unsigned long long g_band_stats[ 16 ]; // [8..16): the same for the long jobs (one per wavefront)

// extension jobs this kernel may try (same regime as ksw_ext_slots; the query rows are kept in LDS: <= 254 of them)
MA_HD int ksw_band_ok( const KswScoring& SC, i32 qlen, i32 tlen, i32 w, i32 zdrop, i32 flag, i32 qmin )
{
    if( !( flag & KSW_EZ_EXTZ_ONLY ) || qlen < qmin || qlen > KSW_BAND_QMAX || w > 512 || w < 2 * KSW_BAND_B + 2 )
        return 0;
    if( zdrop < 0 || zdrop > 16000 )
        return 0;
    return ksw_ext_slots( SC, qlen, tlen, w, zdrop, flag ) != 0 ? 1 : 0;
}

// long extension jobs the one-job-per-wavefront variant may try: queries beyond the short variant's, the band well inside the wide one
MA_HD int ksw_bandl_ok( const KswScoring& SC, i32 qlen, i32 tlen, i32 w, i32 zdrop, i32 flag )
{
    if( !( flag & KSW_EZ_EXTZ_ONLY ) || qlen <= KSW_BAND_QMAX || qlen > 32000 || tlen < 1 || tlen > 32000 || w > 16000 )
        return 0;
    if( ( qlen < tlen ? qlen : tlen ) > KSW_BANDL_NMAX || ( qlen < tlen ? qlen : tlen ) < KSW_BANDL_B + 8 || w < 2 * KSW_BANDL_B + 34 )
        return 0;
    if( zdrop < 0 || zdrop > 16000 )
        return 0;
    // the difference vectors stay inside int8 (ksw_ext_slots)
    const i32 a = SC.q + SC.e, b = SC.q2 + SC.e2, mch = SC.match < 0 ? -SC.match : SC.match;
    const i32 mis = SC.mismatch < 0 ? -SC.mismatch : SC.mismatch;
    if( SC.q < 0 || SC.e < 1 || SC.q2 < 0 || SC.e2 < 1 || 2 * ( a > b ? a : b ) + mch + mis > 120 || mch < 1 ||
        mch * ( ( qlen < tlen ? qlen : tlen ) + 8 ) > 16000 ) // (H and H + a * cells-left are packed 16-bit values)
        return 0;
    return 1;
}
// ... and which are worth it: a greedy walk over the first `span` query bases (a mismatch is a substitution, an inserted or a deleted
// base, whichever lets the next three bases match) must get by with maxEdits edits -- 10 kb reads at 1 % errors pass, the noisy
// gaps between the seeds of 50 kb reads at 10 % (which lose > 200 points and can never pass check 1) do not.  A heuristic only: every
// tried job is proved or handed on.
template <typename QF, typename TF> MA_HD bool ksw_bandl_likely( const QF& qf, const TF& tf, i32 qlen, i32 tlen, i32 maxEdits = 5, i32 span = 160 )
{
    i32 i = 0, j = 0, edits = 0;
    const i32 n = qlen < span ? qlen : span;
    while( i < n && j + 4 < tlen && i + 4 < qlen )
    {
        if( (u32)qf( i ) == (u32)tf( j ) )
        {
            i++, j++;
            continue;
        }
        if( ++edits > maxEdits )
            return false;
        const u32 q1 = qf( i + 1 ), q2 = qf( i + 2 ), q3 = qf( i + 3 ), t1 = tf( j + 1 ), t2 = tf( j + 2 ), t3 = tf( j + 3 );
        if( q1 == t1 && q2 == t2 && q3 == t3 )
            i++, j++; // substitution
        else if( q1 == (u32)tf( j ) && q2 == t1 && q3 == t2 )
            i++; // the query has a base more
        else if( (u32)qf( i ) == t1 && q1 == t2 && q2 == t3 )
            j++; // the target has a base more
        else
            i++, j++;
    }
    return true;
}

// Which of the eligible jobs are WORTH the attempt: the checks pass for alignments that lose less than ~44 points against a
// perfect one, and two thirds of the extension jobs of a 150 bp batch are junk (a read end that does not continue where its seed
// lies: profiles/r05_band_stats_150bp.txt, 64 % fail check 1).  A job is tried when the query's first min(qlen, tlen) bases differ
// from the target's in at most KSW_BAND_MAXMIS places on the main diagonal (junk leaves the loop after a handful of bases); the
// others go to the kernels they went to before.  qf / tf: base j of the query / t of the target in DP order.
#define KSW_BAND_MAXMIS 5
template <typename QF, typename TF> MA_HD bool ksw_band_likely( const QF& qf, const TF& tf, i32 qlen, i32 tlen, i32 maxMis = KSW_BAND_MAXMIS )
{
    if( tlen < qlen )
        return false;
    int mis = 0;
    for( i32 i = 0; i < qlen; i++ )
    {
        mis += (u32)qf( i ) != (u32)tf( i ) ? 1 : 0;
        if( mis > maxMis )
            return false;
    }
    return true;
}

// lane i <- lane i - 1 of its group, the group's first lane <- its last (the ring of query rows is circular)
template <int LANES> __device__ __forceinline__ u32 band_ror1( u32 x )
{
    return LANES == 64 ? lanes_ror1( x ) : (u32)dpp_ctrl<0x121>( (i32)x );
}
template <int LANES> __device__ __forceinline__ u32 band_sum_u32( u32 c ) // sum over the lanes of a group, in every lane
{
    c += (u32)dpp_ctrl<0x121>( (i32)c );
    c += (u32)dpp_ctrl<0x122>( (i32)c );
    c += (u32)dpp_ctrl<0x124>( (i32)c );
    c += (u32)dpp_ctrl<0x128>( (i32)c );
    if( LANES == 64 )
    {
        auto a = __builtin_amdgcn_permlane16_swap( c, c, false, false );
        c = a[ 0 ] + a[ 1 ];
        auto b = __builtin_amdgcn_permlane32_swap( c, c, false, false );
        c = b[ 0 ] + b[ 1 ];
    }
    return c;
}

// The same for the one-job-per-wavefront variant (64 lanes, B = KSW_BANDL_B), through LDS and class by class -- it runs once per job.
// NC = 8 classes when kswcpp tracks H as int16, 4 when as int32 (T_SIMD_VEC::SIZE, kswcpp_core.h:183-237).  sc: 144 words of LDS.
__device__ __forceinline__ void band_exact_max_lds( u32 Hs, u32 Jpk, i32 rr, i32 qlen, i32 tlen, i32 w, i32 NC, int lane, i32* sc, i32& mH, i32& mT,
                                                    i32& minClass, i32& hStart )
{
    constexpr i32 B = KSW_BANDL_B;
    const i32 NONE = (i32)0x80000000;
    const i32 st0 = max( max( 0, rr - qlen + 1 ), ( rr - w + 1 ) >> 1 ), en0 = min( min( rr, tlen - 1 ), ( rr + w ) >> 1 );
    const i32 span = en0 - st0, nS = ( span / NC ) * NC;
    // the band's cells of diagonal rr: t = tmin .. tmax (at most B + 1 of them)
    const i32 tmin = max( max( 0, rr - qlen + 1 ), ( rr - B + 1 ) >> 1 ), tmax = min( min( rr, tlen - 1 ), ( rr + B ) >> 1 );
    __syncthreads( );
    sc[ lane ] = NONE;
    sc[ lane + 64 ] = NONE;
    __syncthreads( );
    {
        const i32 jl = (i32)( Jpk & 0xffffu ), jh = (i32)( Jpk >> 16 );
        const i32 tlo = rr - jl, thi = rr - jh;
        if( tlo >= tmin && tlo <= tmax && jl < qlen )
            sc[ tlo - tmin ] = (i32)( Hs << 16 ) >> 16;
        if( thi >= tmin && thi <= tmax && jh < qlen )
            sc[ thi - tmin ] = (i32)Hs >> 16;
    }
    __syncthreads( );
    const i32 hEn0 = en0 >= tmin && en0 <= tmax ? sc[ en0 - tmin ] : NONE;
    // lane c < NC: the first maximum of class c over the chunks [st0, st0 + nS)
    if( lane < NC )
    {
        i32 best = NONE, bestT = en0;
        i32 t = tmin + ( ( ( lane - ( tmin - st0 ) ) % NC ) + NC ) % NC; // first band cell with (t - st0) mod NC == lane
        for( ; t <= tmax && t - st0 < nS; t += NC )
        {
            const i32 h = sc[ t - tmin ];
            if( h != NONE && h > best )
                best = h, bestT = t;
        }
        sc[ 128 + 2 * lane ] = best;
        sc[ 128 + 2 * lane + 1 ] = best != NONE && best > hEn0 ? st0 + ( ( bestT - st0 ) / NC ) * NC : en0;
    }
    __syncthreads( );
    mH = hEn0, mT = en0;
    minClass = 0x7fffffff; // no classes on this diagonal (fewer than NC cells below en0)
    if( nS > 0 )
    {
        i32 vh = hEn0, vt = NONE, mc = 0x7fffffff;
        for( i32 c = 0; c < NC; c++ )
        {
            const i32 b = sc[ 128 + 2 * c ];
            vh = max( vh, b );
            vt = max( vt, sc[ 128 + 2 * c + 1 ] );
            mc = min( mc, b ); // NONE when a class has no band cell
        }
        mH = vh, mT = vt, minClass = mc;
    }
    // the cells after the chunks, in the order the reference visits them
    for( i32 t = max( st0 + nS, tmin ); t < en0 && t <= tmax; t++ )
    {
        const i32 h = sc[ t - tmin ];
        if( h != NONE && h > mH )
            mH = h, mT = t;
    }
    hStart = mT >= tmin && mT <= tmax ? sc[ mT - tmin ] : NONE;
    __syncthreads( );
}

// calcMaxScore over the band cells of diagonal rr (grp_exact_max of ksw_grp.h with the cells' rows per lane in Jpk and the band
// test), plus what the checks need: the smallest of the eight class maxima (NONE when a class has no band cell) and the band H
// of cell (mT, rr - mT).  st0 / en0 are the WIDE band's.  16 lanes per group.
__device__ __forceinline__ void band_exact_max( u32 Hs, u32 Jpk, i32 rr, i32 qlen, i32 tlen, i32 w, int lane, int l, i32& mH, i32& mT, i32& minClass,
                                                i32& hStart )
{
    const i32 NONE = (i32)0x80000000;
    const i32 st0 = max( max( 0, rr - qlen + 1 ), ( rr - w + 1 ) >> 1 ), en0 = min( min( rr, tlen - 1 ), ( rr + w ) >> 1 );
    const i32 span = en0 - st0, nS = ( span / 8 ) * 8;
    // t - st0 of the lane's two cells, and whether they are cells of the band on this diagonal
    const i32 jl = (i32)( Jpk & 0xffffu ), jh = (i32)( Jpk >> 16 );
    const i32 tlo = rr - jl, thi = rr - jh;
    const bool liveLo = tlo >= 0 && tlo < tlen && jl < qlen && tlo - jl >= -KSW_BAND_B && tlo - jl <= KSW_BAND_B;
    const bool liveHi = thi >= 0 && thi < tlen && jh < qlen && thi - jh >= -KSW_BAND_B && thi - jh <= KSW_BAND_B;
    const i32 hLo = (i32)( Hs << 16 ) >> 16, hHi = (i32)Hs >> 16;
    // H[en0]: a band cell only on the first diagonals
    i32 hEn0 = NONE;
    {
        const i32 j0 = rr - en0;
        const i32 v = __builtin_amdgcn_ds_bpermute( ( ( lane - l ) + ( ( j0 & 31 ) >> 1 ) ) << 2, (i32)Hs );
        const i32 jj = __builtin_amdgcn_ds_bpermute( ( ( lane - l ) + ( ( j0 & 31 ) >> 1 ) ) << 2, (i32)Jpk );
        const i32 rowThere = ( j0 & 1 ) ? ( jj >> 16 ) & 0xffff : jj & 0xffff;
        const bool inBand = en0 - j0 >= -KSW_BAND_B && en0 - j0 <= KSW_BAND_B && j0 >= 0 && j0 < qlen;
        if( inBand && rowThere == j0 )
            hEn0 = ( j0 & 1 ) ? ( v >> 16 ) : ( (i32)( (u32)v << 16 ) >> 16 );
    }
    // keys: H << 16 | 0xffff - (t - st0): the first maximum of a class wins
    auto keyOf = [ & ]( bool live, i32 t, i32 h, bool part8 ) -> i32 {
        const i32 d = t - st0;
        if( !live || d < 0 || d >= span || ( d < nS ) != part8 )
            return NONE;
        return (i32)( ( (u32)h << 16 ) | (u32)( 0xffff - d ) );
    };
    i32 kLo = keyOf( liveLo, tlo, hLo, true ), kHi = keyOf( liveHi, thi, hHi, true );
    // the class of a cell is (t - st0) mod 8 = (rr - st0 - j) mod 8: lanes 4 apart hold the same two classes
    kLo = max( kLo, dpp_ctrl<0x124>( kLo ) );
    kHi = max( kHi, dpp_ctrl<0x124>( kHi ) );
    kLo = max( kLo, dpp_ctrl<0x128>( kLo ) );
    kHi = max( kHi, dpp_ctrl<0x128>( kHi ) );
    const i32 hl = kLo >> 16, hh = kHi >> 16;
    const i32 tl = ( kLo != NONE && hl > hEn0 ) ? st0 + ( ( 0xffff - ( kLo & 0xffff ) ) & ~7 ) : en0;
    const i32 th = ( kHi != NONE && hh > hEn0 ) ? st0 + ( ( 0xffff - ( kHi & 0xffff ) ) & ~7 ) : en0;
    i32 vh = max( max( kLo != NONE ? hl : hEn0, kHi != NONE ? hh : hEn0 ), hEn0 ), vt = max( tl, th );
    i32 mc = min( kLo != NONE ? hl : NONE, kHi != NONE ? hh : NONE );
    vh = max( vh, dpp_ctrl<0xB1>( vh ) ); // the four lanes of a quad hold the eight classes
    vt = max( vt, dpp_ctrl<0xB1>( vt ) );
    mc = min( mc, dpp_ctrl<0xB1>( mc ) );
    vh = max( vh, dpp_ctrl<0x4E>( vh ) );
    vt = max( vt, dpp_ctrl<0x4E>( vt ) );
    mc = min( mc, dpp_ctrl<0x4E>( mc ) );
    mH = hEn0, mT = en0;
    minClass = 0x7fffffff; // no classes on this diagonal (fewer than 8 cells below en0): the position is H[en0]'s or a tail cell's
    if( nS > 0 )
        mH = vh, mT = vt, minClass = mc;
    // the cells after the 8-lane part, in the order the reference visits them: the first of the largest wins, if it is larger
    const i32 tk = grp_max_i32<16>( max( keyOf( liveLo, tlo, hLo, false ), keyOf( liveHi, thi, hHi, false ) ) );
    if( tk != NONE && ( tk >> 16 ) > mH )
    {
        mH = tk >> 16;
        mT = st0 + ( 0xffff - ( tk & 0xffff ) );
    }
    // band H of the cell the back-trace starts from
    {
        const i32 js = rr - mT;
        const i32 v = __builtin_amdgcn_ds_bpermute( ( ( lane - l ) + ( ( js & 31 ) >> 1 ) ) << 2, (i32)Hs );
        const i32 jj = __builtin_amdgcn_ds_bpermute( ( ( lane - l ) + ( ( js & 31 ) >> 1 ) ) << 2, (i32)Jpk );
        const i32 rowThere = ( js & 1 ) ? ( jj >> 16 ) & 0xffff : jj & 0xffff;
        const bool inBand = js >= 0 && js < qlen && mT >= 0 && mT < tlen && mT - js >= -KSW_BAND_B && mT - js <= KSW_BAND_B;
        hStart = inBand && rowThere == js ? ( ( js & 1 ) ? ( v >> 16 ) : ( (i32)( (u32)v << 16 ) >> 16 ) ) : NONE;
    }
}

// One set of up to G jobs (queue entries [at0, min(at0 + G, n))), all of them left- (LEFT) or right-aligned extensions.
// G = 4: the short jobs (B = 24, 16 lanes and 32 query rows each); G = 1: a long job (B = 120, 64 lanes, 128 query rows).
template <int G, bool LEFT, typename FETCH>
__device__ void ksw_band_set( const FETCH& F, const KswScoring& SC, const u32* list, u32 n, u32 at0, uint8_t* P /*KSW_GRP_ROWS x 128 B*/, uint8_t* lds,
                              const KswOut& O, KswWaveAcc& acc, u32* ext1, u32 nExt1, u32* ext2, u32 nExt2, unsigned int* extMore, unsigned long long* sOff,
                              u32* pf /*the wave's statistics*/, u32 rowsCap /*direction rows of P (LONG)*/ )
{
    constexpr int LANES = 64 / G, CJ = 2 * LANES, B = G == 1 ? KSW_BANDL_B : KSW_BAND_B;
    constexpr bool LONG = G == 1;
    static_assert( G == 1 || G == 4, "two shapes" );
    static_assert( B + 8 <= CJ, "the ring holds the band's rows and the rows being recycled" );
    const int lane = threadIdx.x & 63, g = lane / LANES, l = lane % LANES;
    uint8_t* stage = lds; // KSW_GRP_STAGE_ROWS x 128: direction rows (ksw_grp.h)
    u32* cigLds = (u32*)( lds + KSW_GRP_STAGE_ROWS * 128 ); // KSW_GRP_CIG_WORDS, CIGCAP per group
    uint8_t* tring = lds + KSW_GRP_STAGE_ROWS * 128 + KSW_GRP_CIG_WORDS * 4; // G x 2 CJ = 256 bytes: two blocks of CJ target bases per job
    uint8_t* qlds = tring + 4 * 64; // G = 4: 4 x 256, the jobs' queries; G = 1: a ring of two blocks of 128 query rows + 144 words for band_exact_max_lds
    i32* gflag = (i32*)( qlds + 4 * 256 );
    constexpr u32 CIGCAP = KSW_GRP_CIG_WORDS / G;
    // ---- the group's job
    const bool has = at0 + (u32)g < n;
    const u32 slot = list[ has ? at0 + g : at0 ];
    i32 qlen, tlen, rEnd, nDiag, zdrop, wJob;
    {
        const KswJobView J = F.view( slot );
        qlen = J.qlen, tlen = J.tlen, zdrop = J.zdrop, wJob = J.w;
        nDiag = qlen + tlen - 1;
        rEnd = LONG ? nDiag : min( nDiag, J.w + 1 ); // the job ends BEFORE diagonal rEnd: all diagonals done, or (short jobs) r > w (handed back)
    }
    auto qf = F.qfetch( slot );
    auto tf = F.tfetch( slot );
    typedef decltype( tf ) TF;
    // ---- scoring (as ksw_ext_core)
    int8_t q = (int8_t)SC.q, e = (int8_t)SC.e, q2 = (int8_t)SC.q2, e2 = (int8_t)SC.e2;
    const i32 sc_mch = (int8_t)( SC.match < 0 ? -SC.match : SC.match );
    const i32 sc_mis = (int8_t)( SC.mismatch > 0 ? -SC.mismatch : SC.mismatch );
    const i32 qe0 = q + e;
    if( q2 + e2 < q + e )
    {
        int8_t t = q;
        q = q2;
        q2 = t;
        t = e;
        e = e2;
        e2 = t;
    }
    const bool untouched = -( sc_mis < 0 ? sc_mis : 0 ) > 2 * ( q + e ); // kswcpp returns an untouched ez (kswcpp_core.h:340-341)
    i32 long_thres = e != e2 ? ( q2 - q ) / ( e - e2 ) - 1 : 0;
    if( q2 + e2 + long_thres * e2 > q + e + long_thres * e )
        ++long_thres;
    const i32 long_diff = long_thres * ( e - e2 ) - ( q2 - q ) - e2;
    auto initOf = [ & ]( i32 r ) -> i32 {
        return (int8_t)( r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2 );
    };
    auto hBoundary = [ & ]( i32 nn ) -> i32 { // H(nn-1, -1) = H(-1, nn-1) (ksw_ext.h)
        const i32 a = max( 0, min( nn, long_thres ) - 1 );
        const i32 hs = long_thres >= 1 && long_thres < nn ? 1 : 0;
        const i32 rest = ( nn - 1 ) - a - hs;
        return ( q + e ) - qe0 + ( nn < 1 ? 0 : -( q + e ) - e * a + ( hs ? long_diff : 0 ) - e2 * rest );
    };
    // the cost of leaving the band, and the offset between kswcpp's H and the path score ((sic) ksw_ext.h: H[0] is seeded with the
    // UNswapped q + e): the bounds below are on kswcpp's H
    const i32 gapOut = min( (i32)q + ( B + 1 ) * (i32)e, (i32)q2 + ( B + 1 ) * (i32)e2 );
    const i32 hOff = ( q + e ) - qe0;
    auto ubOf = [ & ]( i32 r ) -> i32 { // largest H a cell outside the band can have on diagonal r
        return r < B + 1 ? (i32)0x80000000 : sc_mch * ( ( r - B - 1 ) / 2 + 1 ) - gapOut + hOff;
    };
    constexpr u32 tS = LEFT ? 4 : 0, tX = LEFT ? 3 : 1, tY = 2, tX2 = LEFT ? 1 : 3, tY2 = 0;
    const u32 K_X0 = pk_val( -q - e, tX ), K_Y0 = pk_val( -q - e, tY ), K_X20 = pk_val( -q2 - e2, tX2 ), K_Y20 = pk_val( -q2 - e2, tY2 );
    const u32 K_TX = pk_val( 0, tX ), K_TY = pk_val( 0, tY ), K_TX2 = pk_val( 0, tX2 ), K_TY2 = pk_val( 0, tY2 );
    const u32 K_FX = pk_sub( K_TX, LEFT ? 0u : 0x00010001u ), K_FY = pk_sub( K_TY, LEFT ? 0u : 0x00010001u ),
              K_FX2 = pk_sub( K_TX2, LEFT ? 0u : 0x00010001u ), K_FY2 = pk_sub( K_TY2, LEFT ? 0u : 0x00010001u );
    const u32 K_NEG = 0x80008000u, K_MATCH = pk_bcast( sc_mch );
    const u32 V_CLIP = pk_val( sc_mch, 0xff );
    const u32 V_SCLO = ( (u32)sc_mch & 0xffu ) | ( ( (u32)sc_mis & 0xffu ) * 0x01010100u );
    const u32 V_SCHI = ( (u32)( -e2 ) & 0xffu ) | tS << 8;
    const u32 V_Q = pk_val( q, 0 ), V_Q2 = pk_val( q2, 0 ), V_QE = pk_val( q + e, 0 ), V_QE2 = pk_val( q2 + e2, 0 );
    const u32 K_GAP = pk_val( -q - e, 0 ); // u / v of a cell whose neighbour lies outside the band
    // ---- the query in LDS (a lane takes new rows every 64 diagonals), two blocks of the target
    uint8_t* myQ = qlds + g * 256;
    auto fillQ = [ & ]( i32 blk ) { // LONG: query rows [128 blk, 128 blk + 128) -> ring slot blk & 1
        const i32 j = 128 * blk + 2 * l;
        const u32 b0 = has && j < qlen ? (u32)qf( j ) : 4u, b1 = has && j + 1 < qlen ? (u32)qf( j + 1 ) : 4u;
        *(uint16_t*)( myQ + ( j & 255 ) ) = (uint16_t)( b0 | b1 << 8 );
    };
    if( LONG )
    {
        fillQ( 0 );
        fillQ( 1 );
    }
    else
        for( i32 j = l; j < 256; j += LANES )
            myQ[ j ] = has && j < qlen ? (uint8_t)qf( j ) : (uint8_t)4;
    i32 nextQ = 2; // LONG: block k is written on diagonal 256 (k - 1): after the last lane took its rows of block k - 2 (diagonal
                   // 256 (k - 2) + 4 * 63 + B + 3), before the first lane takes rows of block k (256 (k - 1) + B + 3)
    auto tgt2 = [ & ]( i32 t ) -> u32 { // target bases of cells t, t + 1 (codes as in ksw_ext.h: an N of the target is 12)
        if( t >= tlen )
            return 0u;
        u32 ab = tf.pair( t ) & ( t + 1 < tlen ? 0x00ff00ffu : 0x000000ffu );
        if( TF::CLEAN )
            return ab;
        const u32 nn = pk_lshr( ab, 2 );
        return pk_bfi( pk_sub( 0u, pk_minu( nn, 0x00010001u ) ), 0x000c000cu, ab );
    };
    uint8_t* myRing = tring + g * ( 2 * CJ );
    auto fillBlock = [ & ]( i32 blk ) { // target bases [CJ blk, CJ blk + CJ) -> ring slot blk & 1
        const i32 t = CJ * blk + 2 * l;
        const u32 ab = has ? tgt2( t ) : 0u;
        *(uint16_t*)( myRing + ( t & ( 2 * CJ - 1 ) ) ) = (uint16_t)( ( ab & 0xffu ) | ( ab >> 8 & 0xff00u ) );
    };
    fillBlock( 0 );
    fillBlock( 1 );
    i32 nextBlk = 2; // the next block the ring takes: when the band's upper edge (r + B) / 2 reaches it
    // ---- per lane: rows j = 2 l, 2 l + 1 (+ 32 per recycling)
    u32 Jpk = (u32)( 2 * l ) | (u32)( 2 * l + 1 ) << 16;
    auto rowState = [ & ]( u32 jpk, u32& Jmask, u32& Qb, u32& V, u32& H, u32& entOk ) {
        const i32 j0 = (i32)( jpk & 0xffffu ), j1 = (i32)( jpk >> 16 );
        const bool ok = has && !untouched;
        Jmask = ( ok && j0 < qlen ? 0x0000ffffu : 0u ) | ( ok && j1 < qlen ? 0xffff0000u : 0u );
        Qb = (u32)myQ[ j0 & 255 ] | (u32)myQ[ j1 & 255 ] << 16;
        // rows up to B start at the first column (kswcpp_core.h:562-585); the others enter through the band's edge: their upper
        // neighbour is a cell outside the band
        const u32 vLo = j0 <= B ? (u32)initOf( j0 ) & 0xffu : (u32)( -q - e ) & 0xffu, vHi = j1 <= B ? (u32)initOf( j1 ) & 0xffu : (u32)( -q - e ) & 0xffu;
        V = vLo << 8 | vHi << 24;
        H = ( (u32)hBoundary( j0 + 1 ) & 0xffffu ) | (u32)hBoundary( j1 + 1 ) << 16; // H(-1, j): read by the rows up to B only
        entOk = ( j0 > B ? 0x0000ffffu : 0u ) | ( j1 > B ? 0xffff0000u : 0u );
    };
    u32 Jmask, Qb, V, H, entOk;
    __syncthreads( ); // the query bytes
    rowState( Jpk, Jmask, Qb, V, H, entOk );
    u32 X = K_X0, X2 = K_X20, U = K_GAP, Y = K_Y0, Y2 = K_Y20;
    u32 Tpk = pk_sub( 0u, Jpk ); // t = r - j of the lane's cells
    u32 leadLo = l == 0 ? 0x0000ffffu : 0u; // the cell that takes the first-row boundary: row 0
    const u32 tlenpk = pk_bcast( tlen );
    const u32 K_B = pk_bcast( B ), K_2B1 = pk_bcast( 2 * B + 1 );
    // ---- per group, equal in all lanes of the group (0 / -1 words, ksw_grp.h)
    i32 ezmax = 0, maxT = -1, maxQ = -1, pR = 0, pM = 0;
    u32 ezpk = 0, snapH = 0, snapJ = Jpk;
    // (LONG: the launch's scratch is planned for a full set of waves; a job that needs more direction rows goes on to the exact kernels)
    const bool tooBig = LONG && has && (u32)KSW_BANDL_ROWS( min( qlen, tlen ) ) > rowsCap;
    i32 act = has && !untouched && !tooBig ? -1 : 0, handBack = tooBig ? -1 : 0, pend = 0;
    i32 zBad = 0; // check 4 failed
    i32 boundPrev = 0x7fffffff, nextBound = 0;
    const i32 boundRate = max( 1, ( -sc_mis + sc_mch + 1 ) / 2 );
    const i32 qe = q + e;
    i32 rLast = -1;
    const u32 laneOff = (u32)( g * CJ + 2 * l );
    i32 uInS = initOf( 0 );
    i32 ringLo = 0;
    // the last diagonal with a band cell: row qlen - 1 leaves the band after t = qlen - 1 + B
    const i32 rBand = min( 2 * ( qlen - 1 ) + B, 2 * ( tlen - 1 ) + B );
    __syncthreads( );
    i32 r = 0;
    for( ;; ++r )
    {
        // ---- a job ends before this diagonal: all diagonals done, the band has left the rectangle, or it leaves the regime
        {
            const i32 ends = r >= rEnd || r > rBand ? act : 0;
            handBack |= r >= rEnd && rEnd < nDiag && r <= rBand ? ends : 0;
            rLast = ends ? r - 1 : rLast;
            act &= ~ends;
            Jmask &= (u32)~ends;
        }
        if( !__any( act != 0 ) )
            break;
        if( __builtin_expect( r <= long_thres + 1, 0 ) )
            uInS = initOf( r );
        // ---- the next 32 target bases, when the band's upper edge reaches them (wave-uniform: r and B are)
        if( __builtin_expect( ( ( r + B ) >> 1 ) >= CJ * nextBlk - 1, 0 ) )
        {
            fillBlock( nextBlk );
            nextBlk++;
            __syncthreads( );
        }
        if( LONG && __builtin_expect( r == 256 * ( nextQ - 1 ), 0 ) )
        {
            fillQ( nextQ );
            nextQ++;
            __syncthreads( );
        }
        if( __builtin_expect( r >= KSW_GRP_STAGE_ROWS && ( r & ( KSW_GRP_STAGE_ROWS / 2 - 1 ) ) == 0, 0 ) )
        {
            // rows [r - 32, r - 16) -> HBM (ksw_grp.h)
            __syncthreads( );
            const uint4* src = (const uint4*)( stage + ( ( r - KSW_GRP_STAGE_ROWS ) & ( KSW_GRP_STAGE_ROWS - 1 ) ) * 128 );
            uint4* dst = (uint4*)( P + (size_t)( r - KSW_GRP_STAGE_ROWS ) * 128 );
#pragma unroll
            for( int k = 0; k < KSW_GRP_STAGE_ROWS / 16; k++ )
                dst[ lane + 64 * k ] = src[ lane + 64 * k ];
            ringLo = r - KSW_GRP_STAGE_ROWS / 2;
        }
        // ---- neighbours: u, y, y2 come from row j - 1 = the previous lane of the ring; the H of row j - 1 for a row that enters
        const u32 ut0 = cells_shift1( U, band_ror1<LANES>( U ) );
        const u32 yt0 = cells_shift1( Y, band_ror1<LANES>( Y ) );
        const u32 y2t0 = cells_shift1( Y2, band_ror1<LANES>( Y2 ) );
        const u32 hPrev = cells_shift1( H, band_ror1<LANES>( H ) );
        const u32 ut = pk_bfi( leadLo, ( (u32)uInS & 0xffu ) << 8, ut0 );
        const u32 yt = pk_bfi( leadLo, K_Y0, yt0 );
        const u32 y2t = pk_bfi( leadLo, K_Y20, y2t0 );
        // ---- a lane whose rows have both left the band takes the rows CJ further on: lane l on diagonal 4 l + B + 3 (+ 2 CJ k).
        // AFTER the shifts above: what its last live cell computed on the diagonal before is read by the next lane on this one
        if( __builtin_expect( r >= B + 3 && ( ( r - B - 3 ) & 3 ) == 0, 0 ) )
        {
            if( l == ( ( ( r - B - 3 ) >> 2 ) & ( LANES - 1 ) ) )
            {
                Jpk = pk_add( Jpk, pk_bcast( CJ ) );
                rowState( Jpk, Jmask, Qb, V, H, entOk );
                Jmask &= (u32)act;
                X = K_X0, X2 = K_X20, U = K_GAP, Y = K_Y0, Y2 = K_Y20;
                Tpk = pk_sub( pk_bcast( r ), Jpk );
                leadLo = 0;
            }
        }
        // the target bases of the lane's cells: t (low half) and t - 1
        const i32 tLo = (i32)( (u32)( Tpk << 16 ) ) >> 16;
        const u32 tt = (u32)myRing[ tLo & ( 2 * CJ - 1 ) ] | (u32)myRing[ ( tLo - 1 ) & ( 2 * CJ - 1 ) ] << 16;
        // ---- live cells: 0 <= t <= tlen - 1 on a row of the job, -B <= t - j <= B
        const u32 dB = pk_add( pk_sub( Tpk, Jpk ), K_B ); // t - j + B: 0 .. 2 B inside the band
        const u32 inBand = pk_nonzero15( pk_subsatu( K_2B1, dB ) );
        const u32 LM = pk_opaque( pk_nonzero15( pk_subsatu( tlenpk, Tpk ) ) & Jmask & inBand );
        const u32 ENT = LM & entOk & ~pk_nonzero15( dB ); // a cell on the band's edge t - j = -B of a row that did not start at t = 0
        // ---- score and DP cell (kswcpp_core.h:598-766; ksw_ext.h)
        const u32 sel = ( pk_minu( tt ^ Qb, 0x00040004u ) << 8 ) | 0x00050005u;
        u32 z = __builtin_amdgcn_perm( V_SCHI, V_SCLO, sel );
        u32 a = pk_add( X, V );
        u32 b = pk_add( yt, ut );
        u32 a2 = pk_add( X2, V );
        u32 b2 = pk_add( y2t, ut );
        u32 d;
        if( LEFT )
        {
            z = pk_max( pk_max( z, a ), pk_max( pk_max( b, a2 ), b2 ) );
            d = pk_sub( 0x00040004u, z & 0x00070007u );
        }
        else
        {
            z = pk_max( pk_max( z, a ), pk_max( b, a2 ) );
            d = z & 0x00070007u;
            z = pk_max( z, b2 );
        }
        const u32 zc = pk_min( z, V_CLIP ) & 0xff00ff00u;
        const u32 nu = pk_sub( zc, V ), nv = pk_sub( zc, ut );
        u32 tmp = pk_sub( zc, V_Q );
        a = pk_sub( a, tmp );
        b = pk_sub( b, tmp );
        tmp = pk_sub( zc, V_Q2 );
        a2 = pk_sub( a2, tmp );
        b2 = pk_sub( b2, tmp );
        const u32 nx = pk_sub( pk_max( a, K_TX ), V_QE ), ny = pk_sub( pk_max( b, K_TY ), V_QE );
        const u32 nx2 = pk_sub( pk_max( a2, K_TX2 ), V_QE2 ), ny2 = pk_sub( pk_max( b2, K_TY2 ), V_QE2 );
        const u32 fa = pk_sub( K_FX, a ), fb = pk_sub( K_FY, b ), fa2 = pk_sub( K_FX2, a2 ), fb2 = pk_sub( K_FY2, b2 );
        d = and_or( fa >> 12, 0x00080008u, and_or( fb >> 11, 0x00100010u, and_or( fa2 >> 10, 0x00200020u, and_or( fb2 >> 9, 0x00400040u, d ) ) ) );
        // ---- commit.  A cell that is not a cell of the band hands its right neighbour the values of a cell reached by a gap
        // (u = -(q+e), no gap open); its own v, x, x2 keep their initialisation until the cell is born
        U = pk_bfi( LM, nu, K_GAP );
        Y = pk_bfi( LM, ny, K_Y0 );
        Y2 = pk_bfi( LM, ny2, K_Y20 );
        V = pk_bfi( LM, nv, V );
        X = pk_bfi( LM, nx, X );
        X2 = pk_bfi( LM, nx2, X2 );
        if( LM ) // the lane's two bytes of row r of the ring
            *(uint16_t*)( stage + ( r & ( KSW_GRP_STAGE_ROWS - 1 ) ) * 128 + laneOff ) = (uint16_t)__builtin_amdgcn_perm( 0u, d, 0x0c0c0200u );
        // ---- H(t, j) = H(t-1, j) + u(t, j); a row that enters the band: H(t, j-1) + v(t, j)
        const u32 hn = pk_bfi( ENT, pk_add( hPrev, pk_ashr8( nv ) ), pk_add( H, pk_ashr8( nu ) ) );
        H = pk_bfi( LM, hn, H );
        const u32 Hm = pk_bfi( LM, hn, K_NEG );
        Tpk = pk_add( Tpk, 0x00010001u );
        // ---- a larger maximum: the value now, its position when somebody asks (ksw_ext.h)
        i32 raise = 0;
        if( __any( pk_max( Hm, ezpk ) != ezpk ) )
        {
            // (LONG: one job per wave -- the maximum and everything derived from it, ez.max, the last raise, the schedule of the early
            // stop, is wave-uniform: read into a scalar register, the bookkeeping below runs on the scalar unit)
            const i32 gmV = grp_max_i32<LANES>( max( (i32)( Hm << 16 ) >> 16, (i32)Hm >> 16 ) );
            const i32 gm = LONG ? __builtin_amdgcn_readfirstlane( gmV ) : gmV;
            raise = gm > ezmax ? -1 : 0;
            // check 4: between the last raise (pR, pM) and this diagonal the wide run's ez.max - diagonal maximum stayed <= zdrop
            // (LONG: by (L) the wide run's ez.max up to this diagonal is at most max(the band's ez.max so far, UB(r)))
            if( raise && pend && ( LONG ? max( pM, ubOf( r ) ) : sc_mch * ( r / 2 + 1 ) + hOff ) - pM + ( r - pR ) * qe > zdrop )
                zBad = -1;
            // ... and BEFORE the first raise: ksw_apply_zdrop (kswcpp_core.h:22-44) also fires while ez.max is still 0 and max_t = max_q
            // = -1 (every cell passes t >= max_t && q >= max_q).  The maximum of diagonal r is at least that of a cell of the first
            // column reached by one gap from the boundary, -(2 q + (r + 2) e) >= -(r + 2)(q + e) (+ hOff in kswcpp's H); the wide run's
            // ez.max up to diagonal r1 - 1 is 0 while no cell outside the band exists (r < B + 1), at most a (r / 2 + 1) + hOff after
            if( raise && !pend && r > 0 && max( 0, r - 1 < B + 1 ? 0 : sc_mch * ( ( r - 1 ) / 2 + 1 ) + hOff ) + ( r + 1 ) * qe - hOff > zdrop )
                zBad = -1;
            ezmax = raise ? gm : ezmax;
            ezpk = raise ? pk_bcast( gm ) : ezpk;
            snapH = raise ? H : snapH;
            snapJ = raise ? Jpk : snapJ;
            pR = raise ? r : pR;
            pM = raise ? gm : pM;
            pend |= raise;
        }
        // ---- early stop (ksw_reg.h) on the band's cells, every job on its own schedule (ksw_grp.h)
        // (LONG: the chain of a band cell starts on a diagonal <= B + 1, so the bound over the band's cells holds from B + 3 on; it is
        // looked at once the band has reached the last rows or columns)
        const i32 rOk = LONG ? B + 3 : qlen;
        const bool due = ( act & ~raise ) != 0 && r >= rOk - 1 && r >= nextBound && ( !LONG || r >= min( qlen, tlen ) - 1 );
        if( __any( due ) )
        {
            const u32 QLpk = pk_sub( pk_bcast( qlen - 1 ), Jpk ); // rows left below the cell
            const u32 pot = pk_min( QLpk, pk_sub( tlenpk, Tpk ) ); // min( rows left, columns left ): Tpk is t + 1 by now
            const u32 bnd = pk_mad( pot, K_MATCH, H );
            const u32 bm = pk_bfi( LM, bnd, K_NEG );
            const i32 boundV = grp_max_i32<LANES>( max( (i32)( bm << 16 ) >> 16, (i32)bm >> 16 ) );
            const i32 bound = LONG ? __builtin_amdgcn_readfirstlane( boundV ) : boundV;
            const i32 top = LONG ? (i32)0x80000000 : hBoundary( r ) + sc_mch * qlen;
            if( due )
            {
                const i32 all = max( max( bound, boundPrev ), top );
                if( r >= rOk && all <= ezmax )
                {
                    rLast = r;
                    act = 0;
                    Jmask = 0;
                }
                else if( boundPrev != 0x7fffffff && r >= rOk )
                {
                    nextBound = r + 1 + max( 0, ( max( bound, top ) - ezmax ) / boundRate - 1 );
                    boundPrev = 0x7fffffff;
                }
                else
                    boundPrev = bound;
            }
            else
                boundPrev = 0x7fffffff;
        }
        else
            boundPrev = 0x7fffffff;
    }
    // ---- position of the last raise, and the checks
    i32 why = 0; // 1..4: the check that failed
    {
        i32 pH, pT, minClass, hStart;
        if( LONG )
            band_exact_max_lds( snapH, snapJ, pR, qlen, tlen, wJob, ksw_h16( SC, qlen, tlen ) ? 8 : 4, lane, (i32*)( qlds + 256 ), pH, pT, minClass, hStart );
        else
            band_exact_max( snapH, snapJ, pR, qlen, tlen, wJob, lane, l, pH, pT, minClass, hStart );
        if( pend )
        {
            maxT = pT;
            maxQ = pR - pT;
            const i32 ub = ubOf( pR );
            if( !( ezmax > sc_mch * ( LONG ? max( min( tlen - B - 1, qlen ), min( qlen - B - 1, tlen ) ) : qlen ) - gapOut + hOff ) )
                why = 1;
            else if( !( minClass != (i32)0x80000000 && minClass > ub && pH > ub ) && pR >= B + 1 )
                why = 2;
            else if( !( hStart != (i32)0x80000000 && hStart > ub ) )
                why = 3;
            else if( zBad || ( !LONG && sc_mch * ( max( rLast, pR ) / 2 + 1 ) + hOff - pM + ( max( rLast, pR ) - pR ) * qe > zdrop ) )
                why = 4; // (LONG: a z-drop behind the last raise changes nothing the callers read)
        }
        else if( has && !untouched && !handBack )
            why = 1; // no cell ever exceeded 0: nothing proves that none outside the band does
    }
    handBack |= why ? -1 : 0;
    __syncthreads( ); // direction bytes (LDS, and the rows that went to HBM) visible to the back-trace
    // ---- back-trace (ksw_grp.h); a step outside the band cannot happen after check 3 -- it hands the job back all the same
    const bool leader = l == 0 && has && !handBack && !untouched && maxT >= 0 && maxQ >= 0;
    const i32 revCigar = F.view( slot ).flag & KSW_EZ_REV_CIGAR;
    u32* myCig = cigLds + g * CIGCAP;
    u32 nCig = 0, curOp = 3, curLen = 0, steps = 0;
    bool cigOver = false;
    i32 bi = leader ? maxT : -1, bj = leader ? maxQ : -1, state = 0;
    auto pushRun = [ & ]( u32 op, u32 len ) {
        if( op == curOp )
            curLen += len;
        else
        {
            if( curLen )
            {
                if( nCig < CIGCAP )
                    myCig[ nCig ] = curLen << 4 | curOp;
                else
                    cigOver = true;
                nCig++;
            }
            curOp = op;
            curLen = len;
        }
    };
    i32 winLo = ringLo;
    while( true )
    {
        while( bi >= 0 && bj >= 0 && bi + bj >= winLo )
        {
            if( bi - bj > B || bj - bi > B )
            {
                cigOver = true; // (handed back)
                bi = bj = -1;
                break;
            }
            const u32 tb = stage[ ( ( bi + bj ) & ( KSW_GRP_STAGE_ROWS - 1 ) ) * 128 + g * CJ + ( bj & ( CJ - 1 ) ) ];
            if( state != 0 && !( ( tb >> ( state + 2 ) ) & 1 ) )
                state = 0;
            if( state == 0 )
                state = tb & 7;
            const u32 op = state == 0 ? 0u : ( ( state == 1 || state == 3 ) ? 2u : 1u );
            steps++;
            pushRun( op, 1 );
            bi -= op != 1u ? 1 : 0;
            bj -= op != 2u ? 1 : 0;
        }
        const bool walking = bi >= 0 && bj >= 0;
        if( !__any( walking ) )
            break;
        const i32 rhi = wave_max_i32( walking ? bi + bj : -1 );
        winLo = max( 0, rhi - KSW_GRP_STAGE_ROWS + 1 );
        __syncthreads( );
        for( i32 k = lane; k < ( rhi - winLo + 1 ) * 8; k += 64 ) // 16 bytes each: 8 per row
        {
            const i32 row = winLo + ( k >> 3 );
            *(uint4*)( stage + ( row & ( KSW_GRP_STAGE_ROWS - 1 ) ) * 128 + ( k & 7 ) * 16 ) = *(const uint4*)( P + (size_t)row * 128 + ( k & 7 ) * 16 );
        }
        __syncthreads( );
    }
    if( leader && !cigOver )
    {
        if( bi >= 0 )
            pushRun( 2, (u32)( bi + 1 ) );
        if( bj >= 0 )
            pushRun( 1, (u32)( bj + 1 ) );
        if( curLen )
        {
            if( nCig < CIGCAP )
                myCig[ nCig ] = curLen << 4 | curOp;
            else
                cigOver = true;
            nCig++;
        }
    }
    // ---- publish (ksw_grp.h): one pool reservation for the set, every group copies its cigar
    gflag[ g ] = 0;
    __syncthreads( );
    if( l == 0 )
        gflag[ g ] = (i32)( cigOver ? 0x40000000u : nCig );
    __syncthreads( );
    const u32 gw = (u32)gflag[ g ];
    const bool over = ( gw & 0x40000000u ) != 0;
    const u32 myN = over ? 0u : gw;
    const bool publish = has && !handBack && !over;
    u32 total = 0, before = 0;
#pragma unroll
    for( int k = 0; k < G; k++ )
    {
        const u32 v = (u32)gflag[ k ];
        const u32 c = ( v & 0x40000000u ) ? 0u : v;
        const bool pub = (u32)__builtin_amdgcn_readlane( (i32)( publish ? 1 : 0 ), k * LANES ) != 0;
        if( k < g )
            before += pub ? c : 0u;
        total += pub ? c : 0u;
    }
    total = (u32)__builtin_amdgcn_readfirstlane( (i32)total );
    u64 off0;
    if( O.cig_chunk == 0 || total > O.cig_chunk )
    {
        if( threadIdx.x == 0 )
            *sOff = atomicAdd( O.cig_used, (unsigned long long)total );
        __syncthreads( );
        off0 = *sOff;
        __syncthreads( );
    }
    else
    {
        if( total > acc.chunk_left )
        {
            if( threadIdx.x == 0 )
                *sOff = atomicAdd( O.cig_used, (unsigned long long)O.cig_chunk );
            __syncthreads( );
            acc.chunk_off = *sOff;
            acc.chunk_left = O.cig_chunk;
            __syncthreads( );
        }
        off0 = acc.chunk_off;
        acc.chunk_off += total;
        acc.chunk_left -= total;
    }
    const u64 off = off0 + before;
    const bool fits = off + myN <= O.cig_pool_cap;
    // cells the job computed: the band cells of the diagonals 0 .. rLast (the group's lanes take the diagonals in turn)
    u64 cellsJob = 0;
    if( __any( has && rLast >= 0 && publish ) ) // (wave-uniform for one job per wave; the sum needs all lanes of the group)
    {
        u32 c = 0;
        for( i32 rr = l; rr <= rLast && has && publish; rr += LANES )
        {
            const i32 lo = max( max( 0, rr - tlen + 1 ), ( rr - B + 1 ) >> 1 ), hi = min( min( qlen - 1, rr ), ( rr + B ) >> 1 );
            c += hi >= lo ? (u32)( hi - lo + 1 ) : 0u;
        }
        cellsJob = band_sum_u32<LANES>( c );
    }
    if( l == 0 && publish )
    {
        ma_ez rz;
        rz.max = untouched ? 0 : ( ezmax & 0x7fffffff );
        rz.zdropped = 0;
        rz.max_q = maxQ;
        rz.max_t = maxT;
        rz.mqe = (i32)0x80000000;
        rz.mqe_t = -1;
        rz.mte = (i32)0x80000000;
        rz.mte_q = -1;
        rz.score = (i32)0x80000000;
        rz.reach_end = 0;
        rz.n_cigar = (i32)myN;
        O.ez[ slot ] = rz;
        O.cig_off[ slot ] = off;
        if( !fits )
            atomicOr( O.err, MA_ERR_CIGAR_OVERFLOW );
    }
    if( l == 0 && has && ( handBack || over ) )
    {
        // to the list of the extension kernel the job would have gone to (k_ksw_ext<1> / <2> run after this kernel on the same
        // stream and read the number of appended jobs from extMore[ 0 ] / [ 1 ])
        if( LONG ) // to the jobs handed back to the exact kernels: ext1 = that list, extMore = its counter (second pass of ksw_run_all)
            ext1[ atomicAdd( extMore, 1u ) ] = slot;
        else if( ksw_ext_slots( SC, qlen, tlen, wJob, zdrop, F.view( slot ).flag ) == 2 )
            ext2[ nExt2 + atomicAdd( extMore + 1, 1u ) ] = slot;
        else
            ext1[ nExt1 + atomicAdd( extMore + 0, 1u ) ] = slot;
    }
    if( publish && fits )
        for( u32 i = (u32)l; i < myN; i += LANES ) // the walk leaves the cigar reversed (kswcpp_core.h:146-149)
            O.cig_pool[ off + i ] = revCigar ? myCig[ i ] : myCig[ myN - 1 - i ];
    {
        u64 c = l == 0 && publish ? cellsJob : 0, p = l == 0 && publish ? (u64)steps : 0, nj = l == 0 && publish ? 1 : 0, cw = l == 0 && publish ? myN : 0;
#pragma unroll
        for( int k = 0; k < G; k++ )
        {
            acc.cells += (u32)__builtin_amdgcn_readlane( (i32)(u32)c, k * LANES );
            acc.path += (u32)__builtin_amdgcn_readlane( (i32)(u32)p, k * LANES );
            acc.njobs += (u32)__builtin_amdgcn_readlane( (i32)(u32)nj, k * LANES );
            acc.cig_words += (u32)__builtin_amdgcn_readlane( (i32)(u32)cw, k * LANES );
        }
    }
    __syncthreads( ); // LDS and the scratch rows are free for the next set
    if( l == 0 && has ) // (per group leader: summed over the leaders at the kernel's end)
    {
        pf[ 0 ] += 1;
        pf[ 1 ] += publish ? 1 : 0;
        pf[ 2 ] += why == 1, pf[ 3 ] += why == 2, pf[ 4 ] += why == 3, pf[ 5 ] += why == 4;
        pf[ 6 ] += !publish && !why ? 1 : 0;
        pf[ 7 ] += (u32)( rLast + 1 );
    }
}

template <typename FETCH, bool LEFT, int G = 4>
__global__ void __launch_bounds__( 64 ) __attribute__( ( amdgpu_waves_per_eu( 5, 5 ) ) )
k_ksw_band( FETCH F, KswScoring SC, const u32* list, u32 n, unsigned int* next, uint8_t* scratch, u64 stride, KswOut O, u32* ext1, u32 nExt1, u32* ext2,
            u32 nExt2, unsigned int* extMore )
{
    __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t lds[ KSW_BAND_LDS ];
    __shared__ u32 sSet;
    __shared__ unsigned long long sOff;
    uint8_t* P = scratch + (u64)blockIdx.x * stride;
    KswWaveAcc acc;
    u32 stats[ 8 ] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    u32 cur = 0, end = 0;
    while( true )
    {
        if( cur >= end )
        {
            if( threadIdx.x == 0 )
                sSet = atomicAdd( next, (unsigned int)( 4 * G ) );
            __syncthreads( );
            cur = sSet;
            __syncthreads( );
            if( cur >= n )
                break;
            end = cur + 4 * G < n ? cur + 4 * G : n;
        }
        ksw_band_set<G, LEFT>( F, SC, list, n, cur, P, lds, O, acc, ext1, nExt1, ext2, nExt2, extMore, &sOff, stats, (u32)( stride / 128 ) );
        cur += G;
    }
    ksw_flush( O, acc, G == 1 ? 7 : 4 );
    for( int i = 0; i < 8; i++ )
    {
        u32 v = 0;
        for( int k = 0; k < 4; k++ )
            v += (u32)__builtin_amdgcn_readlane( (i32)stats[ i ], k * 16 ); // (G = 1: lanes 16, 32, 48 hold zeros)
        if( threadIdx.x == 0 && v )
            atomicAdd( g_band_stats + ( G == 1 ? 8 : 0 ) + i, (unsigned long long)v );
    }
}
} // namespace ma
#endif
