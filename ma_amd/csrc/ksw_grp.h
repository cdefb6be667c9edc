// ksw_grp.h -- SEVERAL short extension jobs per wavefront (round 5).
//
// The extension kernel of ksw_ext.h gives every job a whole wavefront: 128 cells per diagonal, of which a job with a query
// of q bases uses at most q.  68 % of the extension jobs of a 150 bp batch have q + 2 <= 64 (the read ends a seed left over),
// and for those most of the per-job cost is not even the diagonals but the job itself: descriptor, sequences, state
// initialisation, back-trace, result record (profiles/r04_ext_pairing_bound.txt: 25.7 % of the kernel's time).  Here a
// wavefront takes G = 2 or 4 jobs at once -- a GROUP of 64 / G lanes (= 128 / G cells) each -- and runs them in lock-step:
// the diagonal number r is wave-uniform, everything else that belongs to a job lives in VGPRs, equal in all lanes of its group.
//
// Layout.  ksw_ext.h keeps the TARGET position t of a cell fixed in its lane and lets the query flow through the lanes, which
// needs a ring (cells that left the band are recycled for cells further up the target).  A short job knows its live cells in
// advance: the band never cuts the rectangle (qlen <= w + 1), so on diagonal r the live cells are the query rows
// j = 0 .. qlen-1 with 0 <= t = r - j <= tlen - 1.  Cell (t, j) therefore lives in half (j & 1) of lane (j >> 1) of its group
// for the whole job: the QUERY stays put, the target flows through the lanes, and nothing is ever recycled.
// In kswcpp's arrays (indexed by t) cell (t, j) reads x, v, x2 of cell (t-1, j) -- the SAME lane here -- and u, y, y2 of cell
// (t, j-1) -- the previous lane: three DPP shifts (+ the target base) per diagonal instead of five, none of them circular.
//   first row (j = -1):    the group's first lane takes u = first-row initialisation of cell t = r, y = -q-e, y2 = -q2-e2
//                          (kswcpp_core.h:562-585) and target base r out of a 2 x CJ byte ring in LDS
//   first column (t = -1): v = first-column initialisation of row j, x = -q-e, x2 = -q2-e2 and the boundary score H(-1, j) are
//                          the lanes' initial state and stay until the cell is born (commit under the live mask)
// H(t, j) = H(t-1, j) + u(t, j) needs no shift at all.
//
// What is computed is what ksw_ext.h computes (its header: the callers read max, max_q, max_t and the cigar only; every cell
// of the regime is a true DP cell; early stop of ksw_reg.h), with the same packed int8-in-int16 arithmetic and tags.  The
// position of a maximum follows calcMaxScore (kswcpp_core.h:156-299) class by class, evaluated lazily on the lanes' H of the
// last raise, which stays in a register per lane.  A job that would leave the regime (r > w), or whose cigar outgrows its LDS
// buffer, is handed back to the exact kernel like ksw_ext.h does.
#pragma once
#include "ksw_ext.h"

#if defined( __HIPCC__ )
namespace ma
{
#if defined( MA_KSW_PROF )
// diagnostics build: wave cycles per phase of k_ksw_grp, by jobs per wave G (index 8 * log2 G + phase): 0 set-up (descriptors,
// sequences, state), 1 diagonal loop, 2 position + back-trace, 3 publish, 4 diagonals run, 5 sets, 6 jobs, 7 sum of the jobs'
// OWN diagonals (what the loop would run with perfectly matched sets = phase 4 x G)
static __device__ unsigned long long g_grp_prof[ 24 ];
#define GRP_PROF_T( v ) const unsigned long long v = clock64( )
#else
#define GRP_PROF_T( v )
#endif
#define KSW_GRP_STAGE_ROWS 32 // direction rows (128 B each) of the LDS ring: the last 17..32 diagonals of a set never leave the CU
#define KSW_GRP_CIG_WORDS 256 // cigar words in LDS per wavefront (shared by its G groups)
#define KSW_GRP_ROWS 516 // direction rows of a wavefront's scratch: a job leaves the regime at r > w, w <= 512
#define KSW_GRP_LDS ( KSW_GRP_STAGE_ROWS * 128 + KSW_GRP_CIG_WORDS * 4 + 512 + 64 ) // (the target rings: 256 B per packed register of a lane)

// jobs per wavefront this job can share a wavefront with (0: not a job for this kernel).  Same regime as ksw_ext_slots,
// extension jobs only (a global job's time is its set-up, and its callers read another cell).
MA_HD int ksw_grp_size( const KswScoring& SC, i32 qlen, i32 tlen, i32 w, i32 zdrop, i32 flag )
{
    if( !( flag & KSW_EZ_EXTZ_ONLY ) || w > 512 || ksw_ext_slots( SC, qlen, tlen, w, zdrop, flag ) != 1 )
        return 0;
    if( zdrop > 16000 )
        return 0; // the z-drop threshold is kept as a packed int16
    // 4 / 2 jobs per wave with two rows per lane; 1 = two jobs per wave with FOUR rows per lane (MA_KSW_GRP=1 leaves those to k_ksw_ext<1>)
    return qlen <= 32 ? 4 : ( qlen <= 64 ? 2 : ( qlen <= 128 && SC.grp == 2 ? 1 : 0 ) );
}

template <int LANES> __device__ __forceinline__ i32 grp_max_i32( i32 v ) // maximum over the 64 / G lanes of a group, in every lane
{
    v = max( v, dpp_ctrl<0x121>( v ) ); // row_ror:1
    v = max( v, dpp_ctrl<0x122>( v ) );
    v = max( v, dpp_ctrl<0x124>( v ) );
    v = max( v, dpp_ctrl<0x128>( v ) );
    if( LANES >= 32 )
    {
        auto a = __builtin_amdgcn_permlane16_swap( (u32)v, (u32)v, false, false );
        v = max( (i32)a[ 0 ], (i32)a[ 1 ] );
    }
    if( LANES == 64 )
    {
        auto a = __builtin_amdgcn_permlane32_swap( (u32)v, (u32)v, false, false );
        v = max( (i32)a[ 0 ], (i32)a[ 1 ] );
    }
    return v;
}

// calcMaxScore (kswcpp_core.h:156-299) of diagonal rr of every group at once, from the lanes' packed H (half lo: row j = 2 l,
// half hi: row 2 l + 1): the 8 classes (t - st0) mod 8 over the chunks [st0, st0 + nS) keep their first maximum and the chunk
// base it came from, (H[en0], en0) wins ties, max_t is the largest of the classes' positions (independent horizontal
// maxima, sic); then the cells [st0 + nS, en0) one by one.  All values are per lane and equal within a group.
template <int LANES, int NR>
__device__ __forceinline__ void grp_exact_max( const u32 ( &Hs )[ NR ], const u32 ( &Jpk )[ NR ], i32 rr, i32 qlen, i32 tlen, int lane, int l, i32& mH,
                                               i32& mT )
{
    const i32 st0 = max( 0, rr - qlen + 1 ), en0 = min( rr, tlen - 1 );
    const i32 j0 = rr - en0; // row of cell en0: half (j0 & 1) of register (j0 >> 1) % NR of lane j0 / (2 NR)
    i32 hv = 0;
#pragma unroll
    for( int k = 0; k < NR; k++ )
    {
        const i32 v = __builtin_amdgcn_ds_bpermute( ( ( lane - l ) + j0 / ( 2 * NR ) ) << 2, (i32)Hs[ k ] );
        hv = ( ( j0 >> 1 ) % NR ) == k ? v : hv;
    }
    const i32 hEn0 = ( j0 & 1 ) ? ( hv >> 16 ) : ( (i32)( (u32)hv << 16 ) >> 16 );
    const i32 span = en0 - st0, nS = ( span / 8 ) * 8;
    const i32 NONE = (i32)0x80000000;
    // the class of a cell is (t - st0) mod 8 = (rr - st0 - j) mod 8: lanes 4 / NR apart hold the same 2 NR classes
    i32 vh = hEn0, vt = NONE, tail = NONE; // (a class whose maximum does not beat H[en0] contributes en0 itself, below)
#pragma unroll
    for( int k = 0; k < NR; k++ )
    {
        const u32 ddpk = pk_sub( pk_bcast( rr - st0 ), Jpk[ k ] ); // t - st0 of the two cells (mod 2^16; dead cells: >= 0x8000 or > en0 - st0)
        const u32 inv = pk_sub( 0xffffffffu, ddpk ); // 0xffff - (t - st0): earlier chunks win ties
        const i32 lo = (i32)__builtin_amdgcn_perm( Hs[ k ], inv, 0x05040100u );
        const i32 hi = (i32)__builtin_amdgcn_perm( Hs[ k ], inv, 0x07060302u );
        const u32 dLo = ddpk & 0xffffu, dHi = ddpk >> 16;
        i32 kLo = dLo < (u32)nS ? lo : NONE, kHi = dHi < (u32)nS ? hi : NONE;
        if( NR == 2 )
        {
            kLo = max( kLo, dpp_ctrl<0x122>( kLo ) );
            kHi = max( kHi, dpp_ctrl<0x122>( kHi ) );
        }
        kLo = max( kLo, dpp_ctrl<0x124>( kLo ) );
        kHi = max( kHi, dpp_ctrl<0x124>( kHi ) );
        kLo = max( kLo, dpp_ctrl<0x128>( kLo ) );
        kHi = max( kHi, dpp_ctrl<0x128>( kHi ) );
        if( LANES >= 32 )
        {
            auto a16 = __builtin_amdgcn_permlane16_swap( (u32)kLo, (u32)kLo, false, false );
            kLo = max( (i32)a16[ 0 ], (i32)a16[ 1 ] );
            auto b16 = __builtin_amdgcn_permlane16_swap( (u32)kHi, (u32)kHi, false, false );
            kHi = max( (i32)b16[ 0 ], (i32)b16[ 1 ] );
        }
        if( LANES == 64 )
        {
            auto a32 = __builtin_amdgcn_permlane32_swap( (u32)kLo, (u32)kLo, false, false );
            kLo = max( (i32)a32[ 0 ], (i32)a32[ 1 ] );
            auto b32 = __builtin_amdgcn_permlane32_swap( (u32)kHi, (u32)kHi, false, false );
            kHi = max( (i32)b32[ 0 ], (i32)b32[ 1 ] );
        }
        const i32 hl = kLo >> 16, hh = kHi >> 16;
        const i32 tl = ( kLo != NONE && hl > hEn0 ) ? st0 + ( ( 0xffff - ( kLo & 0xffff ) ) & ~7 ) : en0;
        const i32 th = ( kHi != NONE && hh > hEn0 ) ? st0 + ( ( 0xffff - ( kHi & 0xffff ) ) & ~7 ) : en0;
        vh = max( vh, max( kLo != NONE ? hl : hEn0, kHi != NONE ? hh : hEn0 ) );
        vt = max( vt, max( tl, th ) );
        // the cells after the 8-lane part
        const i32 tLo = ( dLo >= (u32)nS && dLo < (u32)span ) ? lo : NONE, tHi = ( dHi >= (u32)nS && dHi < (u32)span ) ? hi : NONE;
        tail = max( tail, max( tLo, tHi ) );
    }
    // the 4 / NR lanes of a period hold the eight classes between them
    vh = max( vh, dpp_ctrl<0xB1>( vh ) );
    vt = max( vt, dpp_ctrl<0xB1>( vt ) );
    if( NR == 1 )
    {
        vh = max( vh, dpp_ctrl<0x4E>( vh ) );
        vt = max( vt, dpp_ctrl<0x4E>( vt ) );
    }
    mH = hEn0, mT = en0;
    if( nS > 0 )
        mH = vh, mT = vt;
    // the cells after the 8-lane part, in the order the reference visits them: the first of the largest wins, if it is larger
    const i32 tk = grp_max_i32<LANES>( tail );
    if( tk != NONE && ( tk >> 16 ) > mH )
    {
        mH = tk >> 16;
        mT = st0 + ( 0xffff - ( tk & 0xffff ) );
    }
}

// One set of up to G jobs (queue entries [at0, min(at0 + G, n))), all of them left- (LEFT) or right-aligned extensions.
// redo / nRedo: the hand-back list of the extension kernels.
template <int G, int NR, bool LEFT, typename FETCH>
__device__ void ksw_grp_set( const FETCH& F, const KswScoring& SC, const u32* list, u32 n, u32 at0, uint8_t* P /*KSW_GRP_ROWS x 128 NR B*/,
                             uint8_t* lds, const KswOut& O, KswWaveAcc& acc, u32* redo, unsigned int* nRedo, unsigned long long* sOff
#if defined( MA_KSW_PROF )
                             ,
                             unsigned long long* pf
#endif
)
{
    // NR packed registers (2 NR rows) per lane: a group of 64 / G lanes holds CJ = 128 NR / G query rows; a direction row of the
    // set is ROWB bytes, the LDS ring holds SR of them (4 KB whatever NR is)
    constexpr int LANES = 64 / G, CJ = 128 * NR / G, ROWB = 128 * NR, SR = KSW_GRP_STAGE_ROWS / NR;
    static_assert( NR == 1 || NR == 2, "two or four rows per lane" );
    const int lane = threadIdx.x & 63, g = lane / LANES, l = lane % LANES;
    GRP_PROF_T( tp0 );
    // Direction bytes: row r of the set (128 B: the G jobs' cells side by side) lives at ring slot r mod 32 in LDS.  Every 16
    // diagonals the 16 rows that the next 16 diagonals will overwrite are copied to the wave's HBM scratch (one coalesced 2 KB
    // copy), so at the end the ring holds the last 17..32 rows and HBM the older ones: the loop itself never stores to global
    // memory, and the back-trace starts on rows that never left the CU.
    uint8_t* stage = lds; // SR x ROWB = KSW_GRP_STAGE_ROWS x 128
    u32* cigLds = (u32*)( lds + KSW_GRP_STAGE_ROWS * 128 ); // KSW_GRP_CIG_WORDS, CIGCAP per group
    uint8_t* tring = lds + KSW_GRP_STAGE_ROWS * 128 + KSW_GRP_CIG_WORDS * 4; // G x 2 CJ bytes = 256 NR
    i32* gflag = (i32*)( tring + 256 * NR ); // per group: scratch words
    constexpr u32 CIGCAP = KSW_GRP_CIG_WORDS / G;
    // ---- the group's job
    const bool has = at0 + (u32)g < n;
    const u32 slot = list[ has ? at0 + g : at0 ];
    i32 qlen, tlen, rEnd, nDiag, zdrop;
    {
        const KswJobView J = F.view( slot );
        qlen = J.qlen, tlen = J.tlen, zdrop = J.zdrop;
        nDiag = qlen + tlen - 1;
        rEnd = min( nDiag, J.w + 1 ); // the job ends BEFORE diagonal rEnd: all diagonals done, or r > w (handed back)
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
    constexpr u32 tS = LEFT ? 4 : 0, tX = LEFT ? 3 : 1, tY = 2, tX2 = LEFT ? 1 : 3, tY2 = 0;
    // (wave-uniform constants: the scoring is the launch's)
    const u32 K_X0 = pk_val( -q - e, tX ), K_Y0 = pk_val( -q - e, tY ), K_X20 = pk_val( -q2 - e2, tX2 ), K_Y20 = pk_val( -q2 - e2, tY2 );
    const u32 K_TX = pk_val( 0, tX ), K_TY = pk_val( 0, tY ), K_TX2 = pk_val( 0, tX2 ), K_TY2 = pk_val( 0, tY2 );
    const u32 K_FX = pk_sub( K_TX, LEFT ? 0u : 0x00010001u ), K_FY = pk_sub( K_TY, LEFT ? 0u : 0x00010001u ),
              K_FX2 = pk_sub( K_TX2, LEFT ? 0u : 0x00010001u ), K_FY2 = pk_sub( K_TY2, LEFT ? 0u : 0x00010001u );
    const u32 K_NEG = 0x80008000u, K_MATCH = pk_bcast( sc_mch );
    const u32 V_CLIP = pk_val( sc_mch, 0xff );
    const u32 V_SCLO = ( (u32)sc_mch & 0xffu ) | ( ( (u32)sc_mis & 0xffu ) * 0x01010100u );
    const u32 V_SCHI = ( (u32)( -e2 ) & 0xffu ) | tS << 8;
    const u32 V_Q = pk_val( q, 0 ), V_Q2 = pk_val( q2, 0 ), V_QE = pk_val( q + e, 0 ), V_QE2 = pk_val( q2 + e2, 0 );
    const u32 M_LEADLO = l == 0 ? 0x0000ffffu : 0u; // the cell that takes the first-row boundary: row 0 of the group
    // ---- static per lane: rows j = 2 NR l + 2 k (low half of register k), + 1 (high half)
    u32 Jpk[ NR ], Jmask[ NR ], Qb[ NR ];
#pragma unroll
    for( int k = 0; k < NR; k++ )
    {
        const i32 j0 = 2 * NR * l + 2 * k, j1 = j0 + 1;
        Jpk[ k ] = (u32)j0 | (u32)j1 << 16;
        // rows of the job (cleared when it ends: a finished job's cells are dead)
        Jmask[ k ] = ( has && !untouched && j0 < qlen ? 0x0000ffffu : 0u ) | ( has && !untouched && j1 < qlen ? 0xffff0000u : 0u );
        Qb[ k ] = ( j0 < qlen ? (u32)qf( j0 ) & 0xffu : 4u ) | ( j1 < qlen ? (u32)qf( j1 ) & 0xffu : 4u ) << 16;
    }
    auto tgt2 = [ & ]( i32 t ) -> u32 { // target bases of cells t, t + 1 (codes as in ksw_ext.h: an N of the target is 12)
        if( t >= tlen )
            return 0u;
        u32 ab = tf.pair( t ) & ( t + 1 < tlen ? 0x00ff00ffu : 0x000000ffu );
        if( TF::CLEAN )
            return ab;
        const u32 nn = pk_lshr( ab, 2 );
        return pk_bfi( pk_sub( 0u, pk_minu( nn, 0x00010001u ) ), 0x000c000cu, ab );
    };
    // the group's target ring: bytes t mod 2 CJ; all of it now, CJ bytes more every CJ diagonals
    uint8_t* myRing = tring + g * ( 2 * CJ );
    auto fillRing = [ & ]( i32 tFrom ) { // CJ cells from tFrom on: every lane its 2 NR
#pragma unroll
        for( int k = 0; k < NR; k++ )
        {
            const i32 t = tFrom + 2 * NR * l + 2 * k;
            const u32 ab = has ? tgt2( t ) : 0u;
            *(uint16_t*)( myRing + ( t & ( 2 * CJ - 1 ) ) ) = (uint16_t)( ( ab & 0xffu ) | ( ab >> 8 & 0xff00u ) );
        }
    };
    fillRing( 0 );
    fillRing( CJ );
    u32 V[ NR ], X[ NR ], X2[ NR ], U[ NR ], Y[ NR ], Y2[ NR ], T[ NR ], H[ NR ], Tpk[ NR ], snapH[ NR ];
#pragma unroll
    for( int k = 0; k < NR; k++ )
    {
        const i32 j0 = 2 * NR * l + 2 * k, j1 = j0 + 1;
        V[ k ] = ( ( (u32)initOf( j0 ) & 0xffu ) << 8 ) | ( ( (u32)initOf( j1 ) & 0xffu ) << 24 ); // first-column v of the rows
        X[ k ] = K_X0, X2[ k ] = K_X20, U[ k ] = 0, Y[ k ] = K_Y0, Y2[ k ] = K_Y20, T[ k ] = 0;
        H[ k ] = ( (u32)hBoundary( j0 + 1 ) & 0xffffu ) | (u32)hBoundary( j1 + 1 ) << 16; // H(-1, j)
        Tpk[ k ] = pk_sub( 0u, Jpk[ k ] ); // t = r - j of the lane's cells
        snapH[ k ] = 0;
    }
    const u32 tlenpk = pk_bcast( tlen );
    // ---- per group, equal in all lanes of the group.  Flags are 0 / -1 words, not bools: a bool that lives across the loop is
    // an exec-style mask pair in SGPRs, and the scalar file is full of the kernel's arguments.
    i32 ezmax = 0, maxT = -1, maxQ = -1, pR = 0;
    u32 ezpk = 0;
    i32 act = has && !untouched ? -1 : 0, handBack = 0, pend = 0, zdropped = 0;
    i32 boundPrev = 0x7fffffff, nextBound = 0;
    const i32 boundRate = max( 1, ( -sc_mis + sc_mch + 1 ) / 2 );
    const i32 qe = q + e; // the cheaper gap model's opening cost (after the swap): a diagonal's maximum falls by at most this per diagonal
    i32 rLast = -1; // last diagonal the job computed
    // z-drop (kswcpp_core.h:22-44) needs a diagonal whose maximum lies more than zdrop below ez.max.  The maximum of diagonal
    // r + 1 is at least that of diagonal r minus (q + e): the best cell's right or lower neighbour can always open a gap from it
    // (the clip of z is an upper bound only).  So after a diagonal with maximum m the test cannot pass for
    // ( m - (ez.max - zdrop - 1) ) / (q + e) diagonals: a job's next test is scheduled like its next bound evaluation.
    i32 zNext = zdrop >= 0 ? 0 : 0x7fffffff;
    const i32 zStep = zdrop >= 0 ? ( zdrop + qe ) / qe : 0x3fffffff; // diagonals after a raise before the test can pass
    const u32 laneOff = (u32)( g * CJ + 2 * NR * l ); // the lane's 2 NR bytes of a direction row
    i32 uInS = initOf( 0 ); // first-row initialisation of cell t = r (wave-uniform: the scoring is)
    i32 ringLo = 0; // rows below it have been copied to HBM (wave-uniform)
    __syncthreads( );
    GRP_PROF_T( tp1 );
    i32 r = 0;
    for( ;; ++r )
    {
        // ---- a job ends before this diagonal: all diagonals done, or it leaves the regime and is handed back
        {
            const i32 ends = r >= rEnd ? act : 0;
            handBack |= rEnd < nDiag ? ends : 0;
            rLast = ends ? r - 1 : rLast;
            act &= ~ends;
#pragma unroll
            for( int k = 0; k < NR; k++ )
                Jmask[ k ] &= (u32)~ends;
        }
        if( !__any( act != 0 ) ) // (wave-uniform: no lane leaves the loop before the others)
            break;
        if( __builtin_expect( r <= long_thres + 1, 0 ) )
            uInS = initOf( r );
        if( __builtin_expect( r != 0 && ( r & ( CJ - 1 ) ) == 0, 0 ) )
        {
            fillRing( r + CJ );
            __syncthreads( );
        }
        if( __builtin_expect( r >= SR && ( r & ( SR / 2 - 1 ) ) == 0, 0 ) )
        {
            // rows [r - SR, r - SR / 2) -> HBM: their ring slots are the ones rows r .. r + SR / 2 - 1 take (2 KB: 32 B per lane)
            __syncthreads( );
            const uint4* src = (const uint4*)( stage + ( ( r - SR ) & ( SR - 1 ) ) * ROWB );
            uint4* dst = (uint4*)( P + (size_t)( r - SR ) * ROWB );
#pragma unroll
            for( int k = 0; k < KSW_GRP_STAGE_ROWS / 16; k++ )
                dst[ lane + 64 * k ] = src[ lane + 64 * k ];
            ringLo = r - SR / 2;
        }
        // ---- neighbours: u, y, y2 and the target base come from row j - 1: the previous lane's last register for the lane's first
        // row, the lane's own registers for the others
        const u32 tIn = (u32)myRing[ r & ( 2 * CJ - 1 ) ];
        u32 ut[ NR ], yt[ NR ], y2t[ NR ], tt[ NR ];
        ut[ 0 ] = cells_shift1( U[ 0 ], (u32)dpp_wave_shr1( (i32)U[ NR - 1 ] ) );
        yt[ 0 ] = cells_shift1( Y[ 0 ], (u32)dpp_wave_shr1( (i32)Y[ NR - 1 ] ) );
        y2t[ 0 ] = cells_shift1( Y2[ 0 ], (u32)dpp_wave_shr1( (i32)Y2[ NR - 1 ] ) );
        tt[ 0 ] = cells_shift1( T[ 0 ], (u32)dpp_wave_shr1( (i32)T[ NR - 1 ] ) );
        ut[ 0 ] = pk_bfi( M_LEADLO, ( (u32)uInS & 0xffu ) << 8, ut[ 0 ] );
        yt[ 0 ] = pk_bfi( M_LEADLO, K_Y0, yt[ 0 ] );
        y2t[ 0 ] = pk_bfi( M_LEADLO, K_Y20, y2t[ 0 ] );
        tt[ 0 ] = pk_bfi( M_LEADLO, tIn, tt[ 0 ] );
#pragma unroll
        for( int k = 1; k < NR; k++ )
        {
            ut[ k ] = cells_shift1( U[ k ], U[ k - 1 ] );
            yt[ k ] = cells_shift1( Y[ k ], Y[ k - 1 ] );
            y2t[ k ] = cells_shift1( Y2[ k ], Y2[ k - 1 ] );
            tt[ k ] = cells_shift1( T[ k ], T[ k - 1 ] );
        }
        u32 LM[ NR ], Hm[ NR ], dirs[ NR ];
#pragma unroll
        for( int k = 0; k < NR; k++ )
        {
            // ---- live cells: 0 <= t <= tlen - 1 on a row of the job
            LM[ k ] = pk_opaque( pk_nonzero15( pk_subsatu( tlenpk, Tpk[ k ] ) ) & Jmask[ k ] );
            // ---- score and DP cell (kswcpp_core.h:598-766; ksw_ext.h)
            const u32 sel = ( pk_minu( tt[ k ] ^ Qb[ k ], 0x00040004u ) << 8 ) | 0x00050005u;
            u32 z = __builtin_amdgcn_perm( V_SCHI, V_SCLO, sel );
            u32 a = pk_add( X[ k ], V[ k ] );
            u32 b = pk_add( yt[ k ], ut[ k ] );
            u32 a2 = pk_add( X2[ k ], V[ k ] );
            u32 b2 = pk_add( y2t[ k ], ut[ k ] );
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
            const u32 nu = pk_sub( zc, V[ k ] ), nv = pk_sub( zc, ut[ k ] );
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
            dirs[ k ] = __builtin_amdgcn_perm( 0u, d, 0x0c0c0200u ); // the two direction bytes side by side
            // ---- commit: the row's own v, x, x2 keep their first-column initialisation until the cell is born
            U[ k ] = nu;
            Y[ k ] = ny;
            Y2[ k ] = ny2;
            T[ k ] = tt[ k ];
            V[ k ] = pk_bfi( LM[ k ], nv, V[ k ] );
            X[ k ] = pk_bfi( LM[ k ], nx, X[ k ] );
            X2[ k ] = pk_bfi( LM[ k ], nx2, X2[ k ] );
            // ---- H(t, j) = H(t-1, j) + u(t, j)
            const u32 hn = pk_add( H[ k ], pk_ashr8( nu ) );
            H[ k ] = pk_bfi( LM[ k ], hn, H[ k ] );
            Hm[ k ] = pk_bfi( LM[ k ], hn, K_NEG );
            Tpk[ k ] = pk_add( Tpk[ k ], 0x00010001u );
            if( NR > 1 )
                __builtin_amdgcn_sched_barrier( 0 ); // one register set after the other: interleaved, their temporaries add up
        }
        // the lane's 2 NR bytes of row r of the ring
        if( NR == 1 )
        {
            if( LM[ 0 ] )
                *(uint16_t*)( stage + ( r & ( SR - 1 ) ) * ROWB + laneOff ) = (uint16_t)dirs[ 0 ];
        }
        else if( LM[ 0 ] | LM[ NR - 1 ] )
            *(u32*)( stage + ( r & ( SR - 1 ) ) * ROWB + laneOff ) = dirs[ 0 ] | dirs[ NR - 1 ] << 16;
        // the largest H of the lane's live cells
        u32 HmAll = Hm[ 0 ];
#pragma unroll
        for( int k = 1; k < NR; k++ )
            HmAll = pk_max( HmAll, Hm[ k ] );
        // ---- a larger maximum: the value now, its position when somebody asks (ksw_ext.h)
        i32 raise = 0;
        if( __any( pk_max( HmAll, ezpk ) != ezpk ) )
        {
            const i32 gm = grp_max_i32<LANES>( max( (i32)( HmAll << 16 ) >> 16, (i32)HmAll >> 16 ) );
            raise = gm > ezmax ? -1 : 0;
            ezmax = raise ? gm : ezmax;
            ezpk = raise ? pk_bcast( gm ) : ezpk;
#pragma unroll
            for( int k = 0; k < NR; k++ )
                snapH[ k ] = raise ? H[ k ] : snapH[ k ];
            pR = raise ? r : pR;
            pend |= raise;
            // (the diagonal's maximum IS ez.max now: no z-drop before it has fallen by zdrop + 1)
            zNext = raise ? r + zStep : zNext;
        }
        // ---- z-drop, on the job's schedule: rare
        if( __builtin_expect( __any( ( act & ~raise ) != 0 && r >= zNext ) != 0, 0 ) )
        {
            const i32 thr = ezmax - zdrop - 1; // candidates: every cell at or below it
            const i32 gm = grp_max_i32<LANES>( max( (i32)( HmAll << 16 ) >> 16, (i32)HmAll >> 16 ) );
            const bool mine = ( act & ~raise ) != 0 && r >= zNext;
            const bool cand = mine && gm <= thr;
            if( mine && !cand )
                zNext = r + ( gm - thr + qe - 1 ) / qe; // >= r + 1
            if( __any( cand ) )
            {
                i32 mH, mT;
                grp_exact_max<LANES, NR>( H, Jpk, r, qlen, tlen, lane, l, mH, mT );
                i32 pH, pT;
                grp_exact_max<LANES, NR>( snapH, Jpk, pR, qlen, tlen, lane, l, pH, pT );
                if( pend && cand )
                {
                    maxT = pT;
                    maxQ = pR - pT;
                    pend = 0;
                }
                if( cand && mT >= maxT && r - mT >= maxQ )
                {
                    const i32 tl = mT - maxT, ql = ( r - mT ) - maxQ;
                    const i32 dl = tl > ql ? tl - ql : ql - tl;
                    if( ezmax - mH > zdrop + dl * e2 )
                    {
                        zdropped = 1;
                        rLast = r;
                        act = 0;
#pragma unroll
                        for( int k = 0; k < NR; k++ )
                            Jmask[ k ] = 0;
                    }
                }
                if( cand && act )
                    zNext = r + 1; // every cell is below the threshold already: the test is due on every diagonal from here on
            }
        }
        // ---- early stop (ksw_reg.h; every job on the schedule of ksw_ext.h -- its OWN schedule: evaluating a job whenever a
        // neighbour is due would be as exact, but a job's executed cells would then depend on the jobs it shares a wave with)
        const bool due = ( act & ~raise ) != 0 && r >= qlen - 1 && r >= nextBound;
        if( __any( due ) )
        {
            u32 bm = K_NEG;
#pragma unroll
            for( int k = 0; k < NR; k++ )
            {
                const u32 QLpk = pk_sub( pk_bcast( qlen - 1 ), Jpk[ k ] ); // rows left below the cell
                const u32 pot = pk_min( QLpk, pk_sub( tlenpk, Tpk[ k ] ) ); // min( rows left, columns left ): Tpk is t + 1 by now
                const u32 bnd = pk_mad( pot, K_MATCH, H[ k ] );
                bm = pk_max( bm, pk_bfi( LM[ k ], bnd, K_NEG ) );
            }
            const i32 bound = grp_max_i32<LANES>( max( (i32)( bm << 16 ) >> 16, (i32)bm >> 16 ) );
            const i32 top = hBoundary( r ) + sc_mch * qlen;
            if( due )
            {
                const i32 all = max( max( bound, boundPrev ), top );
                if( r >= qlen && all <= ezmax )
                {
                    rLast = r;
                    act = 0;
#pragma unroll
                    for( int k = 0; k < NR; k++ )
                        Jmask[ k ] = 0;
                }
                else if( boundPrev != 0x7fffffff && r >= qlen )
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
    GRP_PROF_T( tp2 );
    // ---- position of the last raise
    {
        i32 pH, pT;
        grp_exact_max<LANES, NR>( snapH, Jpk, pR, qlen, tlen, lane, l, pH, pT );
        if( pend )
        {
            maxT = pT;
            maxQ = pR - pT;
        }
    }
    __syncthreads( ); // direction bytes (LDS, and the rows that went to HBM) visible to the back-trace
    // ---- back-trace (ksw_backtrack__, kswcpp_core.h:76-150; inside the regime the path stays in the rectangle): one lane per
    // job walks, all lanes stage the rows it is about to cross into LDS.  Cell (t, j) of diagonal r = t + j: byte g CJ + j of row r.
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
    // rows [winLo, ...] are in the ring (slot r mod SR): first the ones the loop left there, then SR-row windows out of HBM
    i32 winLo = ringLo;
    while( true )
    {
        while( bi >= 0 && bj >= 0 && bi + bj >= winLo )
        {
            const u32 tb = stage[ ( ( bi + bj ) & ( SR - 1 ) ) * ROWB + g * CJ + bj ];
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
        // the highest diagonal any walker stands on (below winLo: those rows went to HBM), 32 rows down
        const i32 rhi = wave_max_i32( walking ? bi + bj : -1 );
        winLo = max( 0, rhi - SR + 1 );
        __syncthreads( );
        for( i32 k = lane; k < ( rhi - winLo + 1 ) * ( ROWB / 16 ); k += 64 ) // 16 bytes each: ROWB / 16 per row
        {
            const i32 row = winLo + k / ( ROWB / 16 );
            *(uint4*)( stage + ( row & ( SR - 1 ) ) * ROWB + ( k % ( ROWB / 16 ) ) * 16 ) = *(const uint4*)( P + (size_t)row * ROWB + ( k % ( ROWB / 16 ) ) * 16 );
        }
        __syncthreads( );
    }
    if( leader )
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
    GRP_PROF_T( tp3 );
    // ---- publish: one pool reservation for the set, every group copies its cigar
    // (leader's values -> all lanes of the group)
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
    // cells the job computed: sum over r = 0 .. rLast of min( r, tlen - 1 ) - max( 0, r - qlen + 1 ) + 1
    u64 cellsJob = 0;
    if( has && rLast >= 0 )
    {
        const i64 R = rLast, a = tlen - 1, bq = qlen - 1;
        const i64 s1 = R <= a ? R * ( R + 1 ) / 2 : a * ( a + 1 ) / 2 + ( R - a ) * a;
        const i64 s2 = R <= bq ? 0 : ( R - bq ) * ( R - bq + 1 ) / 2;
        cellsJob = (u64)( s1 - s2 + R + 1 );
    }
    if( l == 0 && publish )
    {
        ma_ez rz;
        rz.max = untouched ? 0 : ( ezmax & 0x7fffffff );
        rz.zdropped = zdropped;
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
        redo[ atomicAdd( nRedo, 1u ) ] = slot;
    if( publish && fits )
        for( u32 i = (u32)l; i < myN; i += LANES ) // the walk leaves the cigar reversed (kswcpp_core.h:146-149)
            O.cig_pool[ off + i ] = revCigar ? myCig[ i ] : myCig[ myN - 1 - i ];
    // totals of the set -> the wave's accumulators (lane 0 flushes them)
    {
        u64 c = l == 0 && publish ? cellsJob : 0, p = l == 0 && publish ? (u64)steps : 0, nj = l == 0 && publish ? 1 : 0, cw = l == 0 && publish ? myN : 0;
#pragma unroll
        for( int k = 0; k < G; k++ )
        {
            acc.cells += ( (u64)(u32)__builtin_amdgcn_readlane( (i32)( c >> 32 ), k * LANES ) << 32 ) | (u32)__builtin_amdgcn_readlane( (i32)(u32)c, k * LANES );
            acc.path += (u32)__builtin_amdgcn_readlane( (i32)(u32)p, k * LANES );
            acc.njobs += (u32)__builtin_amdgcn_readlane( (i32)(u32)nj, k * LANES );
            acc.cig_words += (u32)__builtin_amdgcn_readlane( (i32)(u32)cw, k * LANES );
        }
    }
    __syncthreads( ); // LDS and the scratch rows are free for the next set
#if defined( MA_KSW_PROF )
    {
        const unsigned long long tp4 = clock64( );
        u32 own = l == 0 && has ? (u32)( rLast + 1 ) : 0u, nj = l == 0 && has ? 1u : 0u, ownAll = 0, njAll = 0;
        for( int k = 0; k < G; k++ )
        {
            ownAll += (u32)__builtin_amdgcn_readlane( (i32)own, k * LANES );
            njAll += (u32)__builtin_amdgcn_readlane( (i32)nj, k * LANES );
        }
        pf[ 0 ] += tp1 - tp0, pf[ 1 ] += tp2 - tp1, pf[ 2 ] += tp3 - tp2, pf[ 3 ] += tp4 - tp3; // (the wave's own totals: flushed at its end)
        pf[ 4 ] += (unsigned long long)r, pf[ 5 ] += 1ull, pf[ 6 ] += (unsigned long long)njAll, pf[ 7 ] += (unsigned long long)ownAll;
    }
#endif
}

// six job lists: G = 1 left / right (MA_KSW_GRP=2 only), G = 2 left / right, G = 4 left / right
#define KSW_GRP_LISTS 6
// One kernel per (G, direction): a single instantiation of ksw_grp_set per kernel keeps the register allocation of each below
// the budget (all six in one kernel: 128 VGPRs and scratch traffic inside the diagonal loop).
template <typename FETCH, int G, int NR, bool LEFT>
__global__ void __launch_bounds__( 64 ) __attribute__( ( amdgpu_waves_per_eu( NR == 1 ? 5 : 4, NR == 1 ? 5 : 4 ) ) )
k_ksw_grp( FETCH F, KswScoring SC, const u32* list, u32 n, unsigned int* next, uint8_t* scratch, u64 stride, KswOut O, u32* redo, unsigned int* nRedo )
{
    __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t lds[ KSW_GRP_LDS ];
    __shared__ u32 sSet;
    __shared__ unsigned long long sOff;
    uint8_t* P = scratch + (u64)blockIdx.x * stride;
    KswWaveAcc acc;
#if defined( MA_KSW_PROF )
    unsigned long long prof[ 8 ] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    u32 cur = 0, end = 0; // four sets per queue atomic (a set's time is mostly latency: one round trip less)
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
#if defined( MA_KSW_PROF )
        ksw_grp_set<G, NR, LEFT>( F, SC, list, n, cur, P, lds, O, acc, redo, nRedo, &sOff, prof );
#else
        ksw_grp_set<G, NR, LEFT>( F, SC, list, n, cur, P, lds, O, acc, redo, nRedo, &sOff );
#endif
        cur += G;
    }
    ksw_flush( O, acc, G == 2 ? 2 : 3 );
#if defined( MA_KSW_PROF )
    if( threadIdx.x == 0 )
        for( int i = 0; i < 8; i++ )
            atomicAdd( g_grp_prof + 8 * ( NR == 2 ? 0 : ( G == 2 ? 1 : 2 ) ) + i, prof[ i ] );
#endif
}
} // namespace ma
#endif
