/* ma_amd.h -- C ABI of the MI355X-native seed-and-extend engine (libma_amd.so).
 *
 * This is the drop-in boundary for MA's hot path.  Every entry point is plain C (pointers + sizes,
 * no C++/torch types) so the reference's host code (C++ via a thin wrapper, or Python via ctypes)
 * can bind it.  Each function cites the reference interface it replaces (paths relative to the
 * ITBE-Lab/ma source tree).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; ma_last_error() gives the message
 *     (the C++ module wrappers in ma_amd/host turn that into std::runtime_error, mirroring
 *     libs/ms/inc/ms/module/module.h:339-377).
 *   - nothing here falls back to a CPU path: without a HIP device every compute call fails.
 *   - all positions follow the reference: reference positions r in [0, 2F) on T.revcomp(T),
 *     query positions q in [0,|Q|), bases A0 C1 G2 T3, N = 4 (nucSeq.cpp:17-28).
 */
#ifndef MA_AMD_H
#define MA_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MA_AMD_ABI_VERSION 1

typedef struct ma_index ma_index; /* device-resident FMD-index + pack (FMIndex fMIndex.h:195-230, Pack pack.h:39-176) */
typedef struct ma_batch ma_batch; /* device-resident batch of reads + all stage outputs */

/* Parameters the path reads (parameter.h:521-1060); same fields/meaning as the reference's
 * Presetting/GlobalParameter members named in the comments of ma_params_default(). */
typedef struct
{
    int32_t seeding_technique; /* 0 maxSpan, 1 SMEMs, 2 MEMs    xSeedingTechnique  (binarySeeding.h:560-561) */
    int32_t min_seed_len; /* 16                                  xMinSeedLength */
    int32_t min_ambiguity; /* 0                                  xMinimalSeedAmbiguity */
    int32_t max_ambiguity; /* 100                                xMaximalSeedAmbiguity */
    int32_t min_seed_size_drop; /* 15                            xMinimalSeedSizeDrop */
    int32_t max_num_soc, min_num_soc; /* 30 / 1                  xMaxNumSoC / xMinNumSoC */
    int32_t harm_score_min; /* 18                                xHarmScoreMin */
    int32_t max_score_lookahead; /* 3                            xMaxScoreLookahead */
    int32_t switch_qlen; /* 800                                  xSwitchQlen */
    int32_t min_delta_dist; /* 16                                xMinDeltaDist */
    int32_t max_gap_area; /* 20                                  xMaxGapArea */
    int32_t padding; /* 1000                                     xPadding */
    int32_t bandwidth_ext; /* 512                                xBandwidthDPExtension */
    int32_t min_bandwidth_gap; /* 20                             xMinBandwidthGapFilling */
    int32_t zdrop; /* 200                                        xZDrop */
    int32_t sv_penalty; /* 100                                   pGlobalParams->uiSVPenalty */
    int32_t match, mismatch, gap, extend, gap2, extend2; /* 2 4 4 2 24 1 (pGlobalParams) */
    int32_t disable_heuristics; /* 0                             xDisableHeuristics */
    int32_t soc_width; /* 0                                      xSoCWidth */
    uint32_t srand_seed; /* RANSAC draws: glibc srand(seed) state at the start of each read's Harmonization */
    uint64_t genome_size_disable; /* 10 000 000                  xGenomeSizeDisable */
    double rel_min_seed_size_amount; /* 0.005                    xRelMinSeedSizeAmount */
    double harm_score_min_rel; /* 0.002                          xHarmScoreMinRel */
    double soc_score_decrease_tol; /* 0.1                        xSoCScoreDecreaseTolerance */
    double score_diff_tol; /* 0.0001                             xScoreDiffTolerance */
    double max_delta_dist; /* 0.1                                xMaxDeltaDist */
    int32_t min_alignment_score; /* 75                           xMinAlignmentScore */
    int32_t report_n_best; /* 0                                  xReportN */
    int32_t max_supplementary; /* 1                              xMaxSupplementaryPerPrim */
    double max_overlap_supplementary; /* 0.1                     xMaxOverlapSupplementary */
    /* read by the host modules SmallInversions / PairedReads */
    int32_t search_inversions; /* 0                              xSearchInversions */
    int32_t zdrop_inversion; /* 100                              xZDropInversion */
    int32_t use_paired_reads; /* 0                               xUsePairedReads */
    int32_t libm_probe; /* 0. Diagnostics only (tests): != 0 nudges every tan / sin / atan / log result of the chaining
                           stage by one ulp (ma_amd/csrc/chain.h LibmProbe); results must not change */
    double mean_paired_dist; /* 400                              xMeanPairedReadDistance */
    double std_paired_dist; /* 150                               xStdPairedReadDistance */
    double paired_bonus; /* 1.25                                 xPairedBonus */
} ma_params;

/* ParameterSetManager presets (parameter.h:1079-1104): "default" 1081, "illumina" 1083-1087, "illuminapaired" 1089-1094,
 * "pacbio" 1096-1098, "nanopore" 1101-1104. */
void ma_params_default( ma_params* p );
void ma_params_illumina( ma_params* p );
void ma_params_illuminapaired( ma_params* p ); /* illumina + xUsePairedReads (read by the host module PairedReads) */
void ma_params_pacbio( ma_params* p ); /* default + xMaxSupplementaryPerPrim 100, xMinNumSoC 5 */
void ma_params_nanopore( ma_params* p ); /* pacbio + SMEM seeding */
/* ParameterSetManager::setSelected (parameter.h:1163-1170) by key, case-insensitive like the reference's map keys are
 * lower case: 0 on success; an unknown key fails with the reference's text ("The presetting '<key>' can not be found."),
 * "sv-illumina" / "sv-pacbio" (1106-1128) fail as not implemented: they switch xRectangularSoc off
 * (stripOfConsideration.h:41-53), which the device path does not have. */
int ma_params_preset( const char* key, ma_params* p );

/* Records (same layout as the oracle's records so parity tests compare raw arrays) */
typedef struct
{
    int64_t q_start, q_size, sa_start, sa_start_rc, sa_size; /* Segment (segment.h:31-113): size = length-1 */
} ma_segment;
typedef struct
{
    int64_t q_start, len, r_start, delta; /* Seed (seed.h:34-46) */
    uint32_t ambiguity, on_forward;
} ma_seed;
typedef struct
{
    int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end, n_cigar; /* kswcpp_extz_t (kswcpp.h:31-41) */
} ma_ez;
typedef struct
{
    int64_t begin_ref, end_ref, begin_q, end_q, score; /* Alignment (alignment.h:55-84) */
    uint32_t soc_index, n_ops;
    uint64_t ops_off; /* first (type,len) pair of this alignment in the ops array */
    uint32_t secondary, supplementary;
    double mapq;
} ma_alignment;
typedef struct
{
    uint64_t acc_len; /* SoCOrder (soc.h:26-90): accumulated seed length = the score the queue is ordered by */
    uint32_t ambiguity, n_seeds; /* accumulated seed ambiguity (ties: more = smaller), seeds in the strip */
    uint32_t begin, end; /* its seeds: [begin, end) of the read's seeds re-sorted by reference position (soc.h:196-231) */
} ma_soc;
typedef struct
{
    int32_t qlen, tlen, w, zdrop, flag; /* kswcpp_dispatch arguments (kswcpp.h:165-190) */
    uint32_t reserved;
    uint64_t q_off, t_off; /* offsets into the query / target byte arrays handed to ma_ksw_batch */
} ma_ksw_job;

/* ---- runtime ---- */
const char* ma_last_error( void ); /* thread-local message of the last failing call */
int ma_abi_version( void );
int ma_device_count( int* n );
int ma_set_device( int device );

/* Page-locked host memory for the arrays handed to / filled by the batch calls: uploads and downloads of pageable memory go
 * through the runtime's staging buffers at a fraction of the PCIe rate.  (The reference has no counterpart: its containers
 * live in host memory only; this is what NucSeq / Alignment storage becomes on the way to and from the device.) */
int ma_host_alloc( uint64_t bytes, void** out );
int ma_host_free( void* p );
/* Pins the CALLING host thread (and the threads it starts afterwards) to the CPUs next to GPU `device` -- the local_cpulist of
 * its PCI function in sysfs.  On a two-socket node the scheduler is free to spread the threads that feed a device (launches,
 * size read-backs, stream waits, the copies out of and into page-locked memory) over both sockets, differently from run to run;
 * pinning takes that out of the run-to-run spread of the host-to-host rate (the effect itself is within it).  mode 0: the GPU's CPUs; 1: the OTHER CPUs (the experiment that shows the
 * cost); -1: the mask the thread had when it first called this function.  Every mode stays INSIDE that first mask (a taskset /
 * numactl / SLURM affinity of the caller is never widened).  *n_cpus (optional) = CPUs of the new mask, 0 when the mask was left
 * alone (no topology information, a one-node host, none of the wanted CPUs in the caller's mask).  Fails only for a bad argument.  (No counterpart in the reference: its worker threads are
 * placed by the OS, module.h:303-369.) */
int ma_host_bind_thread( int device, int mode, int* n_cpus );

/* ---- index: replaces FMIndex(std::string) / Pack(std::string) loading (fMIndex.h:952-955, pack.h:513-525) ---- */
/* Upload an index whose arrays were read from the reference's own .bwt/.sa/.pac files. */
int ma_index_create( const uint32_t* bwt_words, uint64_t n_words, const int64_t* sa, uint64_t n_sa,
                     const uint64_t L2[ 5 ], int64_t primary, uint64_t ref_len_fwd_rev, const uint8_t* pac,
                     int32_t n_contigs, const uint64_t* contig_starts, const uint64_t* contig_lens, ma_index** out );
/* Build pack + FMD-index on the GPU from N-free contigs (host codes 0..3); replaces
 * FMIndex::build_FMIndex (fMIndex.cpp:316-391) + Pack::vAppendSequence (pack.h:586-698). Produces
 * byte-identical .bwt/.sa/.pac content. */
int ma_index_build( int32_t n_contigs, const uint64_t* contig_lens, const uint8_t* codes_concat, ma_index** out );
/* Same, for a genome that already lives in device memory as 1 byte/base codes. */
int ma_index_build_device( int32_t n_contigs, const uint64_t* contig_lens, const void* d_codes, ma_index** out );
int ma_index_destroy( ma_index* );
int ma_index_device( const ma_index*, int* device ); /* the GPU the index lives on (every call on it binds to that device) */
int ma_index_sizes( const ma_index*, uint64_t* n_words, uint64_t* n_sa, uint64_t* ref_len, int32_t* n_contigs );
/* Download (for FMIndex::vStoreFMIndex-compatible files and for tests); any pointer may be NULL. */
int ma_index_download( const ma_index*, uint32_t* bwt_words, int64_t* sa, uint64_t L2[ 5 ], int64_t* primary,
                       uint8_t* pac, uint64_t* contig_starts, uint64_t* contig_lens );

/* Pack::vExtract (pack.h:1440-1450, vExtractSubsection 1147-1236) for n ranges [begin[i], end[i]) of the doubled
 * text; codes 0..3 back to back in out (host). Replaces the reference extraction SmallInversions does
 * (smallInversions.h:207). */
int ma_pack_extract( const ma_index*, const uint64_t* begin, const uint64_t* end, uint64_t n, uint8_t* out );

/* ---- primitive ops (tests / SuffixArrayInterface seam, fMIndex.h:155-175) ---- */
/* n independent FMIndex::extend_backward calls (fMIndex.cpp:21-101): ik[3n] -> ok[3n] (host arrays) */
int ma_extend_backward_batch( const ma_index*, const int64_t* ik, const uint8_t* c, uint64_t n, int64_t* ok );
/* n independent FMIndex::bwt_sa calls (fMIndex.h:788-814) */
int ma_bwt_sa_batch( const ma_index*, const int64_t* rows, uint64_t n, int64_t* pos );
/* n independent kswcpp_dispatch calls (kswcpp.h:165-190). ez[n]; the cigar of job j is the ez[j].n_cigar words at
 * cigar[cigar_off[j]] (the cigars are packed densely but NOT in job order); cigar_off[n] = words used in total;
 * returns an error if cigar_cap is too small. */
int ma_ksw_batch( const ma_params*, const ma_ksw_job* jobs, uint64_t n, const uint8_t* q_bytes, uint64_t q_len,
                  const uint8_t* t_bytes, uint64_t t_len, ma_ez* ez, uint64_t* cigar_off, uint32_t* cigar,
                  uint64_t cigar_cap );

/* Same call with the semantics the pipeline uses for its extension / small-gap DP (NeedlemanWunsch::dynPrg,
 * ksw_dual_ext, ksw; needlemanWunsch.cpp:82-622 read only these): ez.max, ez.max_q, ez.max_t, n_cigar and the
 * cigar are those of kswcpp_dispatch, the other ez fields are unspecified.  Runs the packed extension kernel with
 * the exact early stop where a job qualifies (ma_amd/csrc/ksw_ext.h) and the exact kernels otherwise. */
int ma_ksw_ext_batch( const ma_params*, const ma_ksw_job* jobs, uint64_t n, const uint8_t* q_bytes, uint64_t q_len,
                      const uint8_t* t_bytes, uint64_t t_len, ma_ez* ez, uint64_t* cigar_off, uint32_t* cigar,
                      uint64_t cigar_cap );

/* ---- batch pipeline: BinarySeeding -> StripOfConsideration -> Harmonization -> NeedlemanWunsch -> MappingQuality
 *      as wired in libMA::setUpCompGraph (libs/ma/src/util/export.cpp:104-108) ---- */
int ma_batch_create( const ma_index*, const ma_params*, uint64_t max_reads, uint64_t max_bases, ma_batch** out );
int ma_batch_destroy( ma_batch* );
/* stream: a hipStream_t (as void*) all stage kernels of this batch are launched on; NULL = default stream */
int ma_batch_set_stream( ma_batch*, void* hip_stream );
/* waits of this batch's calls: 0 = the runtime's spinning stream synchronisation (lowest latency, default), 1 = sleep on
 * an interrupt-driven event (hosts running far more threads than cores: the DeviceBatcher of the host layer) */
int ma_batch_set_blocking_sync( ma_batch*, int on );
/* reads as 1 byte/base codes, CSR offsets[n+1]; host or device pointers */
int ma_batch_set_reads( ma_batch*, const uint8_t* codes, const uint64_t* offsets, uint64_t n_reads );
int ma_batch_set_reads_device( ma_batch*, const void* d_codes, const void* d_offsets, uint64_t n_reads,
                               uint64_t n_bases );
/* stages (asynchronous on the batch stream; each requires the previous one) */
int ma_seed_batch( ma_batch* ); /* BinarySeeding::execute (binarySeeding.cpp:86-178) */
int ma_extract_seeds_batch( ma_batch* ); /* ExtractSeeds::execute (stripOfConsideration.h:138-157) */
int ma_chain_batch( ma_batch* ); /* StripOfConsiderationSeeds::execute + Harmonization::execute */
int ma_dp_batch( ma_batch* ); /* NeedlemanWunsch::execute + MappingQuality::execute */
int ma_align_batch( ma_batch* ); /* all four */
int ma_batch_sync( ma_batch* ); /* wait for the stream; surfaces asynchronous kernel errors / capacity overflows */

/* streams for callers that keep several batches in flight without including HIP headers (host layer: DeviceBatcher,
 * MultiDeviceAligner); created on the index's device, non-blocking with respect to the default stream */
int ma_stream_create( const ma_index*, void** out_hip_stream );
int ma_stream_destroy( const ma_index*, void* hip_stream );

/* stage INPUTS from the host: a module of this library takes over in the middle of a chain whose earlier stages ran
 * elsewhere (e.g. the reference's BinarySeeding followed by the MI355X StripOfConsideration).  Reads must be set; each
 * call replaces the output of the preceding stage (CSR offsets per read, records as the get functions return them). */
int ma_batch_set_segments( ma_batch*, const uint64_t* seg_off /*n+1*/, const ma_segment* segs ); /* -> ma_extract_seeds_batch */
int ma_batch_set_seeds( ma_batch*, const uint64_t* seed_off /*n+1*/, const ma_seed* seeds ); /* -> ma_chain_batch */
int ma_batch_set_hsets( ma_batch*, const uint64_t* hset_off /*n+1*/, const uint64_t* hseed_off /*n_hsets+1*/,
                        const uint32_t* hset_soc, const ma_seed* hseeds ); /* -> ma_dp_batch */
/* A SoC queue per read that was swept elsewhere (SoCPriorityQueue soc.h:96-420 as StripOfConsideration::execute returns
 * it, stripOfConsideration.cpp:162-173): sorted_seeds = its pSeeds (re-sorted by reference position), socs = its vMaxima
 * array (layout of ma_batch_get_soc_heap).  -> ma_chain_batch, which then runs Harmonization::execute only. */
int ma_batch_set_soc_heap( ma_batch*, const uint64_t* soc_off /*n+1*/, const ma_soc* socs, const uint64_t* seed_off /*n+1*/,
                           const ma_seed* sorted_seeds );
/* MappingQuality::execute (mappingQuality.cpp:11-131) alone: alignments that were computed elsewhere (e.g. by the
 * reference's NeedlemanWunsch), per read in the order NeedlemanWunsch::execute left them (needlemanWunsch.h:131-132), in
 * the layout ma_batch_get_alignments returns.  Runs the MappingQuality kernel; ma_batch_get_mapq_alignments then serves
 * its selection (flags, mapping quality, order), ma_batch_get_alignments the input. */
int ma_batch_set_alignments( ma_batch*, const uint64_t* aln_off /*n+1*/, const ma_alignment* alns, const uint64_t* ops );
/* The SoC queue of every read (StripOfConsiderationSeeds::execute stripOfConsideration.cpp:12-161; requires extracted
 * seeds): socs[soc_off[r] .. soc_off[r+1]) are read r's strips in pop() order (SoCPriorityQueue::pop soc.h:240-284, i.e.
 * index_of_strip = position), their seed ranges refer to sorted_seeds[seed_off[r] ..), the read's seeds re-sorted by
 * reference position as rectangularSoC leaves them.  n_socs first (other pointers NULL), then the arrays. */
int ma_batch_get_socs( ma_batch*, uint64_t* n_socs, uint64_t* soc_off /*n+1*/, ma_soc* socs, uint64_t* seed_off /*n+1*/,
                       ma_seed* sorted_seeds );
/* Same arguments, but socs[] is the queue's internal array `vMaxima` (soc.h:140) as StripOfConsiderationSeeds::execute
 * leaves it -- std::make_heap, then rectangularSoC() without re-heapifying (stripOfConsideration.cpp:152-156) -- instead of
 * the pop order.  A binding on the reference's own SoCPriorityQueue fills pSeeds / vMaxima with it and the reference's
 * own pop() (soc.h:240-284) then yields the reference's order. */
int ma_batch_get_soc_heap( ma_batch*, uint64_t* n_socs, uint64_t* soc_off /*n+1*/, ma_soc* socs, uint64_t* seed_off /*n+1*/,
                           ma_seed* sorted_seeds );

/* results: counts first, then download into caller-allocated arrays (any pointer may be NULL) */
int ma_batch_counts( ma_batch*, uint64_t* n_segments, uint64_t* n_seeds, uint64_t* n_hsets, uint64_t* n_hseeds,
                     uint64_t* n_alignments, uint64_t* n_ops, uint64_t* n_aligned_reads );
int ma_batch_get_segments( ma_batch*, uint64_t* seg_off /*n+1*/, ma_segment* segs );
int ma_batch_get_seeds( ma_batch*, uint64_t* seed_off /*n+1*/, ma_seed* seeds );
int ma_batch_get_hsets( ma_batch*, uint64_t* hset_off /*n+1*/, uint64_t* hseed_off /*n_hsets+1*/, uint32_t* hset_soc,
                        ma_seed* hseeds );
int ma_batch_get_alignments( ma_batch*, uint64_t* aln_off /*n+1*/, ma_alignment* alns, uint64_t* ops /*2*n_ops*/ );
int ma_batch_get_mapq_alignments( ma_batch*, uint64_t* aln_off /*n+1*/, ma_alignment* alns, uint64_t* ops );

/* ---- double-buffered I/O of a batch object (the throughput form of the host-to-host path) ----
 * ma_batch_set_reads / ma_batch_get_mapq_alignments put the upload before and the download after a batch's kernels: with B
 * batch objects in flight every object spends that time with nothing of its own on the device.  These four calls move both
 * onto an I/O stream of the batch object, beside its kernels:
 *   ma_batch_stage_reads         starts the upload of the NEXT reads into the object's second read buffer and returns; the kernels
 *                                of the current reads may be running.  `codes` / `offsets` must stay untouched until
 *   ma_batch_use_staged_reads    has returned: it waits for that upload and makes the staged reads the object's reads (what
 *                                ma_batch_set_reads does in one step).
 *   ma_batch_start_mapq_download after ma_align_batch + ma_batch_sync: packs the MappingQuality records (the arrays of
 *                                ma_batch_get_mapq_alignments, same capacities) on the batch's stream and starts their download
 *                                on the I/O stream; returns at once, the next reads may be aligned on the same object meanwhile.
 *   ma_batch_finish_download     waits for it (no-op when none is pending); the host arrays are complete after it.
 * One staged upload and one download can be pending per object.  (No counterpart in the reference, whose containers never
 * leave host memory; the closest is its reader module pulling the next reads while the graph works, module.h:303-369.) */
int ma_batch_stage_reads( ma_batch*, const uint8_t* codes, const uint64_t* offsets, uint64_t n_reads );
int ma_batch_use_staged_reads( ma_batch* );
int ma_batch_start_mapq_download( ma_batch*, uint64_t* aln_off /*n+1*/, ma_alignment* alns, uint64_t* ops );
int ma_batch_finish_download( ma_batch* );
/* work counters for the roofline model (same meaning as the oracle's): [0] extend_backward steps,
 * [1] distinct occ blocks touched, [2] bwt_sa LF steps, [3] SA rows, [4] DP band cells, [5] ksw jobs */
int ma_batch_counters( ma_batch*, uint64_t out[ 8 ] );
/* diagnostics: the kswcpp calls of the last ma_dp_batch, 8 x int32 per call:
 * qlen, tlen, w, zdrop, flag, ez.zdropped, ez.max_q, ez.max_t (shapes may be NULL to only count) */
int ma_batch_get_dp_jobs( ma_batch*, uint64_t* n_jobs, int32_t* shapes, uint64_t cap );
/* per-kernel HIP-event times of the last stage calls in ms: [0] seeding [1] sa-lookup [2] chaining
 * [3] dp-jobs [4] ksw [5] stitch; requires ma_batch_enable_timing(b,1) */
int ma_batch_enable_timing( ma_batch*, int on );
int ma_batch_kernel_ms( ma_batch*, float out[ 8 ] );
/* host wall time in ms of the stage calls of the last ma_align_batch: [0] seed [1] extract [2] chain [3] dp (launches,
 * stream waits and size read-backs included: what a small batch pays beside its kernels) */
int ma_batch_host_ms( ma_batch*, float out[ 8 ] );
/* diagnostics: extension jobs tried on the proven narrow band (ma_amd/csrc/ksw_band.h; MA_KSW_GRP=3) since the library was loaded on
 * the current device: tried, proved, failed check 1 / 2 / 3 / 4, handed back for another reason, diagonals run */
int ma_debug_band_stats( unsigned long long out[ 8 ] );
/* the same for the LONG extension jobs (queries beyond 254 bases: the end extensions of long reads, needlemanWunsch.cpp:708-716,
 * 781-782), one per wavefront on the proven band of 120 cells (ksw_band.h, G = 1; MA_KSW_BANDL=0 switches it off) */
int ma_debug_band_long_stats( unsigned long long out[ 8 ] );
/* diagnostics: kswcpp cells computed / calls answered per DP kernel family since the library was loaded (what bench.py divides a
 * family's instruction count of the rocprofv3 PMC pass by): out[2f], out[2f+1] for f = 0 k_ksw_ext<1>, 1 k_ksw_ext<2>,
 * 2 k_ksw_grp<2>, 3 k_ksw_grp<4>, 4 k_ksw_band (short jobs), 5 k_ksw_pk, 6 k_ksw (state in LDS), 7 k_ksw_band (long jobs); all of
 * them restate kswcpp_core.h:308-879 */
int ma_debug_dp_family_stats( unsigned long long out[ 16 ] );
/* diagnostics: the device libm the chaining stage decides with (harmonization.h:82-89, ransac.cpp:112,131-135 use
 * glibc's): op 0 tan, 1 sin, 2 atan, 3 log over n doubles (host arrays); tests compare the bits with glibc's */
int ma_debug_libm( int op, const double* in, uint64_t n, double* out );

/* ---- synthetic workloads (BASELINE.json configs; deterministic counter-based generators) ---- */
/* genome: d_codes[total] (device, 1 byte/base). reads sampled from it: device CSR. */
int ma_synth_genome_device( uint64_t seed, uint64_t total_len, int32_t with_repeats, void* d_codes );
int ma_synth_reads_device( const ma_index*, uint64_t seed, uint64_t n_reads, uint32_t read_len, double sub_rate,
                           double ins_rate, double del_rate, uint64_t first_read_index, void* d_codes,
                           void* d_offsets, uint64_t codes_cap, uint64_t* n_bases );

#ifdef __cplusplus
}
#endif
#endif /* MA_AMD_H */
