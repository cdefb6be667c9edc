// stage_output.h -- the result records in the order and layout of the C ABI, packed on the device (k_aln_sizes, k_aln_pack).
// Textually part of pipeline.hip.
// ---- results in the order and layout of the C ABI, packed on the device so that a download is three plain copies:
// per read its alignments (NeedlemanWunsch order, or the MappingQuality selection), their ops as (type, length) pairs
__global__ void k_aln_sizes( u32 n_reads, const u64* hset_off, const AlnHeader* hdr, const u32* order, const u32* mq_cnt, int mq,
                             u64* cnt, u64* nops )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if( r >= n_reads )
        return;
    const u64 b = hset_off[ r ];
    const u32 c = mq ? mq_cnt[ r ] : (u32)( hset_off[ r + 1 ] - b );
    u64 o = 0;
    for( u32 k = 0; k < c; k++ )
        o += hdr[ b + order[ b + k ] ].n_ops;
    cnt[ r ] = c;
    nops[ r ] = o;
}
__global__ void k_aln_pack( u32 n_reads, const u64* hset_off, const AlnHeader* hdr, const u32* order, const u64* pool, int mq,
                            const u64* aln_off, const u64* ops_off, ma_alignment* alns, u64* ops )
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if( r >= n_reads )
        return;
    const u64 b = hset_off[ r ];
    const u32 c = (u32)( aln_off[ r + 1 ] - aln_off[ r ] );
    u64 po = ops_off[ r ];
    for( u32 k = 0; k < c; k++ )
    {
        const AlnHeader& h = hdr[ b + order[ b + k ] ];
        ma_alignment a;
        a.begin_ref = (i64)h.begin_ref;
        a.end_ref = (i64)h.end_ref;
        a.begin_q = (i64)h.begin_q;
        a.end_q = (i64)h.end_q;
        a.score = h.score;
        a.soc_index = h.soc_index;
        a.n_ops = h.n_ops;
        a.ops_off = po;
        a.secondary = mq ? h.secondary : 0;
        a.supplementary = mq ? h.supplementary : 0;
        a.mapq = mq ? h.mapq : 0.0;
        alns[ aln_off[ r ] + k ] = a;
        for( u32 j = 0; j < h.n_ops; j++ )
        {
            const u64 o = pool[ h.ops_off + j ];
            ops[ 2 * ( po + j ) ] = op_type( o );
            ops[ 2 * ( po + j ) + 1 ] = op_len( o );
        }
        po += h.n_ops;
    }
}
