/*
 * v2m_hip.h -- C ABI of the MI355X (gfx950) haplotype-splice path for vcf2multialign.
 *
 * Drop-in boundary for ONE hot path of tsnorri/vcf2multialign: the per-row
 * reference+ALT splice (output_sequence) batched over the rows of one A2M file, and the
 * bit-packed path-matrix transpose.  The C++ host keeps parsing the VCF, building the
 * variant_graph and writing the A2M file; everything between "graph built" and "row
 * bytes ready" happens on the GPU behind these entry points.
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   v2m_transpose_bits[_device]   transpose_matrix()            include/vcf2multialign/transpose_matrix.hh:14
 *                                                                libvcf2multialign/transpose_matrix.cc:41-109
 *                                                                (call site libvcf2multialign/variant_graph.cc:453)
 *   v2m_upload_graph              the read-only view of          include/vcf2multialign/variant_graph.hh:57-66
 *                                 variant_graph + ref_seq that
 *                                 output_sequence() walks
 *   v2m_splice_rows[_device]      output_sequence() called once  include/vcf2multialign/sequence_writer.hh:49-56
 *                                 per row by                     libvcf2multialign/sequence_writer.cc:22-85
 *                                 haplotype_output::output_a2m   libvcf2multialign/haplotype_output.cc:38-82
 *                                 and founder_sequence_greedy_   libvcf2multialign/founder_sequence_greedy_output.cc:515-550
 *                                 output::output_a2m
 *   v2m_row_batch                 sequence_writing_delegate      include/vcf2multialign/sequence_writer.hh:16-36
 *                                 (chromosome_copy_index, and    libvcf2multialign/haplotype_output.cc:22-32
 *                                 the founder delegate's copy    libvcf2multialign/founder_sequence_greedy_output.cc:78-115
 *                                 switch at cut nodes)
 *   v2m_upload_path_slice /       transpose_matrix() at its one  libvcf2multialign/variant_graph.cc:453
 *   v2m_upload_path_blocks /      call site, for one GPU's       include/vcf2multialign/variant_graph.hh:62-63
 *   v2m_bind_path_matrix_device   chromosome copies (or all)
 *   v2m_pbwt_cut_trials[_streamed] the edge-by-edge part of      include/vcf2multialign/pbwt.hh:77-134
 *   v2m_pbwt_cut_records          find_cut_positions() and       libvcf2multialign/find_cut_positions.cc:126-176
 *                                 find_matchings() (optional)    libvcf2multialign/founder_sequence_greedy_output.cc:208-251
 *
 * V2M_ABI_VERSION: 1 = round 1 (transpose, graph, rows); 2 = + v2m_upload_path_slice, v2m_alloc_output / v2m_free_output,
 * v2m_profile_get_launches (v2m_bind_path_matrix_device followed without a step); 3 = + v2m_upload_path_blocks (and then
 * v2m_pbwt_cut_trials, v2m_pbwt_cut_records); 4 = + v2m_pbwt_cut_trials_streamed; 5 = + v2m_splice_rows_held / v2m_row_release (rows a
 * sink may keep until it says so).  Entries have only ever been added.
 *
 * Conventions
 *   - Plain C: pointers + sizes, no exceptions, no C++/torch types.  Every function that can
 *     fail returns a V2M_* status; v2m_last_error() gives the message.
 *   - All graph integers are uint64_t exactly as in the reference (variant_graph.hh:38-40);
 *     the library narrows to 32 bits on upload and rejects graphs that do not fit
 *     (V2M_ERR_UNSUPPORTED).
 *   - Bit matrices: column-major, one column = n_rows/64 consecutive uint64_t words, row r of a
 *     column in word r/64 at bit r%64 (LSB first).  Dimensions are multiples of 64.
 *   - One ctx per GPU, driven by one host thread.  The graph is uploaded once and reused.
 *   - There is NO CPU fallback: without a usable HIP device v2m_ctx_create() fails.
 */
#ifndef V2M_HIP_H
#define V2M_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V2M_ABI_VERSION 5

enum {
	V2M_OK = 0,
	V2M_ERR_INVALID_ARGUMENT = 1, /* null pointer, zero-sized where not allowed, bad flag */
	V2M_ERR_PRECONDITION = 2,     /* what the reference asserts: dims % 64, graph invariants, cut nodes */
	V2M_ERR_UNSUPPORTED = 3,      /* valid input outside what this build handles (e.g. > 32-bit positions) */
	V2M_ERR_NO_DEVICE = 4,        /* no HIP device / wrong architecture */
	V2M_ERR_HIP = 5,              /* a HIP runtime call failed */
	V2M_ERR_OUT_OF_MEMORY = 6,
	V2M_ERR_SINK = 7,             /* the sink callback returned non-zero */
	V2M_ERR_STATE = 8             /* call order: no graph uploaded, no path matrix set, ... */
};

/* sequence_writing_delegate::PLOIDY_MAX (sequence_writer.hh:23-25): "follow REF edges only". */
#define V2M_PLOIDY_MAX UINT32_MAX

/* flags of v2m_splice_rows* */
#define V2M_SPLICE_UNALIGNED 0x1u /* should_output_unaligned (sequence_writer.cc:80): no '-' padding.  Bytes are opaque to the
                                   * walk (sequence_writer.cc:73-74 streams whatever the FASTA / VCF held) and aligned mode keeps
                                   * every byte value; the unaligned kernels use byte 0 as their padding marker, so a graph whose
                                   * ref_seq or label pool holds a NUL byte is refused in this mode (V2M_ERR_UNSUPPORTED) rather
                                   * than written without it. */

typedef struct v2m_ctx v2m_ctx;

/* ---- context ---------------------------------------------------------------------------- */

/* Binds to HIP device `device_id`, creates the library's stream.  Fails with V2M_ERR_NO_DEVICE
 * when there is no usable gfx950 device. */
int v2m_ctx_create(int device_id, v2m_ctx **ctx_out);
void v2m_ctx_destroy(v2m_ctx *ctx);

/* Message of the last failure on this ctx (or, with ctx == NULL, of the last failed
 * v2m_ctx_create on this thread).  Never NULL. */
const char *v2m_last_error(const v2m_ctx *ctx);

/* Blocks until everything queued on the ctx's stream has finished. */
int v2m_ctx_synchronize(v2m_ctx *ctx);

/* The hipStream_t all kernels of this ctx are launched on (for event timing by the caller). */
void *v2m_ctx_stream(v2m_ctx *ctx);

/* Human-readable note on what the context has auto-tuned so far (e.g. the store flavour of the aligned
 * splice, calibrated on the first >= 1 GiB launch).  Never NULL; empty before any tuning. */
const char *v2m_ctx_info(const v2m_ctx *ctx);

uint32_t v2m_abi_version(void);

/* ---- transpose_matrix ------------------------------------------------------------------- */

/* dst(c, r) = src(r, c).  src has n_rows x n_cols bits, dst n_cols x n_rows; both column-major
 * uint64 words as above; dst holds n_rows*n_cols/64 words and is fully overwritten.
 * n_cols == 0 is a no-op returning V2M_OK (transpose_matrix.cc:48-49); dimensions that are
 * not multiples of 64 return V2M_ERR_PRECONDITION (asserted at transpose_matrix.cc:53-54).
 * Host-pointer form: copies in, transposes on the GPU, copies out, synchronous. */
int v2m_transpose_bits(v2m_ctx *ctx, const uint64_t *src_words, uint64_t n_rows, uint64_t n_cols, uint64_t *dst_words);

/* Device-pointer form: both buffers already in HBM (8-byte aligned, non-overlapping);
 * asynchronous on the ctx's stream. */
int v2m_transpose_bits_device(v2m_ctx *ctx, const void *d_src_words, uint64_t n_rows, uint64_t n_cols, void *d_dst_words);

/* ---- variant_graph view ----------------------------------------------------------------- */

/* Raw view of the variant_graph fields output_sequence() reads (variant_graph.hh:57-63),
 * with alt_edge_labels flattened to CSR.  All pointers are host pointers owned by the caller
 * and only read during v2m_upload_graph(). */
typedef struct v2m_graph_view {
	uint64_t node_count;                 /* reference_positions.size() (>= 1; last node = sink) */
	uint64_t edge_count;                 /* alt_edge_targets.size() */
	const uint64_t *reference_positions; /* [node_count]  strictly increasing                   */
	const uint64_t *aligned_positions;   /* [node_count]  MSA co-ordinates                      */
	const uint64_t *alt_edge_targets;    /* [edge_count]  node ids                              */
	const uint64_t *alt_edge_count_csum; /* [node_count + 1]; edges of node n = [csum[n], csum[n+1]) */
	const uint64_t *alt_edge_label_offsets; /* [edge_count + 1] offsets into alt_edge_label_bytes */
	const char *alt_edge_label_bytes;
	/* paths_by_chrom_copy_and_edge: rows = edges (padded to path_rows), cols = chromosome copies
	 * (padded to path_cols).  May be NULL when the matrix is supplied later with
	 * v2m_set_paths_device(). */
	const uint64_t *paths_by_chrom_copy_and_edge;
	uint64_t path_rows; /* >= edge_count, multiple of 64 (0 allowed when edge_count == 0) */
	uint64_t path_cols; /* multiple of 64 */
} v2m_graph_view;

/* Validates the invariants output_sequence() relies on (V2M_ERR_PRECONDITION otherwise),
 * narrows to 32 bits, uploads graph + reference, and precomputes the device-side tables
 * (edge source nodes, per-edge aligned spans, the gap-aligned REF row, per-tile edge ranges).
 * Replaces any previously uploaded graph.  Synchronous. */
int v2m_upload_graph(v2m_ctx *ctx, const v2m_graph_view *graph, const char *ref_seq, uint64_t ref_len);

/* Uses a path matrix that already lives in HBM (e.g. the output of v2m_transpose_bits_device)
 * as paths_by_chrom_copy_and_edge of the uploaded graph.  The buffer is BORROWED: it must stay
 * valid and unchanged until the next upload/set or ctx destruction. */
int v2m_set_paths_device(v2m_ctx *ctx, const void *d_words, uint64_t path_rows, uint64_t path_cols);

/* One GPU's share of the path matrix when chromosome copies are sharded over several GPUs (rows are independent,
 * haplotype_output.cc:62-81; the graph and reference are replicated with v2m_upload_graph, the matrix is not).
 * `paths_by_edge_and_chrom_copy` is the WHOLE host-resident transpose input (variant_graph.hh:62: n_rows = chromosome
 * copies, n_cols = ALT edges, both multiples of 64, column-major as above) as build_variant_graph leaves it just before
 * its transpose_matrix call (variant_graph.cc:453).  Only the bits of copies [first_copy, first_copy + n_copies) are
 * moved to this GPU -- bytes [first_copy / 8, ..) of every column, a strided 2-D copy; first_copy must be a multiple
 * of 8 -- zero-padded to a multiple of 64 copies, transposed THERE (the same kernels as v2m_transpose_bits_device) and
 * bound as paths_by_chrom_copy_and_edge of the uploaded graph: n_cols rows (edges) x 64 * ceil(n_copies / 64) columns,
 * column j = copy first_copy + j.  Row batches for this ctx then use copy indices relative to first_copy.
 * first_copy = 0, n_copies = n_rows is the unsharded case: upload + transpose + bind without the matrix ever coming
 * back to the host.  The result is owned by the ctx.  Synchronous. */
int v2m_upload_path_slice(v2m_ctx *ctx, const uint64_t *paths_by_edge_and_chrom_copy, uint64_t n_rows, uint64_t n_cols, uint64_t first_copy, uint64_t n_copies);

/* The same for a GPU whose chromosome copies are not one contiguous range but every stride_copies-th block of block_copies
 * copies: copies first_copy + j * stride_copies + [0, block_copies) for j = 0, 1, ..., clipped to copy_end (first_copy,
 * block_copies and stride_copies multiples of 8).  Column l of the bound matrix is copy
 * first_copy + (l / block_copies) * stride_copies + l % block_copies, and row batches for this ctx use those local indices.
 * This is how several GPUs share an output that has to leave in row order (a pipe, an unaligned A2M file,
 * haplotype_output.cc:62-81 with sequence_writer.cc:80): with blocks dealt round-robin, GPU k produces rows
 * k * block, ..., then (k + G) * block, ... while the others produce the blocks in between, and a single writer can drain
 * them in order from buffers that hold a few rows per GPU; contiguous shards would leave all but one GPU waiting.
 * v2m_upload_path_slice(first, n) is the one-block case.  Synchronous. */
int v2m_upload_path_blocks(v2m_ctx *ctx, const uint64_t *paths_by_edge_and_chrom_copy, uint64_t n_rows, uint64_t n_cols, uint64_t first_copy, uint64_t block_copies, uint64_t stride_copies, uint64_t copy_end);

/* The same for a transpose input that already lives in HBM (n_rows copies x n_cols edges, dense column-major words as
 * above, e.g. generated or assembled on the device): transposes it into a ctx-owned copy of paths_by_chrom_copy_and_edge and
 * binds that -- the one-call form of v2m_transpose_bits_device() + v2m_set_paths_device().  The owned copy is laid out for
 * the GPU (every chromosome copy's column starts on a 128-byte line), which a caller-visible destination cannot be, so
 * this is the faster of the two ways.  Asynchronous on the ctx's stream; the source may be reused once the call's work
 * has finished (v2m_ctx_synchronize()). */
int v2m_bind_path_matrix_device(v2m_ctx *ctx, const void *d_paths_by_edge_and_chrom_copy, uint64_t n_rows, uint64_t n_cols);

/* aligned_positions.back(): the length of every aligned row. 0 before an upload. */
uint64_t v2m_aligned_length(const v2m_ctx *ctx);
/* The row pitch the library itself uses in aligned mode (aligned length rounded up to 256: rows then start on
 * 256-B boundaries).  v2m_splice_rows_device accepts any multiple of 16 that is >= the aligned length rounded up to 16. */
uint64_t v2m_min_row_pitch(const v2m_ctx *ctx);

/* ---- rows -------------------------------------------------------------------------------- */

/* A batch of output rows, i.e. one sequence_writing_delegate per row:
 *   - row i with no cuts follows chromosome copy copy_index[i] for the whole walk
 *     (haplotype_output.cc:22-32); V2M_PLOIDY_MAX = the REF row (haplotype_output.cc:55).
 *   - row i with cuts [cut_offsets[i], cut_offsets[i+1]) starts as V2M_PLOIDY_MAX and switches to
 *     cut_copies[k] when the walk visits node cut_nodes[k] (founder_sequence_greedy_output.cc:106-114).
 *     Cut nodes must be strictly increasing and must not lie strictly inside any ALT edge's
 *     (source, target) node span -- the reference asserts this at :108; violating batches are
 *     rejected with V2M_ERR_PRECONDITION.  cut_copies may be V2M_PLOIDY_MAX.
 * cut_offsets == NULL means no row has cuts.  Copy indices must be < path_cols. */
typedef struct v2m_row_batch {
	uint64_t n_rows;
	const uint32_t *copy_index;  /* [n_rows] */
	const uint64_t *cut_offsets; /* [n_rows + 1] or NULL */
	const uint64_t *cut_nodes;   /* [cut_offsets[n_rows]] */
	const uint32_t *cut_copies;  /* [cut_offsets[n_rows]] */
} v2m_row_batch;

/* Receives one finished row body: exactly the bytes output_sequence() would have streamed
 * after the optional '>'id'\n' header and before the caller's '\n' (haplotype_output.cc:57,76).
 * `bytes` is library-owned pinned memory, valid only during the call.  Rows arrive in batch
 * order, one call per row, on the calling thread.  Return non-zero to abort (V2M_ERR_SINK). */
typedef int (*v2m_sink_fn)(void *user, uint64_t row_index, const char *bytes, uint64_t length);

/* Splices every row of the batch on the GPU and hands the bodies to `sink` in order.
 * Works through the batch in device-sized slices, overlapping the D2H copy of one slice with
 * the kernels of the next.  Synchronous. */
int v2m_splice_rows(v2m_ctx *ctx, const v2m_row_batch *rows, uint32_t flags, v2m_sink_fn sink, void *user);

/* The same rows for a sink that does not finish with a row inside the call -- one that queues it for a pool of writer threads
 * (output.cc:47-76 with more than one thread behind the file descriptor: a file per sequence, haplotype_output.cc:85-132, scales
 * with writers where one writer on one stream does not).  `bytes` stays valid, in a pinned slot of the library's, until the sink
 * calls v2m_row_release(hold) -- once per accepted row, from any thread, during or after the sink call.  Rows still arrive in batch
 * order on the calling thread, and the call goes on launching slices and copies meanwhile; a slot is copied into again only when
 * every row of it has been released.  n_slots (2 ... 64) pinned slots of the slice size v2m_splice_rows uses (512 MB for batches
 * of 8 GB and more, else 128 MB) are allocated on first use and kept by the ctx: rows in writers' hands + rows crossing the link
 * + rows being delivered all need a slot, so n_slots = writers' rows / rows per slot + 2 is the least that keeps the link busy.
 * A sink that returns non-zero has NOT accepted that row (it must not release it) and ends the call with V2M_ERR_SINK.
 * The call returns only when every accepted row has been released, also after an error.  Synchronous. */
typedef struct v2m_row_hold v2m_row_hold;
typedef int (*v2m_hold_sink_fn)(void *user, uint64_t row_index, const char *bytes, uint64_t length, v2m_row_hold *hold);
int v2m_splice_rows_held(v2m_ctx *ctx, const v2m_row_batch *rows, uint32_t flags, uint32_t n_slots, v2m_hold_sink_fn sink, void *user);
void v2m_row_release(v2m_row_hold *hold);

/* Device-resident form: row i is written to d_out + i * row_pitch and stays in HBM.
 * Aligned mode: every row has v2m_aligned_length() bytes; row_pitch a multiple of 16 and >= the aligned length
 * rounded up to 16 (v2m_min_row_pitch() is what the library uses itself); d_out 16-byte aligned; the bytes between
 * the row's end and the next multiple of 16 are clobbered.  Unaligned mode: row_pitch must be >= the longest row
 * (reference length + total label bytes is always enough; see v2m_max_unaligned_length()).
 * row_lengths_out (host, optional, [n_rows]) receives each row's length.
 * The call returns once the kernels are QUEUED on the ctx's stream (the batch's small row tables travel through two
 * pinned staging areas used in turn, so consecutive calls do not wait for each other's kernels); the first >= 1 GiB
 * launch of a ctx also times both store flavours, synchronously.
 * With row_lengths_out in unaligned mode the call waits for the kernels.  Use v2m_ctx_synchronize() before reading d_out. */
int v2m_splice_rows_device(v2m_ctx *ctx, const v2m_row_batch *rows, uint32_t flags, void *d_out, uint64_t row_pitch, uint64_t *row_lengths_out);

/* Device memory for row output (the d_out of v2m_splice_rows_device), chosen by measurement: on MI355X the write rate
 * the splice's store pattern reaches differs by up to ~25 % between hipMalloc'ed buffers, depending on how fragmented
 * their physical backing is, which a caller cannot see from a pointer.  Allocates up to `candidates` buffers of
 * `bytes` (fewer if HBM runs out -- all of them are held until the choice is made), times the store pattern on each,
 * keeps the fastest and frees the others; v2m_ctx_info() reports the rates.  candidates <= 1 (or a buffer too small to
 * probe) is a plain allocation.  Free with v2m_free_output().  Synchronous.
 * A side effect to know about: the driver wipes freed device memory that has been written, in the background, and while it
 * does (about 3 s per freed 63-GB candidate, measured) the process's device-to-host copies run at ~70 % of the link's rate
 * (profiles/r04/e2e_slow_after_alloc_output.txt).  A caller that is about to stream rows to the host (v2m_splice_rows) and
 * cares about those first seconds asks for one candidate. */
int v2m_alloc_output(v2m_ctx *ctx, uint64_t bytes, int candidates, void **d_out);
int v2m_free_output(v2m_ctx *ctx, void *d_ptr);

/* Upper bound of any unaligned row's length for the uploaded graph. */
uint64_t v2m_max_unaligned_length(const v2m_ctx *ctx);

/* ---- founder search: the chunk walks ---------------------------------- */

/* The edge-by-edge part of find_initial_cut_positions_lambda_min (libvcf2multialign/find_cut_positions.cc:126-176): the pBWT
 * steps of pbwt_context::update_divergence (include/vcf2multialign/pbwt.hh:77-134) and, at every candidate cut node, the walk
 * over the distinct divergence values from the largest down (find_cut_positions.cc:134-165) -- for chunks of consecutive
 * candidates whose start state the caller has built (the state after k edges is the copies sorted by their reversed k-edge
 * prefixes plus their divergence values; csrc/host/founder.cc builds it from the transposed matrix).  One workgroup walks
 * one chunk; the score updates of find_cut_positions.cc:55-63, which depend on each other, stay with the caller.
 *
 * The ctx must hold the uploaded graph WITH its path matrix (v2m_upload_graph with paths_by_chrom_copy_and_edge): the first
 * v2m_pbwt_* call after a matrix is bound transposes it back to edge-major bits on the device, and that copy (as large as the
 * matrix) is kept for the following v2m_pbwt_* calls -- one founder run makes two -- until the matrix is bound anew, a v2m_splice_rows*
 * call starts (the output that follows the searches gets the memory back) or the ctx is destroyed.  The copy is made from the matrix as it
 * is at that first call: a caller that changes a v2m_set_paths_device() matrix in place must bind it again before the next search.
 * n_copies <= 20480 and a bound matrix of at most 20480 copy columns (V2M_ERR_UNSUPPORTED beyond: a workgroup keeps the pBWT state
 * and one edge column in LDS; v2m_pbwt_cut_records as well: its class arrays sit beside them up to 12288 copies and in device memory above).  V2M_ERR_INVALID_ARGUMENT for candidate edges that decrease or lie outside the graph, aligned
 * positions that decrease (find_cut_positions.cc:129,151) and start_order entries >= n_copies -- checked on the host before any
 * kernel indexes with them.
 *   cand_edge[c], cand_aligned_pos[c]   of all n_candidates candidates: the index of the node's first ALT edge (ascending, one
 *                                       candidate per distinct edge index, :129) and the node's aligned position (:151)
 *   chunk_first[k] .. chunk_first[k+1]  the candidates of chunk k (n_chunks + 1 entries, ascending, chunk_first[0] >= 1)
 *   start_order, start_divergence       [n_chunks][n_copies]: the state after the first cand_edge[chunk_first[k]] edges;
 *                                       divergence values BIASED by one (0 = the reference's DIVERGENCE_MAX, pbwt.hh:25-42)
 * Outputs (host): for chunk k up to trial_capacity pairs at trial_pred / trial_class_count + k * trial_capacity -- (earlier
 * candidate, class count) in the order the reference's loop tries them --, trial_end[c] = the pairs of c's chunk up to and
 * including candidate c (written for the candidates of this call's chunks only), and chunk_status[k] = 0, or 1 when the chunk was left undone (more than 1024 distinct earlier
 * candidates at one node, or trial_capacity exceeded): the caller walks that chunk itself.  Synchronous. */
int v2m_pbwt_cut_trials(v2m_ctx *ctx, uint64_t n_copies, uint64_t min_distance,
	uint64_t n_candidates, const uint32_t *cand_edge, const uint64_t *cand_aligned_pos,
	uint64_t n_chunks, const uint64_t *chunk_first, const uint32_t *start_order, const uint32_t *start_divergence,
	uint64_t trial_capacity, uint32_t *trial_pred, uint32_t *trial_class_count, uint64_t *trial_end, uint32_t *chunk_status);

/* The same walks with the pairs handed over chunk by chunk instead of landing in arrays of the caller's: once all chunks are
 * walked, runs of whole chunks come back through two pinned slots of the library's in turn, and `sink` is called once per chunk,
 * in chunk order, on the calling thread -- (user, chunk index, status as above, the chunk's pairs, their number; the pointers are
 * valid during the call only; 0 pairs for a chunk left undone or empty) -- while the next slice is crossing the link.  trial_end
 * and chunk_status are complete before the first call.  The caller's score updates (find_cut_positions.cc:55-63) thus run under the
 * copies, and no large pageable array is ever a copy target (config 4: 800 MB of pairs; touching, pinning and releasing such an
 * array cost more than using it).  A sink that returns non-zero ends the call with V2M_ERR_SINK. */
typedef int (*v2m_trials_sink)(void *user, uint64_t chunk, uint32_t status, const uint32_t *pred, const uint32_t *class_count, uint64_t n_pairs);
int v2m_pbwt_cut_trials_streamed(v2m_ctx *ctx, uint64_t n_copies, uint64_t min_distance,
	uint64_t n_candidates, const uint32_t *cand_edge, const uint64_t *cand_aligned_pos,
	uint64_t n_chunks, const uint64_t *chunk_first, const uint32_t *start_order, const uint32_t *start_divergence,
	uint64_t trial_capacity, uint64_t *trial_end, uint32_t *chunk_status, v2m_trials_sink sink, void *sink_user);

/* The same for founder_sequence_greedy_output::find_matchings (libvcf2multialign/founder_sequence_greedy_output.cc:154-512): the
 * pBWT steps between cut positions and, at every cut, what the reference's loop :215-251 collects -- the number of path classes
 * of the block that ends there, the first copy in pBWT order and whether it followed REF edges only (:454-462), and the joined
 * classes {class in the previous block, class in this block, copies} of the two-block span, in pBWT order (the caller sorts
 * them by size, :256, and runs the greedy assignment, :254-457, which is strictly sequential).
 *   cut_edge[j]                   ALT edges before cut node j (alt_edge_count_csum[cut_positions[j]]), ascending, n_cuts entries
 *   chunk_first_cut[k] .. [k+1]   the cuts chunk k handles (>= 1); start_edge[k] <= cut_edge[chunk_first_cut[k] - 1]: the given
 *                                 state (start_order / start_divergence, [n_chunks][n_copies], divergence biased) is the one after
 *                                 that many edges; the kernel walks on to the cut before the chunk's first one by itself
 * Outputs (host): per chunk up to pool_capacity joined classes at pool_lhs / pool_rhs / pool_size + k * pool_capacity
 * (0xFFFFFFFF = PLOIDY_MAX, "no class"); per cut j >= 1 rec_pool_end[j] (the chunk's joined classes up to and including cut j),
 * rec_distinct[j], rec_first_class[j], rec_first_is_ref[j]; chunk_status[k] = 0, or 1 when the chunk was left undone
 * (pool_capacity exceeded).  Same requirements on the ctx and the start states as v2m_pbwt_cut_trials; cut edges that decrease or
 * lie outside the graph are V2M_ERR_INVALID_ARGUMENT.  Synchronous. */
int v2m_pbwt_cut_records(v2m_ctx *ctx, uint64_t n_copies, uint64_t n_cuts, const uint32_t *cut_edge,
	uint64_t n_chunks, const uint64_t *chunk_first_cut, const uint32_t *start_edge, const uint32_t *start_order, const uint32_t *start_divergence,
	uint64_t pool_capacity, uint32_t *pool_lhs, uint32_t *pool_rhs, uint32_t *pool_size,
	uint64_t *rec_pool_end, uint32_t *rec_distinct, uint32_t *rec_first_class, uint32_t *rec_first_is_ref, uint32_t *chunk_status);

/* ---- verification helper ------------------------------------------------------------------ */

/* 64-bit position-sensitive checksum of each of n_rows device rows (row i = d_rows + i*row_pitch,
 * lengths[i] bytes, or `length` for all rows when lengths == NULL; d_rows and row_pitch multiples of 8):
 *   the row is read as 8-byte little-endian words w[0 .. ceil(len/8)), the last one zero-padded past the row's end;
 *   checksum = sum over k of mix64((k + 1) * 0x9E3779B97F4A7C15 ^ w[k])  +  mix64(len)   (mod 2^64),
 *   mix64 = the splitmix64 finaliser (z ^= z >> 30; z *= 0xBF58476D1CE4E5B9; z ^= z >> 27; z *= 0x94D049BB133111EB; z ^= z >> 31).
 * Used to check full-size outputs against the CPU without moving them.  Synchronous. */
int v2m_checksum_rows_device(v2m_ctx *ctx, const void *d_rows, uint64_t row_pitch, uint64_t n_rows, uint64_t length, const uint64_t *lengths, uint64_t *checksums_out);

/* ---- kernel timing ------------------------------------------------------------------------ */

enum {
	V2M_KERNEL_TRANSPOSE = 0,       /* transpose_bits_kernel */
	V2M_KERNEL_RESOLVE = 1,         /* resolve_effective_edges_kernel (per-row skip-rule scan) */
	V2M_KERNEL_SPLICE_ALIGNED = 2,  /* splice_aligned_kernel (dominant) */
	V2M_KERNEL_SPLICE_UNALIGNED = 3, /* splice_unaligned_kernel (pass 2 of unaligned mode: build + compact the tiles) */
	V2M_KERNEL_TEMPLATE = 4,        /* expand_reference_row_kernel (once per upload) */
	V2M_KERNEL_UNALIGNED_COUNT = 5, /* count_unaligned_kernel + scan_tile_counts_kernel (pass 1 of unaligned mode) */
	V2M_KERNEL_COUNT = 6
};

/* When enabled, every launch of the kernels above is bracketed by HIP events on the ctx's
 * stream.  v2m_profile_get() synchronises and returns launch count and summed device time. */
int v2m_profile_enable(v2m_ctx *ctx, int enabled);
int v2m_profile_reset(v2m_ctx *ctx);
int v2m_profile_get(v2m_ctx *ctx, int kernel, uint64_t *launches_out, double *total_ms_out);
/* Per-launch device times (ms) in launch order; writes min(capacity, launches) values, returns the launch count in *launches_out. */
int v2m_profile_get_launches(v2m_ctx *ctx, int kernel, double *ms_out, uint64_t capacity, uint64_t *launches_out);

#ifdef __cplusplus
}
#endif

#endif /* V2M_HIP_H */
