/* The multi-GPU call pattern from plain C, runnable on one GPU: two contexts (one per GPU in real use; both on device 0
 * unless two devices are named), graph and reference replicated, the path matrix NOT: each context receives only the bits of
 * its own chromosome copies out of the host-resident transpose input, transposes them itself (v2m_upload_path_slice) and
 * splices the rows of those copies, with copy indices relative to its shard.  No collective, no matrix coming back.
 *
 *   gcc -std=c99 -I../include sharded_rows.c -L../vcf2multialign_amd -lv2m_hip -Wl,-rpath,../vcf2multialign_amd -o sharded_rows
 *   ./sharded_rows [device_a device_b]
 *
 * Reference: ACGTACGTAC.  Edge 0: node 1 -> 2, "G" -> "GTT" (insertion); edge 1: node 3 -> 4, "GT" -> "G" (deletion).
 * 16 chromosome copies: copy c follows edge 0 if c is odd and edge 1 if c % 4 >= 2.  Context 0 owns copies [0, 8) and the
 * REF row, context 1 copies [8, 16).  Every row is compared with what output_sequence() gives for it.
 */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <v2m_hip.h>

enum { N_COPIES = 16, SHARD = 8 };

struct sink_state { int first_row; char rows[1 + N_COPIES][16]; };

static int keep_row(void *user, uint64_t row, const char *bytes, uint64_t length)
{
	struct sink_state *st = (struct sink_state *) user;
	if (length >= sizeof(st->rows[0])) return 1;
	memcpy(st->rows[st->first_row + row], bytes, length);
	st->rows[st->first_row + row][length] = 0;
	return 0;
}

static const char *expected_row(int copy)   /* copy < 0: REF */
{
	int const e0 = copy >= 0 && (copy & 1), e1 = copy >= 0 && (copy % 4 >= 2);
	if (e0 && e1) return "ACGTTTACG-AC";
	if (e0) return "ACGTTTACGTAC";
	if (e1) return "ACG--TACG-AC";
	return "ACG--TACGTAC";
}

int main(int argc, char **argv)
{
	static const uint64_t ref_pos[] = {0, 2, 3, 6, 8, 10};
	static const uint64_t aln_pos[] = {0, 2, 5, 8, 10, 12};
	static const uint64_t targets[] = {2, 4};
	static const uint64_t csum[] = {0, 0, 1, 1, 2, 2, 2};
	static const uint64_t label_offsets[] = {0, 3, 4};
	static const char label_bytes[] = "GTTG";
	static const char reference[] = "ACGTACGTAC";
	int const devices[2] = {argc > 2 ? atoi(argv[1]) : 0, argc > 2 ? atoi(argv[2]) : 0};

	/* paths_by_edge_and_chrom_copy as the builder leaves it before variant_graph.cc:453: rows = copies (64), cols = edges (64) */
	uint64_t by_edge[64] = {0};
	for (int c = 0; c < N_COPIES; ++c) {
		if (c & 1) by_edge[0] |= (uint64_t) 1 << c;
		if (c % 4 >= 2) by_edge[1] |= (uint64_t) 1 << c;
	}

	v2m_graph_view view;
	memset(&view, 0, sizeof(view));
	view.node_count = 6;
	view.edge_count = 2;
	view.reference_positions = ref_pos;
	view.aligned_positions = aln_pos;
	view.alt_edge_targets = targets;
	view.alt_edge_count_csum = csum;
	view.alt_edge_label_offsets = label_offsets;
	view.alt_edge_label_bytes = label_bytes;            /* no path matrix here: it arrives per context below */

	struct sink_state st;
	memset(&st, 0, sizeof(st));
	for (int k = 0; k < 2; ++k) {
		v2m_ctx *ctx = NULL;
		int rc = v2m_ctx_create(devices[k], &ctx);
		if (V2M_OK != rc) { fprintf(stderr, "v2m_ctx_create: error %d: %s\n", rc, v2m_last_error(NULL)); return 2; }
		rc = v2m_upload_graph(ctx, &view, reference, 10);
		if (V2M_OK == rc) rc = v2m_upload_path_slice(ctx, by_edge, 64, 64, (uint64_t) k * SHARD, SHARD);   /* this context's copies only */
		if (V2M_OK != rc) { fprintf(stderr, "%s\n", v2m_last_error(ctx)); return 1; }

		uint32_t copies[1 + SHARD];
		uint64_t n = 0;
		if (0 == k) copies[n++] = V2M_PLOIDY_MAX;          /* the REF row belongs to the first context */
		for (uint32_t c = 0; c < SHARD; ++c) copies[n++] = c;   /* relative to the shard's first copy */
		v2m_row_batch batch;
		memset(&batch, 0, sizeof(batch));
		batch.n_rows = n;
		batch.copy_index = copies;
		st.first_row = 0 == k ? 0 : 1 + SHARD;
		rc = v2m_splice_rows(ctx, &batch, 0, keep_row, &st);
		if (V2M_OK != rc) { fprintf(stderr, "%s\n", v2m_last_error(ctx)); return 1; }

		/* a copy of the other shard is not here */
		copies[0] = 64;
		batch.n_rows = 1;
		if (V2M_OK == v2m_splice_rows(ctx, &batch, 0, keep_row, &st)) { fprintf(stderr, "context %d accepted a copy outside its slice\n", k); return 1; }
		v2m_ctx_destroy(ctx);
	}

	int bad = 0;
	for (int r = 0; r <= N_COPIES; ++r) {
		printf("%-7s %s\n", r ? "copy" : "REF", st.rows[r]);
		if (strcmp(st.rows[r], expected_row(r - 1))) { fprintf(stderr, "row %d: expected %s\n", r, expected_row(r - 1)); bad = 1; }
	}
	return bad;
}
