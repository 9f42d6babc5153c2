/* The drop-in boundary from plain C: a five-node graph with two ALT edges and two chromosome copies, built by hand,
 * spliced into aligned rows on GPU 0 through libv2m_hip.so, and the path matrix transposed on the way.
 *
 *   gcc -std=c99 -I../include splice_rows.c -L../vcf2multialign_amd -lv2m_hip -Wl,-rpath,../vcf2multialign_amd -o splice_rows
 *
 * Reference: ACGTACGT.  Edge 0: node 1 -> 2, "G" -> "GTT" (insertion); edge 1: node 3 -> 4, "GT" -> "G" (deletion).
 * Copy 0 follows edge 0, copy 1 follows edge 1.  Expected output (what output_sequence() gives):
 *   REF     ACG--TACGT
 *   copy 0  ACGTTTACGT
 *   copy 1  ACG--TACG-
 */
#include <stdio.h>
#include <string.h>
#include <v2m_hip.h>

static int print_row(void *user, uint64_t row, const char *bytes, uint64_t length)
{
	static const char *const names[] = {"REF", "copy 0", "copy 1"};
	char (*seen)[16] = (char (*)[16]) user;
	printf("%-7s %.*s\n", names[row], (int) length, bytes);
	if (length < sizeof(seen[row])) { memcpy(seen[row], bytes, length); seen[row][length] = 0; }
	return 0;
}

int main(void)
{
	/* nodes: (reference position, aligned position) */
	static const uint64_t ref_pos[] = {0, 2, 3, 6, 8};
	static const uint64_t aln_pos[] = {0, 2, 5, 8, 10};
	static const uint64_t targets[] = {2, 4};
	static const uint64_t csum[] = {0, 0, 1, 1, 2, 2};        /* edges of node n: [csum[n], csum[n+1]) */
	static const uint64_t label_offsets[] = {0, 3, 4};
	static const char label_bytes[] = "GTTG";
	static const char reference[] = "ACGTACGT";

	/* paths_by_edge_and_chrom_copy as the builder fills it: rows = copies, cols = edges, 64 x 64 bits */
	uint64_t by_edge[64] = {0}, by_copy[64] = {0};
	by_edge[0] = 1u << 0;                                      /* edge 0 is used by copy 0 */
	by_edge[1] = 1u << 1;                                      /* edge 1 is used by copy 1 */

	v2m_ctx *ctx = NULL;
	int rc = v2m_ctx_create(0, &ctx);
	if (V2M_OK != rc) { fprintf(stderr, "v2m_ctx_create: error %d (no gfx950 device?)\n", rc); return 2; }

	rc = v2m_transpose_bits(ctx, by_edge, 64, 64, by_copy);    /* transpose_matrix() */
	if (V2M_OK != rc) { fprintf(stderr, "%s\n", v2m_last_error(ctx)); return 1; }

	v2m_graph_view view;
	memset(&view, 0, sizeof(view));
	view.node_count = 5;
	view.edge_count = 2;
	view.reference_positions = ref_pos;
	view.aligned_positions = aln_pos;
	view.alt_edge_targets = targets;
	view.alt_edge_count_csum = csum;
	view.alt_edge_label_offsets = label_offsets;
	view.alt_edge_label_bytes = label_bytes;
	view.paths_by_chrom_copy_and_edge = by_copy;
	view.path_rows = 64;
	view.path_cols = 64;
	rc = v2m_upload_graph(ctx, &view, reference, 8);
	if (V2M_OK != rc) { fprintf(stderr, "%s\n", v2m_last_error(ctx)); return 1; }

	static const uint32_t copies[] = {V2M_PLOIDY_MAX, 0, 1};   /* REF, then the two chromosome copies */
	v2m_row_batch batch;
	memset(&batch, 0, sizeof(batch));
	batch.n_rows = 3;
	batch.copy_index = copies;
	char seen[3][16];
	memset(seen, 0, sizeof(seen));
	rc = v2m_splice_rows(ctx, &batch, 0, print_row, seen);     /* the output_sequence() calls of one output_a2m() */
	if (V2M_OK != rc) { fprintf(stderr, "%s\n", v2m_last_error(ctx)); return 1; }
	v2m_ctx_destroy(ctx);

	if (strcmp(seen[0], "ACG--TACGT") || strcmp(seen[1], "ACGTTTACGT") || strcmp(seen[2], "ACG--TACG-")) {
		fprintf(stderr, "unexpected rows\n");
		return 1;
	}
	return 0;
}
