"""Host-side variant graph container (numpy arrays), field for field what the hot path reads
from the reference's variant_graph (include/vcf2multialign/variant_graph.hh:57-66), with
alt_edge_labels flattened to CSR.  Integers stay 64-bit at this level, as in the reference."""

import numpy as np

PLOIDY_MAX = 0xFFFFFFFF


class VariantGraph:
	def __init__(self, reference_positions, aligned_positions, alt_edge_targets, alt_edge_count_csum,
			label_offsets, label_bytes, paths_by_chrom_copy_and_edge=None, path_rows=0, path_cols=0,
			sample_names=(), ploidy_csum=None):
		u64 = lambda x: np.ascontiguousarray(x, dtype=np.uint64)
		self.reference_positions = u64(reference_positions)       # [N]
		self.aligned_positions = u64(aligned_positions)           # [N]
		self.alt_edge_targets = u64(alt_edge_targets)             # [E]
		self.alt_edge_count_csum = u64(alt_edge_count_csum)       # [N + 1]
		self.label_offsets = u64(label_offsets)                   # [E + 1]
		self.label_bytes = bytes(label_bytes)
		# rows = edges (path_rows = Ep), cols = chromosome copies (path_cols = Hp); column-major u64 words
		self.paths_by_chrom_copy_and_edge = None if paths_by_chrom_copy_and_edge is None else u64(paths_by_chrom_copy_and_edge)
		self.path_rows = int(path_rows)
		self.path_cols = int(path_cols)
		self.sample_names = list(sample_names)
		self.ploidy_csum = np.ascontiguousarray(ploidy_csum if ploidy_csum is not None else [0], dtype=np.uint32)

	@classmethod
	def from_object(cls, g):
		"""From anything exposing the same attribute names (e.g. the test oracle's graph)."""
		return cls(g.reference_positions, g.aligned_positions, g.alt_edge_targets, g.alt_edge_count_csum,
			g.label_offsets, g.label_bytes, g.paths_by_chrom_copy_and_edge, g.path_rows, g.path_cols,
			getattr(g, "sample_names", ()), getattr(g, "ploidy_csum", None))

	@property
	def node_count(self):
		return len(self.reference_positions)

	@property
	def edge_count(self):
		return len(self.alt_edge_targets)

	@property
	def aligned_length(self):
		return int(self.aligned_positions[-1]) if self.node_count else 0

	# variant_graph.hh:73-74
	def sample_ploidy(self, sample_idx):
		return int(self.ploidy_csum[sample_idx + 1]) - int(self.ploidy_csum[sample_idx])

	def total_chromosome_copies(self):
		return int(self.ploidy_csum[-1])
