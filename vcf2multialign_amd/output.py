"""Host-side mirror of the reference's output classes for this path (include/vcf2multialign/output.hh:41-130):
the row order, FASTA identifiers and '\\n' placement of haplotype_output::output_a2m
(libvcf2multialign/haplotype_output.cc:38-82) and founder_sequence_greedy_output::output_a2m
(libvcf2multialign/founder_sequence_greedy_output.cc:515-550), with every row body produced by the
GPU through v2m_splice_rows instead of output_sequence()."""

from .context import RowBatch
from .variant_graph import PLOIDY_MAX


class Output:
	def __init__(self, ctx, chromosome_id=None, should_output_reference=True, should_output_unaligned=False):
		self.ctx = ctx
		self.chromosome_id = chromosome_id
		self.should_output_reference = should_output_reference
		self.should_output_unaligned = should_output_unaligned

	def _fasta_id(self, name):
		return ((self.chromosome_id + "\t") if self.chromosome_id else "") + name

	def _write_rows(self, stream, ids, rows):
		def sink(i, body):
			stream.write(b">" + ids[i].encode() + b"\n")
			stream.write(body)
			stream.write(b"\n")
		self.ctx.splice_rows(RowBatch(rows), sink=sink, unaligned=self.should_output_unaligned)


class HaplotypeOutput(Output):
	"""haplotype_output (output.hh:71-78).  The graph must already be uploaded to ctx."""

	def output_a2m(self, graph, stream):
		ids, rows = [], []
		if self.should_output_reference:                                  # haplotype_output.cc:48-59
			ids.append(self._fasta_id("REF"))
			rows.append(PLOIDY_MAX)
		for sample_idx, sample in enumerate(graph.sample_names):          # :62
			for chr_copy_idx in range(graph.sample_ploidy(sample_idx)):   # :65
				ids.append(self._fasta_id("%s-%d" % (sample, 1 + chr_copy_idx)))     # :69-72
				rows.append(int(graph.ploidy_csum[sample_idx]) + chr_copy_idx)     # :28-31
		self._write_rows(stream, ids, rows)
		return len(rows)


class FounderSequenceGreedyOutput(Output):
	"""The output half of founder_sequence_greedy_output (output.hh:81-130): given cut positions and the
	matching matrix (found on the host), writes REF + one row per founder."""

	def output_a2m(self, graph, cut_positions, assigned_samples_column_major, stream):
		n_rows = len(cut_positions) - 1
		n_founders = len(assigned_samples_column_major) // n_rows if n_rows else 0
		ids, rows = [], []
		if self.should_output_reference:                                  # founder_sequence_greedy_output.cc:519-531
			ids.append(self._fasta_id("REF"))
			rows.append(PLOIDY_MAX)
		for col_idx in range(n_founders):                                 # :533-549
			ids.append(self._fasta_id(str(1 + col_idx)))
			col = assigned_samples_column_major[col_idx * n_rows:(col_idx + 1) * n_rows]
			rows.append(list(zip(cut_positions[:-1], col)))               # delegate: switch copy at each cut node (:106-114)
		self._write_rows(stream, ids, rows)
		return len(rows)
