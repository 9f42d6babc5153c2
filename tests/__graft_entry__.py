"""Driver hooks: build() compiles every native piece for gfx950; smoke() runs one tiny invocation of
the hot path on cuda:0 and checks it against the CPU oracle."""

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
	if p not in sys.path:
		sys.path.insert(0, p)


def build():
	"""hipcc --offload-arch=gfx950 for the product library, g++ for the CPU oracle (the checker)."""
	from vcf2multialign_amd import build as b
	b.build_native(verbose=True)
	subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
	import vcf2multialign_amd  # noqa: F401
	lib = vcf2multialign_amd.load_library()
	assert lib.v2m_abi_version() == 1


def smoke():
	"""Reference fixture test-4 (SNV, MNP, ins, del, <DEL>, overlapping deletions): --haplotypes A2M through
	the GPU vs the oracle, plus one bit-matrix transpose."""
	import io

	import numpy as np

	import oracle
	import vcf2multialign_amd as v2m

	d = os.path.join(ROOT, "tests", "golden", "reference-fixtures", "variant-graph")
	g = oracle.build_variant_graph(os.path.join(d, "test-4.fa"), os.path.join(d, "test-4.vcf"), "1")
	with v2m.Context(0) as ctx:
		vg = v2m.VariantGraph.from_object(g)
		ctx.upload_graph(vg, g.ref)
		out = io.BytesIO()
		v2m.HaplotypeOutput(ctx).output_a2m(vg, out)
		with open(os.path.join(ROOT, "tests", "golden", "derived", "test-4.haplotypes.a2m"), "rb") as f:
			expected = f.read()
		assert out.getvalue() == expected, "GPU A2M differs from the oracle's"
		rows = [v2m.PLOIDY_MAX] + list(range(g.total_chromosome_copies))
		for r, body in zip(rows, ctx.splice_rows(rows)):
			assert body == g.output_sequence(g.ref, copy_index=r)

		hp, ep = g.paths_by_edge_and_chrom_copy_dims
		got = ctx.transpose_matrix(g.paths_by_edge_and_chrom_copy, hp, ep)
		assert np.array_equal(got, g.paths_by_chrom_copy_and_edge)
		assert np.array_equal(got, oracle.transpose_matrix(g.paths_by_edge_and_chrom_copy, hp, ep))
	print("smoke ok: %d rows of %d aligned bases, transpose %dx%d" % (len(rows), g.aligned_length, hp, ep))


if __name__ == "__main__":
	build()
	if len(sys.argv) > 1 and sys.argv[1] == "smoke":
		smoke()
