#!/usr/bin/env python3
"""Writes tests/golden/derived/*.a2m: --haplotypes output of the CPU oracle on the reference's
variant-graph fixtures.

DERIVED, NOT REFERENCE-PRODUCED: the reference holds no golden for haplotype_output
("Unpinned by the reference"); these files are what the oracle -- which
reproduces every golden the reference does hold -- emits, and they equal SURVEY.md Appendix A.
They guard against regressions of the oracle and give the HIP path file-level targets.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402

CASES = [("test-1.fa", "test-1a.vcf"), ("test-1.fa", "test-1b.vcf"), ("test-2.fa", "test-2.vcf"), ("test-3.fa", "test-3.vcf"), ("test-4.fa", "test-4.vcf")]


def main():
	out = os.path.join(HERE, "derived")
	os.makedirs(out, exist_ok=True)
	d = os.path.join(HERE, "reference-fixtures", "variant-graph")
	for fa, vcf in CASES:
		g = oracle.build_variant_graph(os.path.join(d, fa), os.path.join(d, vcf), "1")
		stem = vcf[:-4]
		g.haplotype_output_a2m(g.ref, os.path.join(out, stem + ".haplotypes.a2m"))
		g.haplotype_output_a2m(g.ref, os.path.join(out, stem + ".haplotypes.unaligned.fa"), unaligned=True)
		g.haplotype_output_a2m(g.ref, os.path.join(out, stem + ".haplotypes.chr.noref.a2m"), chromosome_id="chrT", output_reference=False)


if __name__ == "__main__":
	main()
