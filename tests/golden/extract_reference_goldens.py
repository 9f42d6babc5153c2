#!/usr/bin/env python3
"""Regenerates tests/golden/reference_goldens.json from the reference's own test sources.

Runs only where /root/reference exists (the build container).  It reads the
reference's Catch2 test files AS TEXT and extracts the DATA they hold -- the
expected A2M strings, node/edge tables, cut positions, matching matrices, the
expected overlap report and the fixed transpose cases -- into a JSON fixture.
No reference code is copied, compiled, imported or executed.

Sources (file:line in /root/reference):
  tests/founder_sequences.cc:118-188   five (vcf, fasta, cuts, matchings, A2M) cases
  tests/variant_graph.cc:247-339       five node tables + expected overlaps
  tests/transpose_matrix.cc:188-251    three fixed single-bit cases
"""

import json
import os
import re
import sys

REF = "/root/reference/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_goldens.json")


def c_string_literals(block):
	"""Concatenate adjacent C string literals, decoding \\n and \\t."""
	parts = re.findall(r'"((?:[^"\\]|\\.)*)"', block)
	return "".join(p.encode().decode("unicode_escape") for p in parts)


def ints(s):
	return [int(x) for x in re.findall(r"\d+", s)]


def founder_cases(text):
	cases = []
	# each GIVEN block: expected_output( "..." ... ); test_founders("vcf", "fa", {cuts}, {{vals}, rows}, expected_output);
	for m in re.finditer(
		r'expected_output\(\s*((?:"(?:[^"\\]|\\.)*"\s*)+)\);\s*'
		r'test_founders\("([^"]+)",\s*"([^"]+)",\s*\{([^}]*)\},\s*\{\{([^}]*)\},\s*(\d+)\},\s*expected_output\);',
		text,
	):
		a2m, vcf, fa, cuts, vals, rows = m.groups()
		cases.append({
			"vcf": vcf,
			"fasta": fa,
			"chromosome": "1",
			"minimum_distance": 0,
			"founder_count": 2,
			"cut_positions": ints(cuts),
			"assigned_samples_column_major": ints(vals),
			"assigned_samples_rows": int(rows),
			"expected_a2m": c_string_literals(a2m),
		})
	return cases


def graph_cases(text):
	cases = []
	for m in re.finditer(
		r'node_comparator cmp\{\s*\{(.*?)\}\s*\};\s*test_variant_graph\("([^"]+)",\s*"([^"]+)",\s*cmp,\s*\{(.*?)\}\);',
		text,
		re.S,
	):
		body, vcf, fa, overlaps = m.groups()
		nodes = []
		for nm in re.finditer(r'\{(\d+),\s*(\d+),\s*(\d+),\s*"([^"]*)",\s*\{((?:\{[^}]*\},?\s*)*)\}\}', body):
			node, pos, aln, ref, edges = nm.groups()
			alt_edges = [
				{"target": int(t), "label": l}
				for t, l in re.findall(r'\{(\d+),\s*"([^"]*)"\}', edges)
			]
			nodes.append({"node": int(node), "ref_pos": int(pos), "aln_pos": int(aln), "ref": ref, "alt_edges": alt_edges})
		ov = [
			{"sample": s, "chrom_copy_idx": int(c), "ref_pos": int(p), "var_id": v, "gt": int(g)}
			for s, c, p, v, g in re.findall(r'make_alt\s*<std::string>\("([^"]+)",\s*(\d+),\s*(\d+),\s*"([^"]+)",\s*(\d+)\)', overlaps)
		]
		cases.append({"vcf": vcf, "fasta": fa, "chromosome": "1", "nodes": nodes, "expected_overlaps": ov})
	return cases


def transpose_cases(text):
	cases = []
	for m in re.finditer(
		r'lb::bit_matrix input\((\d+),\s*(\d+)\);\s*lb::bit_matrix expected\((\d+),\s*(\d+)\);\s*'
		r'input\((\d+),\s*(\d+)\)\s*\|=\s*1;\s*expected\((\d+),\s*(\d+)\)\s*\|=\s*1;',
		text,
	):
		v = [int(x) for x in m.groups()]
		cases.append({
			"rows": v[0], "cols": v[1], "expected_rows": v[2], "expected_cols": v[3],
			"set_bit": [v[4], v[5]], "expected_bit": [v[6], v[7]],
		})
	return cases


def main():
	if not os.path.isdir(REF):
		sys.exit("the reference tree is not present; the committed JSON is the fixture")
	with open(os.path.join(REF, "founder_sequences.cc")) as f:
		founders = founder_cases(f.read())
	with open(os.path.join(REF, "variant_graph.cc")) as f:
		graphs = graph_cases(f.read())
	with open(os.path.join(REF, "transpose_matrix.cc")) as f:
		transposes = transpose_cases(f.read())
	assert len(founders) == 5, len(founders)
	assert len(graphs) == 5, len(graphs)
	assert len(transposes) == 3, len(transposes)
	doc = {
		"_provenance": "extracted by tests/golden/extract_reference_goldens.py from /root/reference/tests/*.cc (data only)",
		"founder_sequences": founders,
		"variant_graph": graphs,
		"transpose_matrix": transposes,
	}
	with open(OUT, "w") as f:
		json.dump(doc, f, indent=1, sort_keys=True)
		f.write("\n")
	print("wrote", OUT)


if __name__ == "__main__":
	main()
