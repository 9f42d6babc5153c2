"""Pins the CPU oracle against every golden value the reference's tests hold for this path.

Goldens: tests/golden/reference_goldens.json (extracted from /root/reference/tests/*.cc by
tests/golden/extract_reference_goldens.py); fixtures: tests/golden/reference-fixtures/ (the
reference's own FASTA/VCF test data).
"""

import os

import numpy as np
import pytest

import oracle


def _case_id(c):
	return c["vcf"] + "+" + c["fasta"]


def _load(name):
	import json
	here = os.path.dirname(os.path.abspath(__file__))
	with open(os.path.join(here, "golden", "reference_goldens.json")) as f:
		return json.load(f)[name]


# --- tests/variant_graph.cc:247-339 ---------------------------------------------------------
@pytest.mark.parametrize("case", _load("variant_graph"), ids=_case_id)
def test_graph_tables(case, fixtures_dir):
	d = os.path.join(fixtures_dir, "variant-graph")
	g = oracle.build_variant_graph(os.path.join(d, case["fasta"]), os.path.join(d, case["vcf"]), case["chromosome"])
	nodes = case["nodes"]
	assert g.node_count == len(nodes)
	labels = g.labels()
	for n in nodes:
		i = n["node"]
		assert int(g.reference_positions[i]) == n["ref_pos"]
		assert int(g.aligned_positions[i]) == n["aln_pos"]
		lo, hi = int(g.alt_edge_count_csum[i]), int(g.alt_edge_count_csum[i + 1])
		got = [{"target": int(g.alt_edge_targets[e]), "label": labels[e]} for e in range(lo, hi)]
		assert got == n["alt_edges"], "node %d" % i
		# the table's ref column (not checked by the reference, tests/variant_graph.cc:85-116) holds for our reader too
		if i + 1 < len(nodes):
			assert g.ref[n["ref_pos"]:nodes[i + 1]["ref_pos"]].decode() == n["ref"]
	got_ov = [{k: o[k] for k in ("sample", "chrom_copy_idx", "ref_pos", "var_id", "gt")} for o in g.overlaps()]
	assert got_ov == case["expected_overlaps"]


# --- tests/founder_sequences.cc:118-188 -----------------------------------------------------
@pytest.mark.parametrize("case", _load("founder_sequences"), ids=_case_id)
def test_founder_a2m(case, fixtures_dir):
	d = os.path.join(fixtures_dir, "founder-sequences")
	g = oracle.build_variant_graph(os.path.join(d, case["fasta"]), os.path.join(d, case["vcf"]), case["chromosome"])
	out = g.founder_output_a2m(g.ref, case["cut_positions"], case["assigned_samples_column_major"], case["founder_count"])
	assert out.decode() == case["expected_a2m"]
	# the same rows one by one through output_sequence with (cut node, copy) pairs
	rows = case["assigned_samples_rows"]
	assert rows == len(case["cut_positions"]) - 1
	expected_rows = case["expected_a2m"].split("\n")
	assert g.output_sequence(g.ref).decode() == expected_rows[1]
	for f in range(case["founder_count"]):
		col = case["assigned_samples_column_major"][f * rows:(f + 1) * rows]
		cuts = list(zip(case["cut_positions"][:-1], col))
		assert g.output_sequence(g.ref, cuts=cuts).decode() == expected_rows[3 + 2 * f]


# REQUIRE(expected_cut_positions == output.cut_positions()), REQUIRE(expected_matchings == output.assigned_samples())
# (tests/founder_sequences.cc:126-128): the oracle's literal pBWT / cut search / greedy matching
@pytest.mark.parametrize("case", _load("founder_sequences"), ids=_case_id)
def test_founder_search(case, fixtures_dir):
	d = os.path.join(fixtures_dir, "founder-sequences")
	g = oracle.build_variant_graph(os.path.join(d, case["fasta"]), os.path.join(d, case["vcf"]), case["chromosome"])
	cuts, assigned, _ = g.find_founders(case["founder_count"], case["minimum_distance"], keep_ref_edges=False)
	assert cuts == case["cut_positions"]
	assert assigned == case["assigned_samples_column_major"]


# --- tests/transpose_matrix.cc:188-251 ------------------------------------------------------
def _set_bit(words, nrows, r, c):
	idx = c * nrows + r
	words[idx >> 6] |= np.uint64(1) << np.uint64(idx & 63)


@pytest.mark.parametrize("case", _load("transpose_matrix"), ids=lambda c: "%dx%d" % (c["rows"], c["cols"]))
def test_transpose_fixed(case):
	src = np.zeros(case["rows"] * case["cols"] // 64, dtype=np.uint64)
	exp = np.zeros_like(src)
	_set_bit(src, case["rows"], *case["set_bit"])
	_set_bit(exp, case["expected_rows"], *case["expected_bit"])
	for naive in (False, True):
		got = oracle.transpose_matrix(src, case["rows"], case["cols"], naive=naive)
		assert np.array_equal(got, exp)


# --- tests/transpose_matrix.cc:254-279 (RapidCheck property: 64h x 64w, 1/3 of the bits set) --
@pytest.mark.parametrize("seed", range(12))
def test_transpose_property(seed):
	rng = np.random.default_rng(seed)
	h, w = int(rng.integers(1, 9)), int(rng.integers(1, 9))
	rows, cols = 64 * h, 64 * w
	count = rows * cols
	pos = rng.choice(count, size=count // 3, replace=False)
	r, c = pos // cols, pos % cols
	src = np.zeros(count // 64, dtype=np.uint64)
	exp = np.zeros(count // 64, dtype=np.uint64)
	si = c * rows + r   # input(row, col): column-major, rows per column = rows
	ei = r * cols + c   # expected(col, row): its "rows" are cols
	np.bitwise_or.at(src, si >> 6, np.uint64(1) << (si & 63).astype(np.uint64))
	np.bitwise_or.at(exp, ei >> 6, np.uint64(1) << (ei & 63).astype(np.uint64))
	assert np.array_equal(oracle.transpose_matrix(src, rows, cols), exp)
	assert np.array_equal(oracle.transpose_matrix(src, rows, cols, naive=True), exp)
	# involution
	assert np.array_equal(oracle.transpose_matrix(exp, cols, rows), src)


def test_transpose_empty_and_bad_dims():
	assert oracle.transpose_matrix(np.zeros(0, np.uint64), 64, 0).size == 0   # transpose_matrix.cc:48-49
	with pytest.raises(ValueError):
		oracle.transpose_matrix(np.zeros(2, np.uint64), 64, 2)


# --- the builder's own final transpose (variant_graph.cc:453) -------------------------------
@pytest.mark.parametrize("case", _load("variant_graph"), ids=_case_id)
def test_graph_path_matrices_are_transposes(case, fixtures_dir):
	d = os.path.join(fixtures_dir, "variant-graph")
	g = oracle.build_variant_graph(os.path.join(d, case["fasta"]), os.path.join(d, case["vcf"]), case["chromosome"])
	hp, ep = g.paths_by_edge_and_chrom_copy_dims
	assert hp % 64 == 0 and ep % 64 == 0 and (g.path_rows, g.path_cols) == (ep, hp)
	assert np.array_equal(oracle.transpose_matrix(g.paths_by_edge_and_chrom_copy, hp, ep, naive=True), g.paths_by_chrom_copy_and_edge)


def test_row_checksum_is_the_documented_formula(fixtures_dir):
	"""v2mo_row_checksum (used by the full-size GPU tests and bench.py) == the numpy form of v2m_checksum_rows_device's
	formula over the bytes output_sequence() returns; haplotype rows, REF, founder rows, aligned and unaligned."""
	import numpy as np
	from vcf2multialign_amd.context import checksum_rows_host
	d = os.path.join(fixtures_dir, "variant-graph")
	g = oracle.build_variant_graph(os.path.join(d, "test-4.fa"), os.path.join(d, "test-4.vcf"), "1")
	rows = [oracle.PLOIDY_MAX] + list(range(g.total_chromosome_copies)) + [[(0, 1), (g.node_count - 1, 0)]]
	for unaligned in (False, True):
		exp = [g.output_sequence(g.ref, cuts=r, unaligned=unaligned) if isinstance(r, list) else g.output_sequence(g.ref, copy_index=r, unaligned=unaligned) for r in rows]
		sums, lengths = g.row_checksums(g.ref, rows, unaligned=unaligned, threads=3)
		assert np.array_equal(sums, checksum_rows_host(exp))
		assert lengths.tolist() == [len(e) for e in exp]
