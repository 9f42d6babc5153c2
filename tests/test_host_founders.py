"""The host's cut-position search and greedy matching (vcf2multialign_amd/csrc/host/founder.cc) against the values the
reference's own test pins (tests/founder_sequences.cc:118-188: find_cut_positions(graph, 0), find_matchings(graph, 2))."""

import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
	with open(os.path.join(HERE, "golden", "reference_goldens.json")) as f:
		return json.load(f)["founder_sequences"]


@pytest.fixture(scope="module")
def HostGraph():
	from vcf2multialign_amd import build
	build.build_native()
	from vcf2multialign_amd.host import HostGraph
	return HostGraph


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["vcf"] + "+" + c["fasta"])
def test_cut_positions_and_matchings(HostGraph, case, fixtures_dir):
	d = os.path.join(fixtures_dir, "founder-sequences")
	h = HostGraph(os.path.join(d, case["fasta"]), os.path.join(d, case["vcf"]), case["chromosome"])
	cuts, assigned, score = h.find_founders(case["founder_count"], case["minimum_distance"], keep_ref_edges=False)
	assert cuts == case["cut_positions"]                               # REQUIRE(expected_cut_positions == output.cut_positions())
	assert assigned == case["assigned_samples_column_major"]           # REQUIRE(expected_matchings == output.assigned_samples())
	assert len(assigned) == case["assigned_samples_rows"] * case["founder_count"]


def test_minimum_distance_and_founder_count_are_honoured(HostGraph, fixtures_dir):
	d = os.path.join(fixtures_dir, "founder-sequences")
	h = HostGraph(os.path.join(d, "test-1.fa"), os.path.join(d, "test-1.vcf"), "1")
	cuts0, _, score0 = h.find_founders(2, 0)
	cuts_far, assigned, score_far = h.find_founders(3, 100)            # no two cuts can be 100 aligned positions apart: one block
	assert cuts_far == [0, len(h.reference_positions) - 1]
	assert score_far >= score0
	assert len(assigned) == 3 and len(set(assigned)) == 3             # the three largest path classes of the single block
	for k in (1, 2, 5, 14):
		cuts, assigned, _ = h.find_founders(k, 0)
		assert cuts == cuts0 and len(assigned) == (len(cuts) - 1) * k


# The host's pBWT skips work the reference does literally (divergence counts touched only on a change, 32-bit biased
# divergence values, a flat count table) and, with more than one thread, rebuilds the pBWT state from scratch at chunk
# boundaries instead of stepping to it; the oracle's restatement makes every update of the reference.  Same answers on
# random inputs.
@pytest.mark.parametrize("seed,ref_len,n_variants,n_samples,kw", [
	(1, 3000, 120, 6, dict()),
	(2, 5000, 400, 9, dict(multi_allelic=0.3)),
	(3, 20000, 900, 40, dict(mix=(0.6, 0.2, 0.2))),
	(4, 8000, 300, 3, dict(density=0.5)),
	(5, 8000, 300, 70, dict(density=0.02)),
	(6, 60000, 500, 12, dict(long_every=50)),
	(7, 2000, 60, 1, dict(ploidy=1)),
	(8, 30000, 2500, 33, dict(multi_allelic=0.1, mix=(0.7, 0.15, 0.15))),
], ids=lambda v: str(v) if isinstance(v, int) else None)
def test_random_inputs_against_the_literal_restatement(HostGraph, tmp_path, seed, ref_len, n_variants, n_samples, kw):
	import numpy as np
	import oracle
	import synth
	rng = np.random.default_rng(1000 + seed)
	ref = synth.random_reference(rng, ref_len)
	recs = synth.random_records(rng, ref, n_variants, n_samples, **kw)
	fa, vcf = synth.write_inputs(str(tmp_path), ref, recs, n_samples)
	og = oracle.build_variant_graph(fa, vcf, "1")
	hg = HostGraph(fa, vcf, "1")
	assert og.edge_count > 0
	# the transpose's result, for the chunked (multi-threaded) matching: here from the oracle, in the product from the GPU
	hg.set_transposed_paths(og.paths_by_chrom_copy_and_edge, og.path_rows, og.path_cols)
	for founders, min_distance, keep in ((1, 0, False), (2, 0, False), (3, 10, True), (7, 50, False), (25, 50, False), (4, 1000, True), (5, 10 * ref_len, False)):
		exp = og.find_founders(founders, min_distance, keep_ref_edges=keep)
		for threads in (1, 2, 7):
			assert hg.find_founders(founders, min_distance, keep_ref_edges=keep, threads=threads) == exp, (founders, min_distance, keep, threads)


@pytest.mark.parametrize("seed,ref_len,n_variants,n_samples,kw", [
	(11, 5000, 400, 9, dict(multi_allelic=0.3)),
	(12, 30000, 2500, 33, dict(multi_allelic=0.1, mix=(0.7, 0.15, 0.15))),
	(13, 120000, 12000, 150, dict(mix=(0.8, 0.1, 0.1))),              # several chunks of thousands of edges, a few hundred copies
], ids=lambda v: str(v) if isinstance(v, int) else None)
def test_walked_searches_with_the_hosts_own_walker(HostGraph, tmp_path, monkeypatch, seed, ref_len, n_variants, n_samples, kw):
	"""find_cut_positions / find_matchings with their chunk walks handed to a founder_walker -- what the GPU path runs around its two
	kernels (tests/test_gpu_founders.py) -- with the walker that walks on the host, edge by edge from the states it is given:
	chunk bounds, the cut search's states reused by the matching, chunks handed back when their output does not fit, a walker that
	refuses the copy count.  Same answers as the sequential loops."""
	import numpy as np
	import oracle
	import synth
	rng = np.random.default_rng(2000 + seed)
	ref = synth.random_reference(rng, ref_len)
	recs = synth.random_records(rng, ref, n_variants, n_samples, **kw)
	fa, vcf = synth.write_inputs(str(tmp_path), ref, recs, n_samples)
	og = oracle.build_variant_graph(fa, vcf, "1")
	hg = HostGraph(fa, vcf, "1")
	hg.set_transposed_paths(og.paths_by_chrom_copy_and_edge, og.path_rows, og.path_cols)
	for founders, min_distance, keep in ((2, 0, False), (3, 10, True), (25, 50, False), (4, 1000, True)):
		want = hg.find_founders(founders, min_distance, keep_ref_edges=keep, threads=1)
		for threads in (1, 3):
			assert hg.find_founders_walked_on_host(founders, min_distance, keep_ref_edges=keep, threads=threads) == want, (founders, min_distance, keep, threads)
			if want is not None and len(want[0]) > 2:
				assert hg.gpu_chunks[0] >= 1 and hg.gpu_chunks[1] == 0 and hg.gpu_chunks[2] >= 1 and hg.gpu_chunks[3] == 0, hg.gpu_chunks
	want = hg.find_founders(5, 20, threads=1)
	monkeypatch.setenv("V2M_FOUNDER_TRIAL_CAPACITY", "50")                    # the walker's output does not fit: the chunks come back
	monkeypatch.setenv("V2M_FOUNDER_POOL_CAPACITY", "40")
	assert hg.find_founders_walked_on_host(5, 20, threads=2) == want
	assert hg.gpu_chunks[1] >= 1 and hg.gpu_chunks[3] >= 1, hg.gpu_chunks
	monkeypatch.delenv("V2M_FOUNDER_TRIAL_CAPACITY")
	monkeypatch.delenv("V2M_FOUNDER_POOL_CAPACITY")
	assert hg.find_founders_walked_on_host(5, 20, threads=2, max_copies=4) == want   # a walker that cannot hold the copies: the plain search
	assert hg.gpu_chunks == (0, 0, 0, 0)


def test_cut_position_file_round_trip(HostGraph, tmp_path):
	"""--output-cut-positions / --input-cut-positions: {min_distance, cut_positions, score} in cereal's portable-binary
	layout (output.hh:133-139); the reference holds no such file, so only the layout stated in founder.hh is checked."""
	import struct
	from vcf2multialign_amd import host
	p = tmp_path / "cuts.bin"
	host.write_cut_positions(p, [0, 3, 17, 2 ** 40 + 5], 50, 7)
	raw = p.read_bytes()
	assert raw == b"\x01" + struct.pack("<IQQ", 0, 50, 4) + struct.pack("<4Q", 0, 3, 17, 2 ** 40 + 5) + struct.pack("<I", 7)
	assert host.read_cut_positions(p) == ([0, 3, 17, 2 ** 40 + 5], 50, 7)
	host.write_cut_positions(p, [], 0, 0)
	assert host.read_cut_positions(p) == ([], 0, 0)
	for bad in (b"", b"\x00" + raw[1:], raw[:-2], raw[:5] + struct.pack("<QQ", 50, 10 ** 15)):
		p.write_bytes(bad)
		with pytest.raises(ValueError):
			host.read_cut_positions(p)
