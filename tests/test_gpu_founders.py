"""The founder cut search with its chunk walks on the GPU (v2m_pbwt_cut_trials: pBWT steps of pbwt.hh:77-134 and the
per-candidate value walk of find_cut_positions.cc:134-165, one workgroup per chunk) against the host's search, the oracle's
literal restatement and the cut positions the reference's own test pins (tests/founder_sequences.cc:130-186)."""

import json
import os

import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def v2m():
	import vcf2multialign_amd as v
	return v


@pytest.fixture(scope="module")
def HostGraph():
	from vcf2multialign_amd.host import HostGraph
	return HostGraph


def _gpu_cuts(v2m, HostGraph, og, fa, vcf, min_distance, threads=3):
	hg = HostGraph(fa, vcf, "1")
	hg.set_transposed_paths(og.paths_by_chrom_copy_and_edge, og.path_rows, og.path_cols)
	with v2m.Context(0) as ctx:
		ctx.upload_graph(v2m.VariantGraph.from_object(og), og.ref)           # with the transposed path matrix
		return hg.find_cut_positions_gpu(ctx, min_distance, threads), hg


def _cases():
	with open(os.path.join(HERE, "golden", "reference_goldens.json")) as f:
		return json.load(f)["founder_sequences"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["vcf"] + "+" + c["fasta"])
def test_reference_cut_positions(v2m, HostGraph, case, fixtures_dir):
	d = os.path.join(fixtures_dir, "founder-sequences")
	fa, vcf = os.path.join(d, case["fasta"]), os.path.join(d, case["vcf"])
	og = oracle.build_variant_graph(fa, vcf, case["chromosome"])
	(cuts, score), _ = _gpu_cuts(v2m, HostGraph, og, fa, vcf, case["minimum_distance"])
	assert cuts == case["cut_positions"]                               # REQUIRE(expected_cut_positions == output.cut_positions())
	# ... and the matchings (v2m_pbwt_cut_records + the host's greedy assignment)
	hg = HostGraph(fa, vcf, case["chromosome"])
	hg.set_transposed_paths(og.paths_by_chrom_copy_and_edge, og.path_rows, og.path_cols)
	with v2m.Context(0) as ctx:
		ctx.upload_graph(v2m.VariantGraph.from_object(og), og.ref)
		cuts, assigned, _ = hg.find_founders_gpu(ctx, case["founder_count"], case["minimum_distance"])
	assert cuts == case["cut_positions"]
	assert assigned == case["assigned_samples_column_major"]           # REQUIRE(expected_matchings == output.assigned_samples())


@pytest.mark.parametrize("seed,ref_len,n_variants,n_samples,kw", [
	(1, 3000, 120, 6, dict()),
	(2, 5000, 400, 9, dict(multi_allelic=0.3)),
	(3, 20000, 900, 40, dict(mix=(0.6, 0.2, 0.2))),
	(4, 8000, 300, 3, dict(density=0.5)),
	(5, 8000, 300, 70, dict(density=0.02)),
	(6, 60000, 500, 12, dict(long_every=50)),
	(7, 2000, 60, 1, dict(ploidy=1)),
	(8, 30000, 2500, 33, dict(multi_allelic=0.1, mix=(0.7, 0.15, 0.15))),
	(9, 200000, 20000, 600, dict(mix=(0.8, 0.1, 0.1))),                  # 1200 copies, several chunks of thousands of edges
], ids=lambda v: str(v) if isinstance(v, int) else None)
def test_random_inputs(v2m, HostGraph, tmp_path, seed, ref_len, n_variants, n_samples, kw):
	rng = np.random.default_rng(1000 + seed)
	ref = synth.random_reference(rng, ref_len)
	recs = synth.random_records(rng, ref, n_variants, n_samples, **kw)
	fa, vcf = synth.write_inputs(str(tmp_path), ref, recs, n_samples)
	og = oracle.build_variant_graph(fa, vcf, "1")
	assert og.edge_count > 0
	for min_distance in (0, 10, 50, 1000, 10 * ref_len):
		got, hg = _gpu_cuts(v2m, HostGraph, og, fa, vcf, min_distance)
		assert hg.gpu_chunks_walked >= 1 and hg.gpu_chunks_left == 0        # the GPU walked every chunk
		want = hg.find_founders(1, min_distance, threads=1)                 # the host's sequential search (== the oracle's, test_host_founders.py)
		assert (got is None) == (want is None), min_distance
		if want is not None:
			assert got[0] == want[0] and got[1] == want[2], min_distance
	# the matchings: cut search + matching with the chunk walks of both on the GPU == the host's sequential loops
	hg = HostGraph(fa, vcf, "1")
	hg.set_transposed_paths(og.paths_by_chrom_copy_and_edge, og.path_rows, og.path_cols)
	with v2m.Context(0) as ctx:
		ctx.upload_graph(v2m.VariantGraph.from_object(og), og.ref)
		for founders, min_distance, keep in ((1, 0, False), (2, 0, False), (3, 10, True), (7, 50, False), (25, 50, False), (4, 1000, True)):
			want = hg.find_founders(founders, min_distance, keep_ref_edges=keep, threads=1)
			got = hg.find_founders_gpu(ctx, founders, min_distance, keep_ref_edges=keep, threads=3)
			assert got == want, (founders, min_distance, keep)
			if want is not None and len(want[0]) > 2:
				assert hg.gpu_chunks[1] == 0 and hg.gpu_chunks[3] == 0 and hg.gpu_chunks[2] >= 1, hg.gpu_chunks
				# given cut positions (--input-cut-positions): the states are built for the matching alone
				assert hg.find_founders_gpu(ctx, founders, min_distance, keep_ref_edges=keep, threads=2, cut_positions=want[0])[1] == want[1]
	if seed <= 8:                                                            # and the oracle's literal restatement directly
		exp = og.find_founders(2, 50)
		got, _ = _gpu_cuts(v2m, HostGraph, og, fa, vcf, 50)
		assert (exp is None and got is None) or (got[0] == exp[0] and got[1] == exp[2])


def test_chunks_the_gpu_leaves_undone_are_walked_on_the_host(v2m, HostGraph, tmp_path, monkeypatch):
	"""A chunk whose pairs do not fit the buffer it was given comes back marked; the host walks it: same cut positions."""
	rng = np.random.default_rng(77)
	ref = synth.random_reference(rng, 100000)
	recs = synth.random_records(rng, ref, 8000, 150, mix=(0.8, 0.1, 0.1))
	fa, vcf = synth.write_inputs(str(tmp_path), ref, recs, 150)
	og = oracle.build_variant_graph(fa, vcf, "1")
	want, hg = _gpu_cuts(v2m, HostGraph, og, fa, vcf, 20)
	assert hg.gpu_chunks_left == 0 and hg.gpu_chunks_walked > 1
	monkeypatch.setenv("V2M_FOUNDER_TRIAL_CAPACITY", "2000")
	got, hg = _gpu_cuts(v2m, HostGraph, og, fa, vcf, 20)
	assert hg.gpu_chunks_left >= 1
	assert got == want
	monkeypatch.delenv("V2M_FOUNDER_TRIAL_CAPACITY")
	with v2m.Context(0) as ctx:
		ctx.upload_graph(v2m.VariantGraph.from_object(og), og.ref)
		full = hg.find_founders_gpu(ctx, 6, 20, threads=4)
		assert hg.gpu_chunks[3] == 0 and full == hg.find_founders(6, 20, threads=1)
		monkeypatch.setenv("V2M_FOUNDER_POOL_CAPACITY", "300")
		assert hg.find_founders_gpu(ctx, 6, 20, threads=4) == full
		assert hg.gpu_chunks[3] >= 1                                       # some chunks' joined classes did not fit: walked on the host


def test_streamed_and_array_form_of_the_chunk_walks_agree(v2m, HostGraph, tmp_path, monkeypatch):
	"""v2m_pbwt_cut_trials_streamed (pairs through pinned slots and a callback, what the searches use) and v2m_pbwt_cut_trials (pairs
	into the caller's arrays): same cut positions, with several slices in flight and with chunks handed back to the host."""
	rng = np.random.default_rng(91)
	ref = synth.random_reference(rng, 120000)
	recs = synth.random_records(rng, ref, 9000, 180, mix=(0.8, 0.1, 0.1))
	fa, vcf = synth.write_inputs(str(tmp_path), ref, recs, 180)
	og = oracle.build_variant_graph(fa, vcf, "1")
	streamed, hg = _gpu_cuts(v2m, HostGraph, og, fa, vcf, 15)
	assert hg.gpu_chunks_left == 0 and hg.gpu_chunks_walked > 1
	monkeypatch.setenv("V2M_FOUNDER_ARRAYS", "1")
	arrays, hg = _gpu_cuts(v2m, HostGraph, og, fa, vcf, 15)
	assert hg.gpu_chunks_left == 0 and arrays == streamed
	monkeypatch.setenv("V2M_FOUNDER_TRIAL_CAPACITY", "3000")                 # some chunks come back undone, in both forms
	assert _gpu_cuts(v2m, HostGraph, og, fa, vcf, 15)[0] == streamed
	monkeypatch.delenv("V2M_FOUNDER_ARRAYS")
	got, hg = _gpu_cuts(v2m, HostGraph, og, fa, vcf, 15)
	assert hg.gpu_chunks_left >= 1 and got == streamed


def test_refuses_what_it_cannot_hold(v2m):
	"""More chromosome copies than a workgroup's LDS holds, or no path matrix on the device: an error, not a silent fallback."""
	from vcf2multialign_amd import _native as N
	import ctypes as C
	with v2m.Context(0) as ctx:
		a = (C.c_uint64 * 4)(1, 2, 0, 0)
		u = (C.c_uint32 * 4)()
		rc = ctx._lib.v2m_pbwt_cut_trials(ctx._h, 4, 0, 2, u, a, 1, a, u, u, 16, u, u, a, u)
		assert rc == N.V2M_ERR_STATE


def test_refuses_bad_candidates_and_start_states(v2m):
	"""v2m_pbwt_cut_trials has the input discipline of v2m_pbwt_cut_records: candidate edges that do not ascend or leave the graph,
	aligned positions that go back (find_cut_positions.cc:129,151), a start order that is no permutation of the copies (it indexes
	the workgroup's LDS state) and a bound matrix wider than the LDS column stash are errors -- before any kernel sees them."""
	from vcf2multialign_amd import _native as N
	import ctypes as C
	d = os.path.join(HERE, "golden", "reference-fixtures", "founder-sequences")
	og = oracle.build_variant_graph(os.path.join(d, "test-1.fa"), os.path.join(d, "test-1.vcf"), "1")
	n_copies, n_edges = og.total_chromosome_copies, og.edge_count
	assert n_edges >= 3
	u32, u64 = C.c_uint32, C.c_uint64

	def arr(t, values):
		return (t * len(values))(*values)

	with v2m.Context(0) as ctx:
		ctx.upload_graph(v2m.VariantGraph.from_object(og), og.ref)
		cap = 4096
		out32, out32b, out64, status = (u32 * cap)(), (u32 * cap)(), (u64 * 8)(), (u32 * 2)()
		order = arr(u32, list(range(n_copies)))
		div = arr(u32, [0] * n_copies)
		chunk_first = arr(u64, [1, 3])

		def trials(cand_edge, cand_aln, start_order=order, copies=n_copies):
			return ctx._lib.v2m_pbwt_cut_trials(ctx._h, copies, 0, len(cand_edge), arr(u32, cand_edge), arr(u64, cand_aln), 1, chunk_first, start_order, div,
				cap, out32, out32b, out64, status)

		good_edge, good_aln = [0, 1, 2], [0, 5, 9]
		assert trials(good_edge, good_aln) == N.V2M_OK and status[0] in (0, 1)
		for bad_edge in ([0, 2, 1], [1, 0, 2], [0, 1, n_edges + 1]):
			assert trials(bad_edge, good_aln) == N.V2M_ERR_INVALID_ARGUMENT
			assert b"candidate edges" in ctx._lib.v2m_last_error(ctx._h)
		assert trials(good_edge, [0, 9, 5]) == N.V2M_ERR_INVALID_ARGUMENT
		assert b"aligned positions" in ctx._lib.v2m_last_error(ctx._h)
		bad_order = list(range(n_copies)); bad_order[-1] = n_copies
		assert trials(good_edge, good_aln, start_order=arr(u32, bad_order)) == N.V2M_ERR_INVALID_ARGUMENT
		assert b"start_order" in ctx._lib.v2m_last_error(ctx._h)
		# the sibling entry point refuses the same start order, and a first cut edge outside the graph
		cut_edge = arr(u32, [0, 1, 2])
		z32 = (u32 * cap)()
		def records(cut_edges, start_order=order):
			return ctx._lib.v2m_pbwt_cut_records(ctx._h, n_copies, len(cut_edges), arr(u32, cut_edges), 1, arr(u64, [1, len(cut_edges)]), arr(u32, [0]), start_order, div,
				cap, z32, (u32 * cap)(), (u32 * cap)(), (u64 * 8)(), (u32 * 8)(), (u32 * 8)(), (u32 * 8)(), status)
		assert records([0, 1, 2]) == N.V2M_OK
		assert records([0, 1, 2], start_order=arr(u32, bad_order)) == N.V2M_ERR_INVALID_ARGUMENT
		assert records([n_edges + 1, n_edges + 2, n_edges + 3]) == N.V2M_ERR_INVALID_ARGUMENT

		# a bound matrix with more copy columns than the kernels' LDS column stash holds (20480), walked with fewer copies: refused; one that
		# needs the matching kernel's class arrays in device memory (more than 12288 columns): taken
		import torch
		for wide, rc in ((20480 + 64, N.V2M_ERR_UNSUPPORTED), (12288 + 64, N.V2M_OK)):
			words = torch.zeros(og.path_rows // 64 * wide, dtype=torch.int64, device="cuda:0")
			ctx.set_paths_device(words.data_ptr(), og.path_rows, wide)
			assert trials(good_edge, good_aln) == rc
			if rc != N.V2M_OK:
				assert b"columns" in ctx._lib.v2m_last_error(ctx._h)
			assert records([0, 1, 2]) == rc
			del words


def test_edge_major_copy_is_kept_between_the_two_searches_and_dropped_with_the_binding(v2m, HostGraph, tmp_path):
	"""One founder run makes two v2m_pbwt_* calls on one binding: the second reuses the first's edge-major copy of the matrix (one
	transpose launch, not two), and a new binding -- another matrix at the same address included -- makes a new one."""
	from vcf2multialign_amd import _native as N
	rng = np.random.default_rng(5)
	ref = synth.random_reference(rng, 40000)
	recs = synth.random_records(rng, ref, 3000, 40, mix=(0.8, 0.1, 0.1))
	fa, vcf = synth.write_inputs(str(tmp_path), ref, recs, 40)
	og = oracle.build_variant_graph(fa, vcf, "1")
	hg = HostGraph(fa, vcf, "1")
	hg.set_transposed_paths(og.paths_by_chrom_copy_and_edge, og.path_rows, og.path_cols)
	want = hg.find_founders(5, 20, threads=1)
	with v2m.Context(0) as ctx:
		ctx.upload_graph(v2m.VariantGraph.from_object(og), og.ref)
		ctx.profile_enable(True)
		ctx.profile_reset()
		assert hg.find_founders_gpu(ctx, 5, 20, threads=2) == want
		assert ctx.profile_get(N.KERNEL_TRANSPOSE)[0] == 1                 # cut search + matching: one transpose back
		assert hg.find_founders_gpu(ctx, 5, 20, threads=2) == want
		assert ctx.profile_get(N.KERNEL_TRANSPOSE)[0] == 1                 # and none for a second run on the same binding
		# the same graph with every path bit cleared, uploaded again: the copy must not survive the binding
		import copy
		vg = v2m.VariantGraph.from_object(og)
		vg0 = copy.copy(vg)
		vg0.paths_by_chrom_copy_and_edge = np.zeros_like(vg.paths_by_chrom_copy_and_edge)
		ctx.upload_graph(vg0, og.ref)
		hg0 = HostGraph(fa, vcf, "1")
		hg0.set_transposed_paths(vg0.paths_by_chrom_copy_and_edge, og.path_rows, og.path_cols)
		got0 = hg0.find_founders_gpu(ctx, 5, 20, threads=2)
		assert ctx.profile_get(N.KERNEL_TRANSPOSE)[0] == 2
		# no copy follows any ALT edge now: one path class from the first node to the last (the stale copy would give `want` again)
		assert got0 != want and got0[0] == [0, len(hg0.reference_positions) - 1] and got0[2] == 1


@pytest.mark.parametrize("n_samples", [1100, 1600, 2100, 2600, 3100, 4096, 4700, 6100, 8100, 10016], ids=lambda n: "%d_copies" % (2 * n))
def test_every_copies_per_thread_instantiation(v2m, HostGraph, tmp_path, n_samples):
	"""The founder kernels are instantiated per copies-per-thread count (ceil(copies / 1024) = 1 .. 8, then 10, 12, 16 and 20; the other tests
	use 1, 2 and 5): 2200 .. 20 032 copies -- BASELINE config 5's scale; the matching's kernel keeps its class arrays in LDS up to 12 288
	copies and in device memory above -- through both searches on the GPU against the host's sequential loops."""
	rng = np.random.default_rng(4000 + n_samples)
	ref = synth.random_reference(rng, 4000)
	recs = synth.random_records(rng, ref, 160, n_samples, mix=(0.8, 0.1, 0.1), density=0.08)
	fa, vcf = synth.write_inputs(str(tmp_path), ref, recs, n_samples)
	og = oracle.build_variant_graph(fa, vcf, "1")
	assert og.total_chromosome_copies == 2 * n_samples
	hg = HostGraph(fa, vcf, "1")
	hg.set_transposed_paths(og.paths_by_chrom_copy_and_edge, og.path_rows, og.path_cols)
	with v2m.Context(0) as ctx:
		ctx.upload_graph(v2m.VariantGraph.from_object(og), og.ref)
		for founders, min_distance in ((3, 0), (25, 20)):
			want = hg.find_founders(founders, min_distance, threads=1)
			got = hg.find_founders_gpu(ctx, founders, min_distance, threads=4)
			assert got == want, (founders, min_distance)
			if want is not None and len(want[0]) > 2:
				assert hg.gpu_chunks[1] == 0 and hg.gpu_chunks[0] >= 1, hg.gpu_chunks            # the cut search: every chunk walked on the GPU
				assert hg.gpu_chunks[3] == 0 and hg.gpu_chunks[2] >= 1, hg.gpu_chunks            # the matching too, at every size
