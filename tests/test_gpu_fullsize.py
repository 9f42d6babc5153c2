"""Full-size runs of BASELINE.json's shapes on the GPU, checked through size-independent properties plus the CPU
oracle on sampled rows (the oracle needs ~0.1 s per 10 MB row; the GPU writes 20 GB in a few ms).

config 2: synthetic 10 Mb reference, 100k SNV-only records, 1000 diploid samples -> all 2001 rows (20 GB).
config 3: synthetic 100 Mb reference, 1M SNV+indel records, 2504 diploid samples -> the full 5056 x 1M path-matrix
          transpose; a 96-row window of the splice (9.6 GB) with padding copies; and EVERY one of the 5009 rows, aligned and
          --unaligned, in the bench's own 64-GB batches, by device checksum against the oracle's streamed checksum of the
          same row (502 GB per mode; ~25 s each on the box's 16 quota cores).
config 4: the config-3 input with --founder-sequences=25 --minimum-distance=50 -> host search on 1 and 16 threads, the
          26 rows (672 495 copy switches per founder row) against the oracle's walk with the same cuts.
config 5: synthetic 250 Mb reference, 6M records incl. MNPs / multi-allelic sites, 10000 diploid samples -> the full
          20032 x 6.24M transpose (15.6 GB), and a 64-row window of the splice (16 GB) holding REF, the last real copies
          and padding copies."""

import numpy as np
import pytest

import full_parity
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
	import torch
	import vcf2multialign_amd as v2m
	from vcf2multialign_amd import synth
	return torch, v2m, synth


def _device_paths(torch, v2m, ds, ctx, copy_base=0, hp=None):
	"""paths_by_edge_and_chrom_copy generated in HBM, transposed on the GPU.  Returns (src, dst) int64 tensors."""
	hp = ds.path_cols if hp is None else hp
	dev = torch.device("cuda", 0)
	thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
	src = torch.empty(ds.path_rows // 64 * hp, dtype=torch.int64, device=dev)
	dst = torch.empty_like(src)
	torch.cuda.synchronize()
	ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), copy_base, hp)
	ctx.transpose_bits_device(src.data_ptr(), hp, ds.path_rows, dst.data_ptr())
	ctx.synchronize()
	return src, dst


_oracle_for = full_parity.oracle_for


def _popcount(torch, t):
	"""Number of set bits of an int64 device tensor (SWAR per word, in 1-GiB pieces)."""
	total = 0
	for c in t.split(1 << 27):
		x = c.clone()
		x -= (x >> 1) & 0x5555555555555555
		x = (x & 0x3333333333333333) + ((x >> 2) & 0x3333333333333333)
		x = (x + (x >> 4)) & 0x0F0F0F0F0F0F0F0F
		total += int((((x * 0x0101010101010101) >> 56) & 0xFF).sum().item())
	return total


def test_popcount_helper(env):
	torch, _, _ = env
	t = torch.tensor([0, 1, -1, 0x7FFFFFFFFFFFFFFF, -0x8000000000000000, 0x0123456789ABCDEF], dtype=torch.int64, device="cuda")
	assert _popcount(torch, t) == 0 + 1 + 64 + 63 + 1 + 32


def test_config3_transpose_full_size(env):
	torch, v2m, synth = env
	ds = synth.dataset("config3")
	with v2m.Context(0) as ctx:
		src, dst = _device_paths(torch, v2m, ds, ctx)
		hp, ep = ds.path_cols, ds.path_rows
		assert (hp, ep) == (5056, 1000000)
		# involution, word for word, on the device
		back = torch.empty_like(src)
		ctx.transpose_bits_device(dst.data_ptr(), ep, hp, back.data_ptr())
		ctx.synchronize()
		assert torch.equal(back, src)
		# the number of set bits survives
		assert _popcount(torch, src) == _popcount(torch, dst)
		# sampled destination columns (= chromosome copies) against the CPU re-derivation of the genotype hash
		wpc = ep // 64
		for copy in (0, 1, 63, 64, 2503, 5007):
			got = dst[copy * wpc:(copy + 1) * wpc].cpu().numpy().view(np.uint64)
			assert np.array_equal(got, ds.copy_column(copy)), "copy %d" % copy
		# padding columns (copies >= H) are empty
		assert int(dst[5008 * wpc:].abs().sum().item()) == 0


def test_config2_all_rows(env):
	torch, v2m, synth = env
	ds = synth.dataset("config2")
	g = ds.graph
	L = g.aligned_length
	assert L == 10_000_000 and ds.n_copies == 2000     # SNV only: no alignment gaps
	with v2m.Context(0) as ctx:
		ctx.upload_graph(g, ds.reference)
		src, dst = _device_paths(torch, v2m, ds, ctx)
		ctx.set_paths_device(dst.data_ptr(), ds.path_rows, ds.path_cols)
		rows = [v2m.PLOIDY_MAX] + list(range(ds.n_copies))
		pitch = ctx.min_row_pitch
		out = torch.empty(len(rows) * pitch, dtype=torch.uint8, device="cuda")
		torch.cuda.synchronize()
		ctx.splice_rows_device(rows, out.data_ptr(), pitch)
		sums = ctx.checksum_rows_device(out.data_ptr(), pitch, len(rows), length=L)

		# REF row == the reference itself (no indels in this config)
		assert sums[0] == v2m.checksum_rows_host([ds.reference])[0]
		assert out[:L].cpu().numpy().tobytes() == ds.reference
		# sampled rows against the oracle, full bytes
		sample = [0, 1, 2, 999, 1000, 1999]
		og = _oracle_for(ds, sample)
		for k, copy in enumerate(sample):
			exp = og.output_sequence(ds.reference, copy_index=k)
			r = 1 + copy
			assert out[r * pitch:r * pitch + L].cpu().numpy().tobytes() == exp, "copy %d" % copy
			assert sums[r] == v2m.checksum_rows_host([exp])[0]
		# every row differs from REF exactly at its set SNV sites: check the count for all rows on the device
		ref_dev = out[:L]
		wpc = ds.path_rows // 64
		for copy in (5, 77, 1234):
			diff = int((out[(1 + copy) * pitch:(1 + copy) * pitch + L] != ref_dev).sum().item())
			bits = int(np.unpackbits(ds.copy_column(copy).view(np.uint8)).sum())
			assert diff == bits    # SNV-only, no overlaps: every set edge changes exactly one base
		# idempotence and batch-split invariance: same rows in three uneven launches into a fresh buffer
		out2 = torch.zeros_like(out)
		torch.cuda.synchronize()
		for lo, hi in ((0, 1), (1, 700), (700, len(rows))):
			ctx.splice_rows_device(rows[lo:hi], out2.data_ptr() + lo * pitch, pitch)
		sums2 = ctx.checksum_rows_device(out2.data_ptr(), pitch, len(rows), length=L)
		assert np.array_equal(sums, sums2)
		# unaligned == aligned here (nothing to remove), lengths all R
		lengths = ctx.splice_rows_device(rows[:300], out2.data_ptr(), (ctx.max_unaligned_length + 255) // 256 * 256, unaligned=True, want_lengths=True)
		assert set(lengths.tolist()) == {L}
		usums = ctx.checksum_rows_device(out2.data_ptr(), (ctx.max_unaligned_length + 255) // 256 * 256, 300, length=L)
		assert np.array_equal(usums, sums[:300])


def test_config3_row_window(env):
	torch, v2m, synth = env
	ds = synth.dataset("config3")
	g = ds.graph
	L = g.aligned_length
	with v2m.Context(0) as ctx:
		ctx.upload_graph(g, ds.reference)
		# this window's copies only: the last two 64-copy words of the matrix (copies 4928..5055, 80 of them real)
		base, hp = 4928, 128
		src, dst = _device_paths(torch, v2m, ds, ctx, copy_base=base, hp=hp)
		ctx.set_paths_device(dst.data_ptr(), ds.path_rows, hp)
		n_real = ds.n_copies - base                 # 80 real copies, the rest of the 128 is padding
		rows = [v2m.PLOIDY_MAX] + list(range(95))   # includes padding copies: they must come out as REF
		pitch = ctx.min_row_pitch
		out = torch.empty(len(rows) * pitch, dtype=torch.uint8, device="cuda")
		torch.cuda.synchronize()
		ctx.splice_rows_device(rows, out.data_ptr(), pitch)
		sums = ctx.checksum_rows_device(out.data_ptr(), pitch, len(rows), length=L)
		sample = [0, 1, 40, n_real - 1]
		og = _oracle_for(ds, [base + c for c in sample])
		ref_row = og.output_sequence(ds.reference)
		assert len(ref_row) == L and sums[0] == v2m.checksum_rows_host([ref_row])[0]
		for k, c in enumerate(sample):
			exp = og.output_sequence(ds.reference, copy_index=k)
			assert sums[1 + c] == v2m.checksum_rows_host([exp])[0], "copy %d" % (base + c)
		assert out[(1 + 40) * pitch:(1 + 40) * pitch + L].cpu().numpy().tobytes() == og.output_sequence(ds.reference, copy_index=2)
		for c in range(n_real, 95):                 # padding copies carry no bits
			assert sums[1 + c] == sums[0]
		# unaligned: row lengths and bytes of the sampled rows
		upitch = (ctx.max_unaligned_length + 255) // 256 * 256
		uout = torch.empty(8 * upitch, dtype=torch.uint8, device="cuda")
		torch.cuda.synchronize()
		urows = [v2m.PLOIDY_MAX] + sample
		lengths = ctx.splice_rows_device(urows, uout.data_ptr(), upitch, unaligned=True, want_lengths=True)
		exp = [og.output_sequence(ds.reference, unaligned=True)] + [og.output_sequence(ds.reference, copy_index=k, unaligned=True) for k in range(len(sample))]
		assert exp[0] == ds.reference
		assert lengths.tolist() == [len(e) for e in exp]
		usums = ctx.checksum_rows_device(uout.data_ptr(), upitch, len(urows), lengths=lengths)
		assert np.array_equal(usums, v2m.checksum_rows_host(exp))


@pytest.mark.parametrize("unaligned", [False, True], ids=["aligned", "unaligned"])
def test_config3_every_row_against_the_oracle(env, unaligned):
	"""BASELINE's headline config, all of it: REF + 5008 haplotype rows of 100.3 Mbases, spliced in the bench's batches (627 rows
	into one 64-GB buffer), every row's length and 64-bit checksum against the oracle's walk of the same row
	(sequence_writer.cc:22-85; --unaligned: :80).  "Bit-exact vs CPU" on the configuration the metric is quoted on is total."""
	torch, v2m, synth = env
	from vcf2multialign_amd.sharding import host_threads_per_rank
	ds = synth.dataset("config3")
	threads = host_threads_per_rank(1)
	with v2m.Context(0) as ctx:
		ctx.upload_graph(ds.graph, ds.reference)
		src, dst = _device_paths(torch, v2m, ds, ctx)
		del src
		ctx.set_paths_device(dst.data_ptr(), ds.path_rows, ds.path_cols)
		out_bytes = 627 * ctx.min_row_pitch
		out_ptr = ctx.alloc_output(out_bytes, candidates=1)
		try:
			res = full_parity.check_all_rows(v2m, ctx, ds, out_ptr, out_bytes, unaligned=unaligned, threads=threads, say=print)
		finally:
			ctx.free_output(out_ptr)
	print("config 3, %s: %d rows in %d batches of %d, GPU %.1f s, oracle %.1f s on %d threads, %d mismatches"
		% ("--unaligned" if unaligned else "aligned", res["rows"], res["batches"], res["rows_per_batch"], res["gpu_s"], res["oracle_s"], threads, len(res["mismatches"])))
	assert res["rows"] == 5009 and res["batches"] == (8 if not unaligned else res["batches"]) and res["rows_per_batch"] <= 627
	assert res["mismatches"] == []
	assert res["distinct_lengths"] == 1 if not unaligned else res["distinct_lengths"] > 1000      # unaligned rows differ in length


def test_config5_every_row_against_the_oracle(env):
	"""BASELINE config 5, all 20 001 rows of 252 Mbases (5.04 TB per mode) the same way.  About 5 min of oracle time per mode on 16
	cores, so it runs only when asked for (V2M_FULL_CONFIG5=1; add "unaligned" to the value for that mode too); the builder runs it
	once per round and records the output under profiles/rNN/."""
	import os
	mode = os.environ.get("V2M_FULL_CONFIG5", "")
	if not mode:
		pytest.skip("set V2M_FULL_CONFIG5=1 (or =unaligned, =both) to walk all 20 001 rows of config 5 on the CPU")
	torch, v2m, synth = env
	from vcf2multialign_amd.sharding import host_threads_per_rank
	ds = synth.dataset("config5")
	threads = host_threads_per_rank(1)
	with v2m.Context(0) as ctx:
		ctx.upload_graph(ds.graph, ds.reference)
		src, dst = _device_paths(torch, v2m, ds, ctx)
		del src
		ctx.set_paths_device(dst.data_ptr(), ds.path_rows, ds.path_cols)
		out_bytes = 253 * ctx.min_row_pitch
		out_ptr = ctx.alloc_output(out_bytes, candidates=1)
		try:
			for unaligned in ([False] if mode == "1" else [True] if mode == "unaligned" else [False, True]):
				res = full_parity.check_all_rows(v2m, ctx, ds, out_ptr, out_bytes, unaligned=unaligned, threads=threads, say=print)
				print("config 5, %s: %d rows in %d batches of %d, GPU %.1f s, oracle %.1f s on %d threads, %d mismatches"
					% ("--unaligned" if unaligned else "aligned", res["rows"], res["batches"], res["rows_per_batch"], res["gpu_s"], res["oracle_s"], threads, len(res["mismatches"])))
				assert res["rows"] == 20001 and res["mismatches"] == []
		finally:
			ctx.free_output(out_ptr)


@pytest.mark.parametrize("config", ["mini5", "mini3"])
def test_small_configs_all_rows_all_bytes(env, config):
	"""The variant mix of config 5 (MNPs, multi-allelic sites, indels under deletions) and of config 3 at a size where
	every row of the synthetic generator's output can be compared byte for byte with the oracle, aligned and unaligned."""
	torch, v2m, synth = env
	ds = synth.dataset(config)
	g = ds.graph
	L = g.aligned_length
	with v2m.Context(0) as ctx:
		ctx.upload_graph(g, ds.reference)
		src, dst = _device_paths(torch, v2m, ds, ctx)
		ctx.set_paths_device(dst.data_ptr(), ds.path_rows, ds.path_cols)
		copies = list(range(ds.n_copies))
		og = _oracle_for(ds, copies)
		rows = [v2m.PLOIDY_MAX] + copies
		exp = [og.output_sequence(ds.reference)] + [og.output_sequence(ds.reference, copy_index=c) for c in copies]
		assert len({len(e) for e in exp}) == 1 and len(exp[0]) == L
		assert len(set(exp)) > len(exp) // 2                      # the rows really differ
		pitch = ctx.min_row_pitch
		out = torch.empty(len(rows) * pitch, dtype=torch.uint8, device="cuda")
		torch.cuda.synchronize()
		ctx.splice_rows_device(rows, out.data_ptr(), pitch)
		host = out.cpu().numpy()
		for r, e in enumerate(exp):
			assert host[r * pitch:r * pitch + L].tobytes() == e, "row %d" % r
		uexp = [og.output_sequence(ds.reference, unaligned=True)] + [og.output_sequence(ds.reference, copy_index=c, unaligned=True) for c in copies]
		upitch = (ctx.max_unaligned_length + 255) // 256 * 256
		uout = torch.empty(len(rows) * upitch, dtype=torch.uint8, device="cuda")
		torch.cuda.synchronize()
		lengths = ctx.splice_rows_device(rows, uout.data_ptr(), upitch, unaligned=True, want_lengths=True)
		assert lengths.tolist() == [len(e) for e in uexp]
		host = uout.cpu().numpy()
		for r, e in enumerate(uexp):
			assert host[r * upitch:r * upitch + len(e)].tobytes() == e, "unaligned row %d" % r
		if config == "mini5":
			labels = [bytes(g.label_bytes[int(a):int(b)]) for a, b in zip(g.label_offsets[:-1], g.label_offsets[1:])]
			assert any(len(l) in (2, 3, 4) for l in labels)      # MNPs are in the mix
			assert (np.diff(g.alt_edge_count_csum.astype(np.int64)) > 1).any()   # and multi-allelic sites


def test_config5_transpose_full_size(env):
	"""BASELINE config 5's whole path matrix: 20032 copies x 6.24 M edges = 15.6 GB, transposed there and back."""
	torch, v2m, synth = env
	ds = synth.dataset("config5")
	hp, ep = ds.path_cols, ds.path_rows
	assert hp == 20032 and ds.n_copies == 20000 and ep % 64 == 0 and ep >= ds.graph.edge_count > 6_000_000
	with v2m.Context(0) as ctx:
		src, dst = _device_paths(torch, v2m, ds, ctx)
		back = torch.empty_like(src)
		ctx.transpose_bits_device(dst.data_ptr(), ep, hp, back.data_ptr())
		ctx.synchronize()
		assert torch.equal(back, src)                                  # involution, word for word
		del back
		n_set = _popcount(torch, src)
		assert n_set == _popcount(torch, dst) and n_set > 0
		wpc = ep // 64
		for copy in (0, 1, 63, 64, 9999, 19967, 19968, 19999):           # destination columns = chromosome copies, vs the CPU genotype hash
			got = dst[copy * wpc:(copy + 1) * wpc].cpu().numpy().view(np.uint64)
			assert np.array_equal(got, ds.copy_column(copy)), "copy %d" % copy
		assert int((dst[20000 * wpc:] != 0).sum().item()) == 0         # padding copies are empty
		# an edge's source column against the same hash, bit by bit for a few edges (rows = copies)
		swc = hp // 64
		cols = {c: ds.copy_column(c) for c in (0, 1, 777, 19999)}
		for e in (0, 1, 12345, ds.graph.edge_count - 1):
			col = src[e * swc:(e + 1) * swc].cpu().numpy().view(np.uint64)
			for c, words in cols.items():
				assert (int(col[c // 64]) >> (c % 64)) & 1 == (int(words[e // 64]) >> (e % 64)) & 1, (e, c)


def test_config5_row_window(env):
	"""64 rows x 252 Mbases of config 5 (16 GB): REF, the last 32 real copies and 31 padding copies of the last 64-copy
	word; every row by device checksum against the oracle, four of them byte for byte, unaligned lengths and checksums."""
	torch, v2m, synth = env
	ds = synth.dataset("config5")
	g = ds.graph
	L = g.aligned_length
	assert len(ds.reference) == 250_000_000 and L > 250_000_000
	with v2m.Context(0) as ctx:
		ctx.upload_graph(g, ds.reference)
		base, hp = 19968, 64
		src, dst = _device_paths(torch, v2m, ds, ctx, copy_base=base, hp=hp)
		ctx.set_paths_device(dst.data_ptr(), ds.path_rows, hp)
		n_real = ds.n_copies - base                                    # 32
		rows = [v2m.PLOIDY_MAX] + list(range(63))
		pitch = ctx.min_row_pitch
		out = torch.empty(len(rows) * pitch, dtype=torch.uint8, device="cuda")
		torch.cuda.synchronize()
		ctx.splice_rows_device(rows, out.data_ptr(), pitch)
		sums = ctx.checksum_rows_device(out.data_ptr(), pitch, len(rows), length=L)

		real = list(range(n_real))
		og = _oracle_for(ds, [base + c for c in real])
		exp_sums, exp_len = og.row_checksums(ds.reference, [oracle.PLOIDY_MAX] + real, threads=12)
		assert set(exp_len.tolist()) == {L}
		assert np.array_equal(sums[:1 + n_real], exp_sums)             # REF + every real copy of the window
		assert len(set(sums[1:1 + n_real].tolist())) > n_real // 2     # and they differ from each other
		for c in range(n_real, 63):                                   # padding copies carry no bits: REF
			assert sums[1 + c] == sums[0]
		for c in (0, 13, n_real - 1):                                 # byte for byte (the last real copy of the data set among them)
			exp = og.output_sequence(ds.reference, copy_index=c)
			assert out[(1 + c) * pitch:(1 + c) * pitch + L].cpu().numpy().tobytes() == exp, "copy %d" % (base + c)
		assert out[:L].cpu().numpy().tobytes() == og.output_sequence(ds.reference)
		del out

		# unaligned: lengths and checksums of REF + 5 copies
		upitch = (ctx.max_unaligned_length + 255) // 256 * 256
		usample = [0, 1, 13, 30, n_real - 1]
		uout = torch.empty((1 + len(usample)) * upitch, dtype=torch.uint8, device="cuda")
		torch.cuda.synchronize()
		lengths = ctx.splice_rows_device([v2m.PLOIDY_MAX] + usample, uout.data_ptr(), upitch, unaligned=True, want_lengths=True)
		exp_usums, exp_ulen = og.row_checksums(ds.reference, [oracle.PLOIDY_MAX] + usample, unaligned=True, threads=6)
		assert lengths.tolist() == exp_ulen.tolist() and lengths[0] == len(ds.reference)
		usums = ctx.checksum_rows_device(uout.data_ptr(), upitch, len(lengths), lengths=lengths)
		assert np.array_equal(usums, exp_usums)


def test_config4_full_size(env):
	"""BASELINE config 4: the config-3 input, --founder-sequences=25 --minimum-distance=50.  transpose on the GPU, cut
	positions + greedy matching on the host (1 thread == 16 threads), REF + 25 founder rows on the GPU against the oracle's
	walk (founder_sequence_greedy_output.cc:106-114 over sequence_writer.cc:22-85) with the same cuts."""
	import zlib
	torch, v2m, synth = env
	from vcf2multialign_amd.host import HostGraph
	founders, min_dist = 25, 50
	ds = synth.dataset("config3")
	g = ds.graph
	L = g.aligned_length
	with v2m.Context(0) as ctx:
		ctx.upload_graph(g, ds.reference)
		src, dst = _device_paths(torch, v2m, ds, ctx)
		hp, ep = ds.path_cols, ds.path_rows
		ctx.set_paths_device(dst.data_ptr(), ep, hp)
		src_host, dst_host = src.cpu().numpy().view(np.uint64), dst.cpu().numpy().view(np.uint64)
		hg = HostGraph.from_arrays(g, src_host, hp, ep, ds.samples, ds.ploidy)
		hg.set_transposed_paths(dst_host, ep, hp)
		res = hg.find_founders(founders, min_dist, keep_ref_edges=False, threads=16)
		assert res is not None
		cuts, assigned, score = res
		n_seg = len(cuts) - 1
		assert cuts[0] == 0 and cuts[-1] == g.node_count - 1 and n_seg > 100_000 and len(assigned) == n_seg * founders
		aln = g.aligned_positions
		assert int(np.diff(aln[np.asarray(cuts, dtype=np.int64)].astype(np.int64)).min()) >= min_dist      # --minimum-distance
		res1 = hg.find_founders(founders, min_dist, keep_ref_edges=False, threads=1)
		crc = lambda r: (zlib.crc32(np.asarray(r[0], dtype=np.uint64).tobytes()), zlib.crc32(np.asarray(r[1], dtype=np.uint32).tobytes()), r[2])
		assert crc(res1) == crc(res)                                  # the chunked multi-thread search == the sequential loop
		# ... == the search with its chunk walks on the GPU (v2m_pbwt_cut_trials): every chunk walked there, same cuts, same score
		import time
		t0 = time.time()
		gpu_cuts, gpu_score = hg.find_cut_positions_gpu(ctx, min_dist, threads=16)
		print("config 4 cut search with GPU chunk walks: %.2f s, %d chunks on the GPU, %d on the host" % (time.time() - t0, hg.gpu_chunks_walked, hg.gpu_chunks_left))
		assert hg.gpu_chunks_walked > 100 and hg.gpu_chunks_left == 0
		assert zlib.crc32(np.asarray(gpu_cuts, dtype=np.uint64).tobytes()) == crc(res)[0] and gpu_score == score
		# ... and both searches that way (v2m_pbwt_cut_records for the matching): same cuts, same 672 495 x 25 matchings, same score
		t0 = time.time()
		res_gpu = hg.find_founders_gpu(ctx, founders, min_dist, keep_ref_edges=False, threads=16)
		print("config 4 cut search + matching with GPU chunk walks: %.2f s, chunks (search GPU, host; matching GPU, host) = %s" % (time.time() - t0, hg.gpu_chunks))
		assert hg.gpu_chunks[1] == 0 and hg.gpu_chunks[3] == 0 and hg.gpu_chunks[2] > 100
		assert crc(res_gpu) == crc(res)

		batch_rows = [v2m.PLOIDY_MAX] + [list(zip(cuts[:-1], assigned[f * n_seg:(f + 1) * n_seg])) for f in range(founders)]
		pitch = ctx.min_row_pitch
		out = torch.empty(len(batch_rows) * pitch, dtype=torch.uint8, device="cuda")
		torch.cuda.synchronize()
		ctx.splice_rows_device(v2m.RowBatch(batch_rows), out.data_ptr(), pitch)
		sums = ctx.checksum_rows_device(out.data_ptr(), pitch, len(batch_rows), length=L)
		assert len(set(sums.tolist())) == len(batch_rows)             # 26 different rows

		og = oracle.graph_from_arrays(g.reference_positions, g.aligned_positions, g.alt_edge_targets, g.alt_edge_count_csum,
			g.label_offsets, g.label_bytes, dst_host, ep, hp)
		check = [0, 1, 7, founders // 2, founders]
		exp_sums, exp_len = og.row_checksums(ds.reference, [batch_rows[r] for r in check], threads=len(check))
		assert set(exp_len.tolist()) == {L}
		assert np.array_equal(sums[check], exp_sums)
		r = founders
		assert out[r * pitch:r * pitch + L].cpu().numpy().tobytes() == og.output_sequence(ds.reference, cuts=batch_rows[r])


def test_config3_path_slices_full_size(env):
	"""SURVEY 8(e) at config 3's full size, through the product entry: the whole host-resident transpose input (5056 copies x
	1 M edges, 632 MB) is offered to each of 8 'GPUs' (contexts on this one), each takes only its own copies
	(v2m_upload_path_slice: packed slice, transposed there, line-aligned copy bound) and splices rows of them; sampled rows
	of three ranks against the oracle, and every rank's shard against the chromosome-copy ranges sharding.py hands out."""
	torch, v2m, synth = env
	from vcf2multialign_amd.sharding import shard_copies
	ds = synth.dataset("config3")
	g = ds.graph
	L = g.aligned_length
	hp, ep = ds.path_cols, ds.path_rows
	with v2m.Context(0) as ctx:
		src, _ = _device_paths(torch, v2m, ds, ctx)
		host_src = src.cpu().numpy().view(np.uint64)                       # paths_by_edge_and_chrom_copy as the builder would hold it
		del src, _
		world = 8
		covered = 0
		for rank in range(world):
			c0, c1, hp_local = shard_copies(ds.n_copies, world, rank)
			assert c0 == covered and c0 % 8 == 0
			covered = c1
			if rank not in (0, 3, 7):
				continue
			ctx.upload_graph(g, ds.reference)
			ctx.upload_path_slice(host_src, hp, ep, c0, c1 - c0)
			local = sorted({0, 1, (c1 - c0) // 2, c1 - c0 - 1})
			rows = ([v2m.PLOIDY_MAX] if rank == 0 else []) + local
			pitch = ctx.min_row_pitch
			out = torch.empty(len(rows) * pitch, dtype=torch.uint8, device="cuda")
			torch.cuda.synchronize()
			ctx.splice_rows_device(rows, out.data_ptr(), pitch)
			sums = ctx.checksum_rows_device(out.data_ptr(), pitch, len(rows), length=L)
			og = _oracle_for(ds, [c0 + c for c in local])
			want, lengths = og.row_checksums(ds.reference, ([oracle.PLOIDY_MAX] if rank == 0 else []) + list(range(len(local))), threads=len(rows))
			assert set(lengths.tolist()) == {L}
			assert np.array_equal(sums, want), (rank, c0, c1)
			with pytest.raises(v2m.V2MError):                               # this context holds its own copies (padded) and no more
				ctx.splice_rows_device([hp_local], out.data_ptr(), pitch)
			del out
		assert covered == ds.n_copies
