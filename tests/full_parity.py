"""EVERY row of a full-size synthetic config against the CPU oracle, without moving a row off the device.

The GPU side splices the rows in the bench's own batches (as many rows as a ~64-GB output buffer holds) through
v2m_splice_rows_device and reduces each row to v2m_checksum_rows_device's 64-bit checksum (+ its length in --unaligned
mode); the oracle side walks the same rows (sequence_writer.cc:22-85 restated, oracle/v2m_oracle.cc) and computes the
same checksum while it writes, on as many threads as the job's CPU quota allows.  The oracle's path-matrix columns are
the CPU re-derivation of the genotype hash (synth.Dataset.copy_column), built batch by batch, so no more than one batch
of columns is ever resident on the host.

TEST INFRASTRUCTURE (imports the oracle): used by tests/test_gpu_fullsize.py; bench.py has its own copy of the idea in
its parity leg.  Config 3 (5009 rows of 100.3 Mbases) costs about 25 s per mode on 16 threads."""

import time

import numpy as np

import oracle


def oracle_for(ds, copies):
	"""Oracle graph whose path matrix holds the CPU re-derivation of the given chromosome copies (as copies 0..k-1)."""
	g = ds.graph
	pad = (-len(copies)) % 64
	cols = np.concatenate([ds.copy_column(c) for c in copies] + [np.zeros(ds.path_rows // 64, np.uint64)] * pad) if len(copies) + pad else np.zeros(0, np.uint64)
	return oracle.graph_from_arrays(g.reference_positions, g.aligned_positions, g.alt_edge_targets, g.alt_edge_count_csum,
		g.label_offsets, g.label_bytes, cols, ds.path_rows, len(copies) + pad)


def check_all_rows(v2m, ctx, ds, out_ptr, out_bytes, unaligned=False, threads=8, first_copy=0, n_copies=None, include_reference=True, say=None):
	"""The ctx holds ds's graph and a path matrix whose column j is chromosome copy first_copy + j.  Splices REF (optionally)
	and copies [first_copy, first_copy + n_copies) in batches that fill [out_ptr, out_ptr + out_bytes) and compares every row's
	checksum and length with the oracle's.  Returns {"rows", "batches", "mismatches": [(global row, why)], "gpu_s", "oracle_s"}."""
	L = ds.graph.aligned_length
	n_copies = ds.n_copies - first_copy if n_copies is None else n_copies
	pitch = (ctx.max_unaligned_length + 255) // 256 * 256 if unaligned else ctx.min_row_pitch
	per_batch = max(1, out_bytes // pitch)
	rows = ([v2m.PLOIDY_MAX] if include_reference else []) + list(range(n_copies))     # local copy indices
	n_batches = max(1, -(-len(rows) // per_batch))
	per_batch = -(-len(rows) // n_batches)                                              # equal batches, as bench.py cuts them
	mismatches, gpu_s, oracle_s, lengths_seen = [], 0.0, 0.0, set()
	for b in range(n_batches):
		part = rows[b * per_batch:(b + 1) * per_batch]
		t0 = time.perf_counter()
		lengths = ctx.splice_rows_device(part, out_ptr, pitch, unaligned=unaligned, want_lengths=unaligned)
		if unaligned:
			got = ctx.checksum_rows_device(out_ptr, pitch, len(part), lengths=lengths)
		else:
			got = ctx.checksum_rows_device(out_ptr, pitch, len(part), length=L)
			lengths = np.full(len(part), L, dtype=np.uint64)
		gpu_s += time.perf_counter() - t0
		t0 = time.perf_counter()
		copies = [first_copy + r for r in part if r != v2m.PLOIDY_MAX]
		og = oracle_for(ds, copies)
		col = {c: i for i, c in enumerate(copies)}
		want, want_len = og.row_checksums(ds.reference, [oracle.PLOIDY_MAX if r == v2m.PLOIDY_MAX else col[first_copy + r] for r in part], unaligned=unaligned, threads=threads)
		del og
		oracle_s += time.perf_counter() - t0
		lengths_seen.update(int(x) for x in want_len)
		for i in np.nonzero((got != want) | (np.asarray(lengths, dtype=np.uint64) != want_len))[0]:
			mismatches.append((b * per_batch + int(i), "length %d vs %d" % (lengths[i], want_len[i]) if lengths[i] != want_len[i] else "checksum"))
		if say:
			say("  batch %d/%d: %d rows, %d mismatches so far (GPU %.1f s, oracle %.1f s on %d threads)" % (b + 1, n_batches, len(part), len(mismatches), gpu_s, oracle_s, threads))
	return {"rows": len(rows), "batches": n_batches, "rows_per_batch": per_batch, "mismatches": mismatches, "gpu_s": gpu_s, "oracle_s": oracle_s,
		"distinct_lengths": len(lengths_seen), "unaligned": unaligned}
