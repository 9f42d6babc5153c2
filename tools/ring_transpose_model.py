#!/usr/bin/env python3
"""CPU model of transpose_bits_ring_kernel's indexing (vcf2multialign_amd/csrc/kernels.hpp): the per-row ring slots, the
"which rows completed a sector" decode from the ballot, the span windows and the partial sectors at row ends -- everything
except the in-register 64x64 tile transpose itself, which is taken as given.  Checks, on small matrices of awkward shapes,
that every destination word is written exactly once with the right value.  Runs on the CPU in seconds; it exists because
there is no GPU in the build container and an indexing mistake costs a GPU round trip."""
import itertools
import sys

import numpy as np


def tile_transpose(words):
	"""words[c] = source column c of a 64x64-bit tile (bit r = row r) -> out[r] = bit c is source (r, c)."""
	bits = ((words[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint64)   # [c][r]
	return (bits.T << np.arange(64, dtype=np.uint64)[None, :]).sum(axis=1, dtype=np.uint64)


def ring_transpose(src, SW, DW, R, W, S, K):
	A = R // W
	dst = np.zeros(SW * 64 * DW, dtype=np.uint64)
	written = np.zeros(dst.size, dtype=np.int32)
	K = (K + S - 1) // S * S
	P, NS = (SW + R - 1) // R, (DW + K - 1) // K
	for panel, span in itertools.product(range(P), range(NS)):
		rw0 = panel * R
		span_begin, span_end = span * K, min(DW, span * K + K)
		first_span, last_span = span == 0, span_end == DW
		cg_lo = 0 if first_span else span_begin - (S - 1)
		ring = np.zeros((W, A, 64, S), dtype=np.uint64)
		for cg in range(cg_lo, span_end):
			for wave in range(W):
				for a in range(A):
					rw = rw0 + A * wave + a
					if rw >= SW:
						continue
					tile = np.array([src[(cg * 64 + c) * SW + rw] for c in range(64)], dtype=np.uint64)
					tv = tile_transpose(tile)
					tile_base = rw * 64 * DW
					lanes = np.arange(64)
					slot = (tile_base + lanes * DW + cg) & (S - 1)
					ring[wave, a, lanes, slot] = tv
					done = (slot == S - 1) | (cg + 1 == DW)
					if not done.any():
						continue
					n = int(done.sum())
					j0 = int(np.argmax(done))
					assert n & (n - 1) == 0, n
					stride = 64 // n
					assert np.array_equal(np.nonzero(done)[0], j0 + stride * np.arange(n)), "not an arithmetic progression"
					per = 64 // S
					f = 0
					while f * per < n:
						for lane in range(64):
							idx = f * per + (lane >> (S.bit_length() - 1))
							w = lane & (S - 1)
							if idx >= n:
								continue
							row = j0 + idx * stride
							row_base = tile_base + row * DW
							g = row_base + cg
							sl = g & (S - 1)
							g0 = g - sl
							ok = w <= sl and g0 + w >= row_base
							if not first_span:
								ok = ok and g0 >= ((row_base + span_begin) & ~(S - 1))
							if not last_span:
								ok = ok and g0 < ((row_base + span_end) & ~(S - 1))
							if ok:
								dst[g0 + w] = ring[wave, a, row, w]
								written[g0 + w] += 1
						f += 1
	return dst, written


def reference(src, SW, DW):
	n_rows, n_cols = SW * 64, DW * 64
	bits = np.zeros((n_rows, n_cols), dtype=np.uint8)
	for c in range(n_cols):
		col = src[c * SW:(c + 1) * SW]
		bits[:, c] = np.unpackbits(col.view(np.uint8), bitorder="little")
	out = np.zeros(n_rows * DW, dtype=np.uint64)
	for r in range(n_rows):
		out[r * DW:(r + 1) * DW] = np.packbits(bits[r], bitorder="little").view(np.uint64)
	return out


def main():
	rng = np.random.default_rng(7)
	cases = 0
	for SW, DW in [(1, 1), (1, 9), (3, 7), (2, 8), (2, 16), (17, 5), (3, 79), (20, 33), (2, 130), (16, 24), (5, 12)]:
		src = rng.integers(0, 2**63, size=SW * DW * 64, dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, size=SW * DW * 64).astype(np.uint64)
		want = reference(src, SW, DW)
		for R, W, S, K in [(16, 8, 8, 16), (16, 4, 8, 64), (8, 4, 4, 8), (16, 8, 16, 32), (8, 8, 8, 24), (16, 16, 8, 8), (16, 8, 2, 4)]:
			got, written = ring_transpose(src, SW, DW, R, W, S, K)
			assert (written == 1).all(), ("coverage", SW, DW, R, W, S, K, np.unique(written, return_counts=True))
			assert np.array_equal(got, want), ("value", SW, DW, R, W, S, K)
			cases += 1
	print("ring transpose model: %d cases ok" % cases)


if __name__ == "__main__":
	sys.exit(main())
