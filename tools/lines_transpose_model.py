#!/usr/bin/env python3
"""CPU model of transpose_bits_lines_kernel's indexing (vcf2multialign_amd/csrc/kernels.hpp): the carried block, the slab
layout, where a destination column's lines begin (s_r), the spans that stream forwards and backwards, the store guards
at the span ends and the merged column ends (kMerge) -- everything except the in-register 64x64 tile transpose itself,
which is taken as given.  Destination words are modelled symbolically as (column, word of the column): the model checks,
on small matrices of awkward shapes, that every destination word is written with the right value, that nothing outside
the matrix is written, and that with merged column ends every word is written exactly once.  Runs on the CPU in
seconds; it exists because there is no GPU in the build container and an indexing mistake costs a GPU round trip."""
import sys

U32 = 1 << 32


def lines_transpose(SW, DW, DP, span_blocks, merge, kTsR=8, kWaves=8, kSlabRows=32):
	"""Returns {flat destination word index: [values written]} with values = (destination column, word of it)."""
	kA = kTsR // kWaves
	P, NB = (SW + kTsR - 1) // kTsR, (DW + 15) // 16
	NS = (NB + span_blocks - 1) // span_blocks
	merge = merge and NS == 1 and DP == DW and DW >= 16 and DP % 16 != 0      # launch_transpose_lines (without its NB <= 8: a speed matter)
	writes = {}
	pitch_lo = DP & 15
	for panel in range(P):
		rw0 = panel * kTsR
		for span in range(NS):
			b_lo = span * span_blocks
			b_hi = min(b_lo + span_blocks, NB)
			span_lo, span_hi = 16 * b_lo, min(16 * b_hi, DW)
			reverse = bool(span & 1)
			slab_cur = 0 if reverse else 16
			slab_prev = 16 - slab_cur
			n_span_blocks = b_hi - b_lo
			for wave in range(kWaves):
				for a in range(kA):
					rw = rw0 + kA * wave + a
					rw_ok = rw < SW
					# y[lane][c]: (column, word) or None where the kernel holds garbage (clamped loads, nothing computed yet)
					y_prev = [[None] * 16 for _ in range(64)]
					y_cur = [[None] * 16 for _ in range(64)]
					y_first = None
					for i in range(n_span_blocks + 1):
						if i < n_span_blocks:
							b = (b_hi - 1 - i) if reverse else (b_lo + i)
							for lane in range(64):
								for c in range(16):
									y_cur[lane][c] = (rw * 64 + lane, 16 * b + c) if (16 * b + c < DW and rw_ok) else None
						if merge and i == 0:
							y_first = [row[:] for row in y_cur]
						e = ((b_hi - 1 - i) if reverse else (b_lo + i - 1)) % U32
						head_round = merge and (e + 1) % U32 == 0
						overlay = merge and (e + 2) % U32 >= NB
						overlay_at = (DW - 16 * e) % U32
						for part in range(64 // kSlabRows):
							if head_round and part > 0:
								continue
							slab = [[None] * 33 for _ in range(kSlabRows)]
							for lane in range(64):
								if lane // kSlabRows == part:
									for c in range(16):
										slab[lane % kSlabRows][slab_prev + c] = y_prev[lane][c]
										slab[lane % kSlabRows][slab_cur + c] = y_cur[lane][c]
							if overlay:
								for lane in range(1, 64):
									if (lane - 1) // kSlabRows == part:
										for c in range(16):
											at = (overlay_at + c) % U32
											slab[(lane - 1) % kSlabRows][at if at < 32 else 32] = y_first[lane][c]
							for lane in range(64):
								lane_col, lane_word = lane >> 4, lane & 15
								for k in range(kSlabRows // 4):
									col = lane_col + 4 * k
									s_word = ((0 - (lane_col + 4 * (k & 3)) * pitch_lo) & 15) + lane_word
									v = slab[col][s_word]
									c_rel = (16 * e + s_word) % U32
									if merge:
										begins = (c_rel - lane_word) % U32
										first_col = part == 0 and k == 0 and lane_col == 0
										last_col = part == 64 // kSlabRows - 1 and k == kSlabRows // 4 - 1 and lane_col == 3
										own = c_rel < DW
										ok = (begins < DW or first_col) if own else (begins < DW and not last_col)
									else:
										ok = span_lo <= c_rel < span_hi
									if rw_ok and ok:
										flat = (rw * 64 + kSlabRows * part + 4 * k + lane_col) * DP + c_rel
										writes.setdefault(flat, []).append(v)
						y_prev = [row[:] for row in y_cur]
	return writes, merge


def check(SW, DW, DP, span_blocks, merge, **geometry):
	writes, merged = lines_transpose(SW, DW, DP, span_blocks, merge, **geometry)
	n_cols = SW * 64
	expect = {}
	for r in range(n_cols):
		for w in range(DW):
			expect[r * DP + w] = (r, w)
	for flat, values in writes.items():
		assert flat in expect, "word %d outside the matrix written (SW %d DW %d DP %d span %d)" % (flat, SW, DW, DP, span_blocks)
		for v in values:
			assert v == expect[flat], "word %d: wrote %r, want %r (SW %d DW %d DP %d span %d merge %s)" % (flat, v, expect[flat], SW, DW, DP, span_blocks, merged)
		assert len(values) == 1, "word %d written %d times (SW %d DW %d DP %d span %d merge %s)" % (flat, len(values), SW, DW, DP, span_blocks, merged)
	missing = [f for f in expect if f not in writes]
	assert not missing, "%d words never written, first %d (SW %d DW %d DP %d span %d merge %s)" % (len(missing), missing[0], SW, DW, DP, span_blocks, merged)
	return merged


if __name__ == "__main__":
	n = n_merged = 0
	for SW, DW in ((1, 1), (1, 2), (2, 17), (9, 16), (3, 33), (8, 79), (10, 15), (1, 40), (17, 31), (2, 49), (5, 100)):
		for DP in (DW, (DW + 15) // 16 * 16, DW + 3):
			for span_blocks in (1, 2, 3, 8):
				check(SW, DW, DP, span_blocks, False); n += 1
			n_merged += check(SW, DW, DP, 64, True); n += 1
	for geometry in (dict(kTsR=8, kWaves=4), dict(kTsR=16, kWaves=16, kSlabRows=16), dict(kTsR=8, kWaves=8, kSlabRows=16)):
		for SW, DW, span_blocks in ((3, 33, 2), (9, 79, 64), (2, 17, 1)):
			n_merged += check(SW, DW, DW, span_blocks, True, **geometry); n += 1
	assert n_merged >= 8
	print("%d cases ok (%d with merged column ends)" % (n, n_merged))
	sys.exit(0)
