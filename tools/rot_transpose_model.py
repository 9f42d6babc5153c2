#!/usr/bin/env python3
"""CPU replay of the rotating-line transpose kernel ("rot8", csrc/kernels.hpp: transpose_bits_rot_kernel): which destination word every
store writes, with which register of the rotating 16-word file -- every word of the destination written exactly once, with the right value,
for dense and padded pitches, odd / even / line-multiple pitches, ragged spans and a destination that does not start on a line.

Model of ONE wave's tile (64 destination columns of one source row-word) per (panel row-word, span); values are the flat source coordinates."""
import itertools, sys


def run(SW, DW, dst_pitch, span_blocks, base_words=0):
	"""Returns dict flat destination word -> (column, column group) written, raising on a word written twice."""
	written = {}
	n_blocks = (DW + 15) // 16
	n_spans = (n_blocks + span_blocks - 1) // span_blocks
	for rw in range(SW):
		for span in range(n_spans):
			b_lo, b_hi = span * span_blocks, min(n_blocks, (span + 1) * span_blocks)
			span_lo, span_hi = 16 * b_lo, min(16 * b_hi, DW)
			y = [[None] * 16 for _ in range(64)]                       # per lane: the rotating register file
			done_at = [(15 - ((base_words + (rw * 64 + l) * dst_pitch) & 15)) & 15 for l in range(64)]
			def emit(c, cg, fast):
				for l in range(64):
					if done_at[l] != c:
						continue
					r = rw * 64 + l
					for j in range(16):
						k = cg - 15 + j                                 # column-relative word of the line's word j
						reg = (c + 1 + j) & 15
						if fast:
							assert span_lo <= k < span_hi, "fast flavour with a word outside the span"
						elif not (span_lo <= k < span_hi):
							continue
						flat = base_words + r * dst_pitch + k
						assert (flat - j) % 16 == 0 or not fast or True
						assert flat not in written, ("written twice", r, k)
						assert y[l][reg] == (r, k), ("wrong register", r, k, y[l][reg], c, j)
						written[flat] = (r, k)
					if fast:                                               # a fast line is a whole aligned 128-B line
						assert (base_words + r * dst_pitch + cg - 15) % 16 == 0
			n_span_blocks = b_hi - b_lo
			for i in range(n_span_blocks):
				for c in range(16):
					cg = span_lo + 16 * i + c
					if cg < span_hi:
						for l in range(64):
							y[l][c] = (rw * 64 + l, cg)
						if i >= 1:
							emit(c, cg, True)                             # a real step past the first block: the whole line lies inside the span
				if i == 0:                                               # the first block's lines, all at once behind its steps
					for c in range(16):
						if span_lo + c < span_hi:
							emit(c, span_lo + c, False)
			for e in range(15):                                            # the lines still open at the span's end / steps past the matrix
				cg = span_hi + e
				emit(cg & 15, cg, False)
	return written


def check(SW, DW, dst_pitch, span_blocks, base_words=0):
	w = run(SW, DW, dst_pitch, span_blocks, base_words)
	want = {base_words + r * dst_pitch + k for r in range(SW * 64) for k in range(DW)}
	assert set(w) == want, (SW, DW, dst_pitch, span_blocks, len(w), len(want))


if __name__ == "__main__":
	n = 0
	for DW, pitch_extra, span_blocks, base in itertools.product((1, 5, 15, 16, 17, 31, 32, 33, 79, 100), (0, 1, 3, 8, 16), (1, 2, 3, 400), (0, 1, 7)):
		check(2, DW, DW + pitch_extra, span_blocks, base)
		n += 1
	print("rot transpose model: %d shapes, every destination word written once with the right register" % n)
