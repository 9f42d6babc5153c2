"""Multi-GPU partition of the splice path: chromosome copies shard across ranks in contiguous blocks of whole
64-copy words (so the bit-packed path matrix splits on word boundaries), graph and reference are replicated,
rows stay in file order when the ranks' outputs are concatenated.  No collective is needed on the data path
(rows are independent, haplotype_output.cc:62-81); torch.distributed is used only for barriers and for the
max-over-ranks of timings."""

PLOIDY_MAX = 0xFFFFFFFF


def shard_copies(n_copies, world, rank):
	"""Returns (first copy, end copy, padded local copy count) of `rank`."""
	n_words = (n_copies + 63) // 64
	base, extra = divmod(n_words, world)
	w0 = rank * base + min(rank, extra)
	w1 = w0 + base + (1 if rank < extra else 0)
	return min(n_copies, 64 * w0), min(n_copies, 64 * w1), 64 * (w1 - w0)


def local_rows(n_copies, world, rank, include_reference=True):
	"""Row specs of `rank` as (global output row index, local copy index or PLOIDY_MAX).
	Output row 0 is REF (rank 0), row 1 + c is chromosome copy c (haplotype_output.cc:48-81)."""
	c0, c1, _ = shard_copies(n_copies, world, rank)
	rows = []
	if include_reference and rank == 0:
		rows.append((0, PLOIDY_MAX))
	off = 1 if include_reference else 0
	rows.extend((off + c, c - c0) for c in range(c0, c1))
	return rows


def max_over_ranks(value, dist=None, device=None):
	"""MAX all-reduce of a python float (identity without an initialised process group)."""
	if dist is None or not dist.is_available() or not dist.is_initialized():
		return value
	import torch
	t = torch.tensor([value], dtype=torch.float64, device=device)
	dist.all_reduce(t, op=dist.ReduceOp.MAX)
	return float(t.item())
