"""Multi-GPU partition of the splice path: chromosome copies shard across ranks in contiguous blocks of whole
bytes of the bit-packed path matrix (8 copies: a rank's slice of paths_by_edge_and_chrom_copy is then a strided 2-D
byte copy, bytes [c0/8, c1/8) of every column, zero-padded to a multiple of 64 rows for the transpose), graph and
reference are replicated, rows stay in file order when the ranks' outputs are concatenated.  No collective is needed on the data path
(rows are independent, haplotype_output.cc:62-81); torch.distributed is used only for barriers and for the
max-over-ranks of timings."""

PLOIDY_MAX = 0xFFFFFFFF


COPY_GRANULE = 8


def shard_copies(n_copies, world, rank):
	"""Returns (first copy, end copy, padded local copy count) of `rank`.  Blocks that do not divide evenly go to the
	last ranks: rank 0 also carries the REF row."""
	n_blocks = (n_copies + COPY_GRANULE - 1) // COPY_GRANULE
	base, extra = divmod(n_blocks, world)
	first_heavy = world - extra
	b0 = rank * base + max(0, rank - first_heavy)
	b1 = b0 + base + (1 if rank >= first_heavy else 0)
	c0, c1 = min(n_copies, COPY_GRANULE * b0), min(n_copies, COPY_GRANULE * b1)
	return c0, c1, 64 * ((c1 - c0 + 63) // 64)


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
