"""Small synthetic inputs for parity tests (test infrastructure): random reference + VCF text, built
into a graph by the oracle's builder, optionally with the path matrix replaced by random bits."""

import os

import numpy as np

import oracle

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_reference(rng, length):
	return BASES[rng.integers(0, 4, size=length)].tobytes()


def _alt_base(rng, ref_base):
	others = [b for b in b"ACGT" if b != ref_base]
	return bytes([others[int(rng.integers(0, 3))]])


def random_records(rng, ref, n_variants, n_samples, ploidy=2, mix=(0.8, 0.1, 0.1), max_indel=32, long_every=0, multi_allelic=0.0, density=None):
	"""Returns a list of (pos0, ref_allele, [alt alleles], gt matrix [n_samples][ploidy] of allele numbers).
	mix = (snv, insertion, deletion) fractions.  long_every > 0 plants a long deletion / insertion
	(hundreds of bases to tens of kb) every that many records.  Overlapping records are left in."""
	R = len(ref)
	pos = np.sort(rng.choice(R - 64, size=n_variants, replace=False))
	recs = []
	for i, p in enumerate(pos):
		p = int(p)
		u = rng.random()
		alts = []
		ref_allele = ref[p:p + 1]
		if long_every and i % long_every == long_every - 1:
			span = int(rng.integers(200, 40000))
			if rng.random() < 0.5:
				ref_allele = ref[p:min(R, p + span)]
				alts = [ref_allele[:1]]
			else:
				alts = [ref_allele + random_reference(rng, min(span, 3000))]
		elif u < mix[0]:
			alts = [_alt_base(rng, ref[p])]
		elif u < mix[0] + mix[1]:
			k = min(max_indel, int(rng.geometric(1 / 3.0)))
			alts = [ref_allele + random_reference(rng, k)]
		else:
			k = min(max_indel, int(rng.geometric(1 / 3.0)))
			ref_allele = ref[p:min(R, p + 1 + k)]
			alts = [ref_allele[:1]]
		if rng.random() < multi_allelic:
			extra = _alt_base(rng, ref[p]) + random_reference(rng, int(rng.integers(0, 3)))
			if extra not in alts and extra != ref_allele:
				alts.append(extra)
		f = density if density is not None else 0.5 * 10 ** (-3 * rng.random())
		gt = (rng.random((n_samples, ploidy)) < f).astype(np.int64)
		if len(alts) > 1:
			gt *= rng.integers(1, len(alts) + 1, size=gt.shape)
		recs.append((p, ref_allele, alts, gt))
	return recs


def write_inputs(dirpath, ref, recs, n_samples, chrom="1", phased=True, name="synth"):
	fa = os.path.join(dirpath, name + ".fa")
	vcf = os.path.join(dirpath, name + ".vcf")
	with open(fa, "wb") as f:
		f.write(b">" + chrom.encode() + b"\n")
		for i in range(0, len(ref), 60):
			f.write(ref[i:i + 60] + b"\n")
	sep = "|" if phased else "/"
	with open(vcf, "w") as f:
		f.write("##fileformat=VCFv4.2\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n")
		f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("S%d" % i for i in range(n_samples)) + "\n")
		for k, (p, ra, alts, gt) in enumerate(recs):
			gts = "\t".join(sep.join(str(int(a)) for a in row) for row in gt)
			f.write("%s\t%d\tv%d\t%s\t%s\t.\tPASS\t.\tGT\t%s\n" % (chrom, p + 1, k, ra.decode(), ",".join(a.decode() for a in alts), gts))
	return fa, vcf


def build_case(tmpdir, seed, ref_len, n_variants, n_samples, **kw):
	rng = np.random.default_rng(seed)
	ref = random_reference(rng, ref_len)
	recs = random_records(rng, ref, n_variants, n_samples, **kw)
	fa, vcf = write_inputs(str(tmpdir), ref, recs, n_samples)
	g = oracle.build_variant_graph(fa, vcf, "1")
	assert g.ref == ref
	return g


def with_random_paths(g, seed, density):
	"""Same nodes/edges, path matrix replaced by iid random bits of the given density (creates many
	overlapping set edges, i.e. exercises the skip rule far beyond what genotypes would)."""
	rng = np.random.default_rng(seed)
	ep, hp = g.path_rows, g.path_cols
	words = np.zeros(ep // 64 * hp, dtype=np.uint64)
	if ep and hp:
		bits = rng.random((hp, ep)) < density
		bits[:, g.edge_count:] = False            # padding rows stay zero (variant_graph.cc:445-451)
		words = np.packbits(bits, axis=1, bitorder="little").view("<u8").reshape(-1).copy()
	g2 = oracle.graph_from_arrays(g.reference_positions, g.aligned_positions, g.alt_edge_targets, g.alt_edge_count_csum,
		g.label_offsets, g.label_bytes, words, ep, hp, g.sample_names, g.ploidy_csum)
	g2.ref = g.ref
	return g2
