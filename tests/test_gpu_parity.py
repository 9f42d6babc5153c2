"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the goldens.
Bit-exact everywhere (byte / bit / index work)."""

import io
import json
import os

import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _goldens(name):
	with open(os.path.join(HERE, "golden", "reference_goldens.json")) as f:
		return json.load(f)[name]


def _case_id(c):
	return c["vcf"] + "+" + c["fasta"]


@pytest.fixture(scope="module")
def v2m():
	import vcf2multialign_amd as v
	return v


@pytest.fixture(scope="module")
def ctx(v2m):
	c = v2m.Context(0)
	yield c
	c.close()


def _upload(v2m, ctx, g):
	vg = v2m.VariantGraph.from_object(g)
	ctx.upload_graph(vg, g.ref)
	return vg


def _oracle_rows(g, rows, unaligned=False):
	out = []
	for r in rows:
		if isinstance(r, (int, np.integer)):
			out.append(g.output_sequence(g.ref, copy_index=int(r), unaligned=unaligned))
		else:
			out.append(g.output_sequence(g.ref, cuts=list(r), unaligned=unaligned))
	return out


# ---- transpose_matrix ----------------------------------------------------------------------

def _set_bit(words, nrows, r, c):
	idx = c * nrows + r
	words[idx >> 6] |= np.uint64(1) << np.uint64(idx & 63)


@pytest.mark.parametrize("case", _goldens("transpose_matrix"), ids=lambda c: "%dx%d" % (c["rows"], c["cols"]))
def test_transpose_fixed(ctx, case):   # tests/transpose_matrix.cc:188-251
	src = np.zeros(case["rows"] * case["cols"] // 64, dtype=np.uint64)
	exp = np.zeros_like(src)
	_set_bit(src, case["rows"], *case["set_bit"])
	_set_bit(exp, case["expected_rows"], *case["expected_bit"])
	assert np.array_equal(ctx.transpose_matrix(src, case["rows"], case["cols"]), exp)


# every kernel of the product build, with the dispatch-order switches (the other shapes and flavours exist in the tuning build
# only: tests/test_gpu_tuning_build.py)
TRANSPOSE_KERNELS = ["8x8", "stream16", "8x8/rr", "stream16/pf", "lines8", "lines8:1", "lines8:3/sf", "lines8:8/rr", "lines8:400",    # (the last: whole columns, merged column ends)
	"lines16", "lines16:1", "lines16:2/sf", "lines16:5/rr", "lines16:400"]


@pytest.mark.parametrize("kernel", TRANSPOSE_KERNELS)
@pytest.mark.parametrize("h,w", [(1, 1), (1, 2), (2, 1), (3, 5), (8, 8), (9, 7), (16, 17), (79, 33), (5, 130), (64, 64), (17, 15), (33, 31), (2, 200), (3, 16), (130, 79), (4, 113)])
def test_transpose_random(ctx, monkeypatch, kernel, h, w):  # tests/transpose_matrix.cc:254-279 (1/3 of the bits set), plus ragged panels
	monkeypatch.setenv("V2M_TRANSPOSE_PANEL", kernel)   # both transpose kernels (the library picks per shape by measurement)
	rng = np.random.default_rng(1000 * h + w)
	rows, cols = 64 * h, 64 * w
	src = rng.integers(0, 2 ** 63, size=rows * cols // 64, dtype=np.uint64) & rng.integers(0, 2 ** 63, size=rows * cols // 64, dtype=np.uint64)
	src |= rng.integers(0, 2, size=src.size, dtype=np.uint64) << np.uint64(63)
	got = ctx.transpose_matrix(src, rows, cols)
	assert np.array_equal(got, oracle.transpose_matrix(src, rows, cols, naive=True))
	assert np.array_equal(ctx.transpose_matrix(got, cols, rows), src)   # involution


@pytest.mark.parametrize("kernel", TRANSPOSE_KERNELS)
@pytest.mark.parametrize("h,w", [(1, 1), (3, 17), (17, 3), (9, 33), (16, 16), (1, 79), (79, 1), (13, 257)])
def test_transpose_writes_only_the_destination(ctx, monkeypatch, kernel, h, w):
	"""Device-resident transpose with guard words around the destination: panels that stick out over the matrix edge
	must not write there."""
	import torch
	monkeypatch.setenv("V2M_TRANSPOSE_PANEL", kernel)
	rng = np.random.default_rng(77 * h + w)
	rows, cols = 64 * h, 64 * w
	n = rows * cols // 64
	src = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64) | (rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63))
	guard = 4096
	d_src = torch.from_numpy(src.view(np.int64)).cuda()
	d_dst = torch.full((n + 2 * guard,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device="cuda")
	torch.cuda.synchronize()
	ctx.transpose_bits_device(d_src.data_ptr(), rows, cols, d_dst.data_ptr() + 8 * guard)
	ctx.synchronize()
	host = d_dst.cpu().numpy().view(np.uint64)
	assert (host[:guard] == 0x5A5A5A5A5A5A5A5A).all() and (host[guard + n:] == 0x5A5A5A5A5A5A5A5A).all()
	assert np.array_equal(host[guard:guard + n], oracle.transpose_matrix(src, rows, cols, naive=True))


def test_product_build_has_no_tuning_kernels(ctx, v2m, monkeypatch):
	"""The shapes that were only ever measured are not compiled into libv2m_hip.so; asking for one is an error, not a fallback."""
	monkeypatch.setenv("V2M_TRANSPOSE_PANEL", "ring:16,8,8,4,64")
	with pytest.raises(v2m.V2MError) as e:
		ctx.transpose_matrix(np.zeros(64, np.uint64), 64, 64)
	assert e.value.code == 1 and "V2M_TUNING_BUILD" in str(e.value)
	monkeypatch.setenv("V2M_TRANSPOSE_PANEL", "4x16")
	with pytest.raises(v2m.V2MError):
		ctx.transpose_matrix(np.zeros(64, np.uint64), 64, 64)


def test_transpose_edge_cases(ctx, v2m):
	assert ctx.transpose_matrix(np.zeros(0, np.uint64), 64, 0).size == 0           # transpose_matrix.cc:48-49
	with pytest.raises(v2m.V2MError) as e:
		ctx.transpose_matrix(np.zeros(2, np.uint64), 64, 2)                         # asserted at transpose_matrix.cc:53-54
	assert e.value.code == 2


# ---- reference goldens through the GPU -----------------------------------------------------

@pytest.mark.parametrize("case", _goldens("founder_sequences"), ids=_case_id)
def test_founder_a2m_goldens(v2m, ctx, case):   # tests/founder_sequences.cc:118-188
	d = os.path.join(HERE, "golden", "reference-fixtures", "founder-sequences")
	g = oracle.build_variant_graph(os.path.join(d, case["fasta"]), os.path.join(d, case["vcf"]), case["chromosome"])
	vg = _upload(v2m, ctx, g)
	out = io.BytesIO()
	v2m.FounderSequenceGreedyOutput(ctx).output_a2m(vg, case["cut_positions"], case["assigned_samples_column_major"], out)
	assert out.getvalue().decode() == case["expected_a2m"]


@pytest.mark.parametrize("stem,fasta", [("test-1a", "test-1.fa"), ("test-1b", "test-1.fa"), ("test-2", "test-2.fa"), ("test-3", "test-3.fa"), ("test-4", "test-4.fa")])
def test_haplotype_a2m_fixtures(v2m, ctx, stem, fasta):   # haplotype_output.cc:38-82 on the variant-graph fixtures
	d = os.path.join(HERE, "golden", "reference-fixtures", "variant-graph")
	g = oracle.build_variant_graph(os.path.join(d, fasta), os.path.join(d, stem + ".vcf"), "1")
	vg = _upload(v2m, ctx, g)
	out = io.BytesIO()
	n = v2m.HaplotypeOutput(ctx).output_a2m(vg, out)
	assert n == 1 + g.total_chromosome_copies
	with open(os.path.join(HERE, "golden", "derived", stem + ".haplotypes.a2m"), "rb") as f:
		assert out.getvalue() == f.read()
	out = io.BytesIO()
	v2m.HaplotypeOutput(ctx, chromosome_id="chrT", should_output_reference=False).output_a2m(vg, out)
	with open(os.path.join(HERE, "golden", "derived", stem + ".haplotypes.chr.noref.a2m"), "rb") as f:
		assert out.getvalue() == f.read()
	out = io.BytesIO()
	v2m.HaplotypeOutput(ctx, should_output_unaligned=True).output_a2m(vg, out)   # --unaligned (sequence_writer.cc:80)
	with open(os.path.join(HERE, "golden", "derived", stem + ".haplotypes.unaligned.fa"), "rb") as f:
		assert out.getvalue() == f.read()


# ---- synthetic graphs ----------------------------------------------------------------------

SYNTH = [
	# seed, ref_len, variants, samples, kwargs
	(1, 3000, 60, 4, {}),
	(2, 50000, 800, 8, {"mix": (1.0, 0.0, 0.0)}),                       # SNV only (config 2 shape)
	(3, 200000, 3000, 12, {}),                                          # config 3 mix, several tiles
	(4, 120000, 1500, 6, {"long_every": 40}),                           # long deletions / insertions crossing tiles
	(5, 40000, 4000, 5, {"multi_allelic": 0.3, "density": 0.3}),        # dense, multi-allelic, many overlaps
	(6, 70000, 900, 3, {"ploidy": 1}),
	(7, 30000, 500, 40, {}),                                            # 80 copies: two words per path-matrix column
]


@pytest.mark.parametrize("seed,ref_len,n_var,n_samples,kw", SYNTH, ids=lambda x: str(x) if isinstance(x, int) else None)
def test_synthetic_haplotypes(v2m, ctx, tmp_path, seed, ref_len, n_var, n_samples, kw):
	g = synth.build_case(tmp_path, seed, ref_len, n_var, n_samples, **kw)
	_upload(v2m, ctx, g)
	rows = [v2m.PLOIDY_MAX] + list(range(g.total_chromosome_copies))
	got = ctx.splice_rows(rows)
	exp = _oracle_rows(g, rows)
	assert len(got) == len(exp)
	for i, (a, b) in enumerate(zip(got, exp)):
		assert len(a) == g.aligned_length
		assert a == b, "row %d differs" % i
	got = ctx.splice_rows(rows, unaligned=True)
	exp = _oracle_rows(g, rows, unaligned=True)
	assert got[0] == g.ref
	for i, (a, b) in enumerate(zip(got, exp)):
		assert a == b, "unaligned row %d differs (lengths %d / %d)" % (i, len(a), len(b))


@pytest.mark.parametrize("density,max_back", [(0.02, None), (0.3, None), (0.9, None), (0.3, "0"), (0.9, "1")])
def test_random_path_bits_skip_rule(v2m, ctx, tmp_path, monkeypatch, density, max_back):
	"""iid random path bits on a graph full of overlapping edges (long deletions, multi-allelic sites):
	the skip semantics of sequence_writer.cc:51-67 decide most rows.  max_back forces the rows whose
	restart point lies in an earlier word through the one-wave-per-row serial kernel."""
	if max_back is not None:
		monkeypatch.setenv("V2M_MAX_BACK_WORDS", max_back)
	g0 = synth.build_case(tmp_path, 11, 60000, 5000, 4, multi_allelic=0.2, long_every=97)
	g = synth.with_random_paths(g0, 5, density)
	_upload(v2m, ctx, g)
	rows = list(range(min(8, g.path_cols)))
	got = ctx.splice_rows(rows)
	for i, (a, b) in enumerate(zip(got, _oracle_rows(g, rows))):
		assert a == b, "row %d differs" % i


@pytest.mark.parametrize("nt", ["0", "1"])
def test_both_store_flavours(v2m, ctx, tmp_path, monkeypatch, nt):
	"""The aligned kernel exists with plain and with nontemporal output stores (chosen by calibration on large
	launches); both must give the same bytes."""
	monkeypatch.setenv("V2M_NT_STORES", nt)
	g = synth.build_case(tmp_path, 8, 100000, 1500, 10, long_every=60)
	_upload(v2m, ctx, g)
	rows = [v2m.PLOIDY_MAX] + list(range(g.total_chromosome_copies))
	assert ctx.splice_rows(rows) == _oracle_rows(g, rows)


def test_extreme_spans_and_cache_limits(v2m, ctx, tmp_path, monkeypatch):
	"""Spans and labels beyond what the workgroup's LDS cache can describe (>= 64 KiB spans, labels past the cached
	2 KiB slice), more long patches in one tile-row than the long-span queue holds, and row groups larger than the
	cached effective-bit rows: all must fall back to the global-memory path without changing a byte."""
	rng = np.random.default_rng(77)
	ref = synth.random_reference(rng, 400000)
	recs = []
	def rec(pos, ref_len, alts, gts):
		recs.append((pos, ref[pos:pos + ref_len], alts, np.array(gts)))
	n = 3   # samples, diploid
	rec(1000, 150000, [ref[1000:1001]], [[1, 0], [0, 0], [1, 1]])                        # 150 kb deletion (span >= 64 KiB)
	for k in range(30):                                                                 # variants under it: skipped by copies 0, 4, 5
		rec(2000 + 500 * k, 1, [synth._alt_base(rng, ref[2000 + 500 * k])], [[1, 1], [1, 0], [0, 1]])
	rec(160000, 1, [ref[160000:160001] + synth.random_reference(rng, 70000)], [[0, 1], [1, 0], [0, 0]])   # 70 kb insertion
	for k in range(40):                                                                 # 40 insertions of 200 bp inside one 16-KiB tile, all carried by copy 1
		p = 250000 + 150 * k
		rec(p, 1, [ref[p:p + 1] + synth.random_reference(rng, 200)], [[0, 1], [0, 0], [1, 0]])
	rec(300000, 5, [ref[300000:300001], ref[300000:300001] + b"ACGTACGTACGT"], [[1, 2], [2, 1], [0, 2]])
	fa, vcf = synth.write_inputs(str(tmp_path), ref, recs, n)
	g = oracle.build_variant_graph(fa, vcf, "1")
	assert g.aligned_length > 470000
	_upload(v2m, ctx, g)
	rows = [v2m.PLOIDY_MAX] + list(range(6)) + [[(0, 0), (g.node_count - 3, 3)]]
	exp = _oracle_rows(g, rows)
	assert ctx.splice_rows(rows) == exp
	assert ctx.splice_rows(rows, unaligned=True) == _oracle_rows(g, rows, unaligned=True)
	many = rows * 6                                                                    # 48 rows in one group of 40: rows 16.. use the uncached path
	monkeypatch.setenv("V2M_ROWS_PER_GROUP", "40")
	assert ctx.splice_rows(many) == exp * 6
	assert ctx.splice_rows(many, unaligned=True) == _oracle_rows(g, rows, unaligned=True) * 6


def test_founder_rows_synthetic(v2m, ctx, tmp_path):
	"""Copy switching at cut nodes that no edge spans (founder_sequence_greedy_output.cc:106-114)."""
	g = synth.build_case(tmp_path, 21, 80000, 1200, 10)
	_upload(v2m, ctx, g)
	# bridge nodes: nodes that no edge jumps over
	reach = 0
	bridges = []
	for n in range(g.node_count - 1):
		if n >= reach and n > 0:
			bridges.append(n)
		for e in range(int(g.alt_edge_count_csum[n]), int(g.alt_edge_count_csum[n + 1])):
			reach = max(reach, int(g.alt_edge_targets[e]))
	rng = np.random.default_rng(3)
	H = g.total_chromosome_copies
	rows = []
	for _ in range(6):
		cuts = [0] + sorted(int(x) for x in rng.choice(bridges, size=min(25, len(bridges)), replace=False))
		copies = [int(x) for x in rng.integers(0, H, size=len(cuts))]
		copies[2] = v2m.PLOIDY_MAX                                  # unassigned slot (founder_sequence_greedy_output.cc:172)
		rows.append(list(zip(cuts, copies)))
	rows.append([(bridges[3], 1)])                                  # first cut not at node 0: REF until then
	got = ctx.splice_rows(rows)
	for i, (a, b) in enumerate(zip(got, _oracle_rows(g, rows))):
		assert a == b, "row %d differs" % i
	got = ctx.splice_rows(rows, unaligned=True)
	for i, (a, b) in enumerate(zip(got, _oracle_rows(g, rows, unaligned=True))):
		assert a == b, "unaligned row %d differs" % i


@pytest.mark.parametrize("max_back", [None, "0"])
def test_founder_rows_many_segments(v2m, ctx, tmp_path, monkeypatch, max_back):
	"""Rows that switch copy at (nearly) every bridge node: dozens of segments per 64-edge word, put together by
	assemble_row_bits_kernel over several waves and workgroups; next to rows with a handful of long segments and plain rows."""
	if max_back is not None:
		monkeypatch.setenv("V2M_MAX_BACK_WORDS", max_back)          # every cross-word restart goes through the serial kernel
	g = synth.with_random_paths(synth.build_case(tmp_path, 77, 1_500_000, 30000, 12, multi_allelic=0.1), 5, 0.3)
	_upload(v2m, ctx, g)
	assert g.edge_count > 64 * 64 * 4 + 100                          # more than one workgroup of four 64-word waves
	reach, bridges = 0, []
	for n in range(g.node_count - 1):
		if n >= reach and n > 0:
			bridges.append(n)
		for e in range(int(g.alt_edge_count_csum[n]), int(g.alt_edge_count_csum[n + 1])):
			reach = max(reach, int(g.alt_edge_targets[e]))
	rng = np.random.default_rng(9)
	H = g.total_chromosome_copies

	def row(cut_nodes):
		copies = rng.integers(0, H, size=len(cut_nodes)).tolist()
		for k in rng.choice(len(cut_nodes), size=max(1, len(cut_nodes) // 50), replace=False):
			copies[int(k)] = v2m.PLOIDY_MAX                            # unassigned slots
		return list(zip(cut_nodes, copies))

	rows = [
		row([0] + bridges),                                            # every bridge node
		row(bridges[1::2]),                                            # first cut not at node 0
		3,
		row([0] + sorted(int(x) for x in rng.choice(bridges, size=len(bridges) // 3, replace=False))),
		row([0, bridges[len(bridges) // 2]]),                          # two segments, thousands of words each
		v2m.PLOIDY_MAX,
		row([0] + sorted(int(x) for x in rng.choice(bridges, size=7, replace=False))),
		row([0] + bridges[:200]),                                      # dense at the start, one long tail
	]
	exp = _oracle_rows(g, rows)
	got = ctx.splice_rows(rows)
	for i, (a, b) in enumerate(zip(got, exp)):
		assert a == b, "row %d differs" % i
	assert ctx.splice_rows(rows[:2], unaligned=True) == _oracle_rows(g, rows[:2], unaligned=True)


@pytest.mark.parametrize("seed", range(int(os.environ.get("V2M_FUZZ_SEEDS", "96"))))   # a longer soak: V2M_FUZZ_SEEDS=2000
def test_fuzz_small_graphs(v2m, ctx, tmp_path, seed):
	"""Random small inputs with random shapes (0 .. 3000 records over 100 .. 70 000 bases, 1-5 samples, any variant mix,
	long indels, multi-allelic sites, genotype or iid path bits): every row, REF, and founder-style rows cut at random
	bridge nodes, aligned and unaligned, against the oracle."""
	rng = np.random.default_rng(5000 + seed)
	ref_len = int(rng.integers(100, 70000))
	n_var = int(rng.integers(0, max(1, min(ref_len // 3, 3000))))
	n_samples = int(rng.integers(1, 6))
	snv = float(rng.random())
	ins = float(rng.random()) * (1 - snv)
	kw = dict(mix=(snv, ins, 1 - snv - ins), multi_allelic=float(rng.choice([0.0, 0.1, 0.4])), max_indel=int(rng.choice([4, 32, 200])),
		long_every=int(rng.choice([0, 0, 7, 30])), ploidy=int(rng.choice([1, 2, 2, 3])))
	density = rng.choice([-1.0, 0.5, 0.05])
	if density > 0:
		kw["density"] = float(density)
	if n_var == 0:
		ref = synth.random_reference(rng, ref_len)
		fa, vcf = synth.write_inputs(str(tmp_path), ref, [], n_samples)
		g = oracle.build_variant_graph(fa, vcf, "1")
	else:
		g = synth.build_case(tmp_path, 6000 + seed, ref_len, n_var, n_samples, **kw)
		if seed % 2:
			g = synth.with_random_paths(g, seed, float(rng.choice([0.02, 0.3, 0.8])))
	_upload(v2m, ctx, g)
	H = g.total_chromosome_copies
	rows = [v2m.PLOIDY_MAX] + list(range(H))
	reach, bridges = 0, []
	for n in range(g.node_count - 1):
		if n >= reach and n > 0:
			bridges.append(n)
		for e in range(int(g.alt_edge_count_csum[n]), int(g.alt_edge_count_csum[n + 1])):
			reach = max(reach, int(g.alt_edge_targets[e]))
	if bridges and H:
		for _ in range(3):
			k = int(rng.integers(1, min(len(bridges), 400) + 1))
			cuts = sorted(int(x) for x in rng.choice(bridges, size=k, replace=False))
			if rng.random() < 0.7:
				cuts = [0] + cuts
			copies = [int(x) if rng.random() > 0.05 else v2m.PLOIDY_MAX for x in rng.integers(0, H, size=len(cuts))]
			rows.append(list(zip(cuts, copies)))
	order = rng.permutation(len(rows))
	rows = [rows[i] for i in order]
	got = ctx.splice_rows(rows)
	for i, (a, b) in enumerate(zip(got, _oracle_rows(g, rows))):
		assert a == b, "seed %d row %d (%r)" % (seed, i, rows[i] if not isinstance(rows[i], list) else "cuts")
	got = ctx.splice_rows(rows, unaligned=True)
	for i, (a, b) in enumerate(zip(got, _oracle_rows(g, rows, unaligned=True))):
		assert a == b, "seed %d unaligned row %d" % (seed, i)


def test_more_rows_than_one_grid_dimension(v2m, ctx):
	"""70 000 rows in one call (grid.y of the resolve / bit-assembly launches is limited to 65 535 rows, so they are
	issued in two pieces), plain, REF and founder rows interleaved, aligned and unaligned."""
	d = os.path.join(HERE, "golden", "reference-fixtures", "founder-sequences")
	g = oracle.build_variant_graph(os.path.join(d, "test-2.fa"), os.path.join(d, "test-2.vcf"), "1")
	_upload(v2m, ctx, g)
	H = g.total_chromosome_copies
	with open(os.path.join(HERE, "golden", "reference_goldens.json")) as f:
		import json
		case = next(c for c in json.load(f)["founder_sequences"] if c["vcf"] == "test-2.vcf")
	k = case["assigned_samples_rows"]
	founder = [list(zip(case["cut_positions"][:-1], case["assigned_samples_column_major"][f * k:(f + 1) * k])) for f in range(case["founder_count"])]
	kinds = [v2m.PLOIDY_MAX] + list(range(H)) + founder
	exp_kinds = _oracle_rows(g, kinds)
	uexp_kinds = _oracle_rows(g, kinds, unaligned=True)
	n = 70000
	rows = [kinds[i % len(kinds)] for i in range(n)]
	got = ctx.splice_rows(rows)
	assert len(got) == n
	for i in (0, 1, 2, 65534, 65535, 65536, 65537, n - 1):
		assert got[i] == exp_kinds[i % len(kinds)], "row %d" % i
	assert all(got[i] == exp_kinds[i % len(kinds)] for i in range(n))
	ugot = ctx.splice_rows(rows, unaligned=True)
	assert all(ugot[i] == uexp_kinds[i % len(kinds)] for i in range(n))


def test_device_rows_checksums(v2m, ctx, tmp_path):
	"""Device-resident output + on-device checksums == host checksums of the oracle rows; and the
	sink path cut into several ring slices gives the same bytes."""
	import torch
	g = synth.build_case(tmp_path, 31, 150000, 2500, 16)
	_upload(v2m, ctx, g)
	rows = [v2m.PLOIDY_MAX] + list(range(g.total_chromosome_copies))
	exp = _oracle_rows(g, rows)
	pitch = ctx.min_row_pitch
	assert pitch % 256 == 0 and pitch >= g.aligned_length
	buf = torch.zeros(len(rows) * pitch, dtype=torch.uint8, device="cuda")
	torch.cuda.synchronize()
	ctx.splice_rows_device(rows, buf.data_ptr(), pitch)
	sums = ctx.checksum_rows_device(buf.data_ptr(), pitch, len(rows), length=g.aligned_length)
	ctx.synchronize()
	assert np.array_equal(sums, v2m.checksum_rows_host(exp))
	host = buf.cpu().numpy().reshape(len(rows), pitch)
	for i, b in enumerate(exp):
		assert host[i, :len(b)].tobytes() == b
	# unaligned, device-resident: ragged row lengths
	exp = _oracle_rows(g, rows, unaligned=True)
	upitch = (ctx.max_unaligned_length + 255) // 256 * 256
	assert upitch >= max(len(b) for b in exp)
	buf = torch.zeros(len(rows) * upitch, dtype=torch.uint8, device="cuda")
	torch.cuda.synchronize()
	lengths = ctx.splice_rows_device(rows, buf.data_ptr(), upitch, unaligned=True, want_lengths=True)
	assert lengths.tolist() == [len(b) for b in exp]
	sums = ctx.checksum_rows_device(buf.data_ptr(), upitch, len(rows), lengths=lengths)
	assert np.array_equal(sums, v2m.checksum_rows_host(exp))


def test_nothing_is_written_outside_the_rows(v2m, ctx, tmp_path):
	"""Guard bytes: with a pitch wider than the rows and guard zones before the first and after the last row, the kernels
	may touch only the rows themselves (aligned: up to the row length rounded up to 16 bytes; unaligned: exactly the
	row's bytes)."""
	import torch
	g = synth.build_case(tmp_path, 41, 90000, 1500, 9, long_every=60)
	_upload(v2m, ctx, g)
	L = g.aligned_length
	rows = [v2m.PLOIDY_MAX] + list(range(g.total_chromosome_copies))           # 19 rows: one full group of 16 + a ragged one
	guard = 1 << 16
	for unaligned in (False, True):
		exp = _oracle_rows(g, rows, unaligned=unaligned)
		pitch = ((ctx.max_unaligned_length if unaligned else L) + 255) // 256 * 256 + 768
		buf = torch.full((2 * guard + len(rows) * pitch,), 0xAB, dtype=torch.uint8, device="cuda")
		torch.cuda.synchronize()
		ctx.splice_rows_device(rows, buf.data_ptr() + guard, pitch, unaligned=unaligned)
		ctx.synchronize()
		host = buf.cpu().numpy()
		assert (host[:guard] == 0xAB).all() and (host[guard + len(rows) * pitch:] == 0xAB).all()
		body = host[guard:guard + len(rows) * pitch].reshape(len(rows), pitch)
		for i, b in enumerate(exp):
			assert body[i, :len(b)].tobytes() == b
			written_to = len(b) if unaligned else (len(b) + 15) // 16 * 16
			assert (body[i, written_to:] == 0xAB).all(), "row %d: bytes past %d were touched" % (i, written_to)


def test_sink_slices(v2m, ctx, tmp_path, monkeypatch):
	g = synth.build_case(tmp_path, 32, 90000, 1500, 8)
	monkeypatch.setenv("V2M_RING_SLOT_BYTES", "300000")   # ~3 rows per slice: several slices, both ring halves reused
	_upload(v2m, ctx, g)
	rows = [v2m.PLOIDY_MAX] + list(range(g.total_chromosome_copies))
	got = ctx.splice_rows(rows)
	for i, (a, b) in enumerate(zip(got, _oracle_rows(g, rows))):
		assert a == b, "row %d differs" % i
	got = ctx.splice_rows(rows, unaligned=True)
	for i, (a, b) in enumerate(zip(got, _oracle_rows(g, rows, unaligned=True))):
		assert a == b, "unaligned row %d differs" % i


def test_unaligned_keeps_dashes_and_case_of_the_reference(v2m, ctx, tmp_path):
	"""Only the padding is dropped in unaligned mode: a '-' or lowercase base that is IN the reference is data."""
	ref = b"ACG-TNNacgtAC--GTACGTTTGACA"
	recs = [(4, b"T", [b"TGG"], np.array([[1, 0]])), (8, b"cg", [b"c"], np.array([[0, 1]])), (15, b"G", [b"C", b"GAAAA"], np.array([[2, 1]]))]
	fa, vcf = synth.write_inputs(str(tmp_path), ref, recs, 1)
	g = oracle.build_variant_graph(fa, vcf, "1")
	_upload(v2m, ctx, g)
	rows = [v2m.PLOIDY_MAX, 0, 1]
	assert ctx.splice_rows(rows, unaligned=True) == _oracle_rows(g, rows, unaligned=True)
	assert ctx.splice_rows(rows) == _oracle_rows(g, rows)
	assert ctx.splice_rows([v2m.PLOIDY_MAX], unaligned=True) == [ref]


def test_alloc_output(v2m, ctx):
	"""v2m_alloc_output: plain allocation for small sizes, measured choice among candidates for large ones."""
	p = ctx.alloc_output(1 << 20, candidates=3)
	assert p and p % 256 == 0
	ctx.free_output(p)
	before = ctx.info
	p = ctx.alloc_output(3 << 30, candidates=3)          # 3 x 3 GiB candidates, probed
	assert p and "output buffer chosen among 3 hipMalloc candidates" in ctx.info[len(before):], ctx.info
	ctx.free_output(p)
	with pytest.raises(v2m.V2MError):
		ctx.alloc_output(0)


# ---- edge cases and error behaviour --------------------------------------------------------

def test_no_variants_and_empty_batches(v2m, ctx, tmp_path):
	ref = b"ACGTACGTAC"
	# one record whose only ALT is '.', i.e. no edge at all (variant_graph.cc:362-363)
	fa, vcf = synth.write_inputs(str(tmp_path), ref, [(2, b"G", [b"."], np.zeros((2, 2), dtype=np.int64))], 2)
	g = oracle.build_variant_graph(fa, vcf, "1")
	assert g.edge_count == 0 and g.node_count == 3 and g.total_chromosome_copies == 4
	_upload(v2m, ctx, g)
	assert ctx.splice_rows([v2m.PLOIDY_MAX, 0, 3]) == [ref, ref, ref]
	assert ctx.splice_rows([v2m.PLOIDY_MAX, 0, 3], unaligned=True) == [ref, ref, ref]
	assert ctx.splice_rows([]) == []


def test_precondition_errors(v2m, ctx, tmp_path):
	g = synth.build_case(tmp_path, 41, 60000, 400, 3, long_every=10)
	vg = _upload(v2m, ctx, g)
	with pytest.raises(v2m.V2MError) as e:                       # copy outside the matrix
		ctx.splice_rows([g.path_cols])
	assert e.value.code == 1
	# a cut node inside an edge span is what the reference asserts against (founder_sequence_greedy_output.cc:108)
	src_of = np.searchsorted(g.alt_edge_count_csum, np.arange(g.edge_count), side="right") - 1
	src = int(src_of[np.nonzero(g.alt_edge_targets.astype(np.int64) - src_of >= 2)[0][0]])
	with pytest.raises(v2m.V2MError) as e:
		ctx.splice_rows([[(0, 0), (src + 1, 1)]])
	assert e.value.code == 2
	# graph invariants: label longer than its aligned span (libbio_assert_lte at sequence_writer.cc:61)
	bad = v2m.VariantGraph.from_object(g)
	bad.aligned_positions = bad.aligned_positions.copy()
	bad.aligned_positions[1:] -= np.uint64(1)
	with pytest.raises(v2m.V2MError) as e:
		ctx.upload_graph(bad, g.ref)
	assert e.value.code == 2
	ctx.upload_graph(vg, g.ref)   # the context stays usable
	assert ctx.splice_rows([0]) == _oracle_rows(g, [0])


def test_nul_bytes_are_kept_in_aligned_mode_and_refused_in_unaligned_mode(v2m, ctx, tmp_path):
	"""Bytes are opaque to the walk: the reference streams whatever the FASTA / the VCF's ALT column held (sequence_writer.cc:73-74).
	Aligned mode keeps a NUL byte of the reference sequence or of a label; the unaligned kernels mark padding with byte 0, so that mode
	refuses such a graph (V2M_ERR_UNSUPPORTED) instead of writing rows that lack the byte."""
	g = synth.build_case(tmp_path, 43, 30000, 300, 3, mix=(0.5, 0.4, 0.1))
	rows = [v2m.PLOIDY_MAX] + list(range(g.total_chromosome_copies))
	# (1) in the reference sequence
	ref = bytearray(g.ref)
	for p in (0, 1234, 16384, len(ref) - 1):
		ref[p] = 0
	ref = bytes(ref)
	vg = v2m.VariantGraph.from_object(g)
	ctx.upload_graph(vg, ref)
	want = [g.output_sequence(ref, copy_index=r) for r in rows]
	assert want[0].count(b"\0") == 4 and ctx.splice_rows(rows) == want
	with pytest.raises(v2m.V2MError) as e:
		ctx.splice_rows(rows, unaligned=True)
	assert e.value.code == 3 and "NUL" in str(e.value)
	# (2) in an ALT label (the first byte of every fifth label)
	lb = bytearray(g.label_bytes)
	for o in g.label_offsets[:-1][::5]:
		if int(o) < len(lb):
			lb[int(o)] = 0
	g2 = oracle.graph_from_arrays(g.reference_positions, g.aligned_positions, g.alt_edge_targets, g.alt_edge_count_csum,
		g.label_offsets, bytes(lb), g.paths_by_chrom_copy_and_edge, g.path_rows, g.path_cols, g.sample_names, g.ploidy_csum)
	ctx.upload_graph(v2m.VariantGraph.from_object(g2), g.ref)
	want = [g2.output_sequence(g.ref, copy_index=r) for r in rows]
	assert b"\0" not in want[0] and any(b"\0" in w for w in want[1:]) and ctx.splice_rows(rows) == want
	with pytest.raises(v2m.V2MError) as e:
		ctx.splice_rows(rows, unaligned=True)
	assert e.value.code == 3
	# the same context takes a graph without such a byte in both modes afterwards
	ctx.upload_graph(vg, g.ref)
	assert ctx.splice_rows(rows, unaligned=True) == [g.output_sequence(g.ref, copy_index=r, unaligned=True) for r in rows]


# ---- --unaligned: chunks with padding ------------------------------------------------------------------------------

@pytest.mark.parametrize("mode", ["", "plain"])
def test_unaligned_rows_with_much_padding(ctx, v2m, tmp_path, monkeypatch, mode):
	"""Graphs dense with insertions: most 16-byte chunks of a tile hold padding -- runs of chunks with a few surviving bytes
	or none, ~1000 short chunks per 16-KiB row tile -- so every way the stream-out kernel stores a chunk is taken: the plain
	16-byte store, the workgroup's queue of 128 short chunks packed a row later by the rotating wave (and after the last row),
	and, for the slots that no longer fit the queue, the pack where they are; exact 8 / 4 / 2 / 1-byte pieces in both."""
	monkeypatch.setenv("V2M_UNALIGNED_STORE", mode)
	g = synth.build_case(tmp_path, 94, 120000, 9000, 6, mix=(0.2, 0.7, 0.1), max_indel=40)   # an insertion every ~20 bases
	vg = v2m.VariantGraph.from_object(g)
	ctx.upload_graph(vg, g.ref)
	rows = [v2m.PLOIDY_MAX] + list(range(g.total_chromosome_copies))
	got = ctx.splice_rows(rows, unaligned=True)
	for r, body in zip(rows, got):
		assert body == g.output_sequence(g.ref, copy_index=r, unaligned=True), r
	assert g.aligned_length > len(g.ref)                                     # there is padding to remove


# ---- v2m_upload_path_slice: one GPU's share of the path matrix ---------------------------

@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_path_slices_reproduce_every_row(ctx, v2m, tmp_path, world):
	"""Every rank uploads only its own copies out of the host-resident transpose input, transposes them on the GPU and
	splices its rows with copy indices relative to its shard; together the ranks produce exactly the oracle's file."""
	from vcf2multialign_amd.sharding import local_rows, shard_copies
	g = synth.build_case(tmp_path, 91, 50000, 700, 21, long_every=150)          # 42 copies: uneven shards, a ragged last block
	hp, ep = g.paths_by_edge_and_chrom_copy_dims
	n_copies = g.total_chromosome_copies
	vg = v2m.VariantGraph.from_object(g)
	vg.paths_by_chrom_copy_and_edge = None
	expected = [g.output_sequence(g.ref)] + [g.output_sequence(g.ref, copy_index=c) for c in range(n_copies)]
	got = {}
	for rank in range(world):
		ctx.upload_graph(vg, g.ref)
		c0, c1, hp_local = shard_copies(n_copies, world, rank)
		ctx.upload_path_slice(g.paths_by_edge_and_chrom_copy, hp, ep, c0, c1 - c0)
		rows = local_rows(n_copies, world, rank)
		for (global_row, _), body in zip(rows, ctx.splice_rows([local for _, local in rows])):
			got[global_row] = body
		if c1 > c0:
			with pytest.raises(v2m.V2MError):                                 # the context holds hp_local columns and no more
				ctx.splice_rows([hp_local])
			if hp_local > c1 - c0:                                            # padding copies of the shard carry no bits, whatever the next shard holds
				assert ctx.splice_rows([hp_local - 1])[0] == expected[0]
	assert [got[r] for r in range(n_copies + 1)] == expected


@pytest.mark.parametrize("kernel", ["", "8x8", "stream16", "lines8", "lines8:2/sf", "lines16", "lines16:3"])
def test_bind_path_matrix_device(ctx, v2m, tmp_path, monkeypatch, kernel):
	"""v2m_bind_path_matrix_device: a device-resident transpose input becomes the context's own (line-aligned) path matrix;
	every transpose kernel with a destination pitch that differs from the word count."""
	import torch
	if kernel:
		monkeypatch.setenv("V2M_TRANSPOSE_PANEL", kernel)
	g = synth.build_case(tmp_path, 95, 40000, 1300, 37, long_every=250)           # 74 copies -> 128 rows; 1300+ edges: 21 words, pitch 32
	hp, ep = g.paths_by_edge_and_chrom_copy_dims
	assert (ep // 64) % 16 != 0
	vg = v2m.VariantGraph.from_object(g)
	vg.paths_by_chrom_copy_and_edge = None
	ctx.upload_graph(vg, g.ref)
	d_src = torch.from_numpy(g.paths_by_edge_and_chrom_copy.view(np.int64)).cuda()
	torch.cuda.synchronize()
	ctx.bind_path_matrix_device(d_src.data_ptr(), hp, ep)
	rows = [v2m.PLOIDY_MAX] + list(range(g.total_chromosome_copies)) + [hp - 1]
	got = ctx.splice_rows(rows)
	assert got[:-1] == [g.output_sequence(g.ref, copy_index=r) for r in rows[:-1]]
	assert got[-1] == got[0]                                                      # a padding copy: REF
	got = ctx.splice_rows(rows[:9], unaligned=True)
	assert got == [g.output_sequence(g.ref, copy_index=r, unaligned=True) for r in rows[:9]]
	with pytest.raises(v2m.V2MError):
		ctx.bind_path_matrix_device(d_src.data_ptr(), hp, 64)                     # fewer edge columns than the graph has edges
	with pytest.raises(v2m.V2MError):
		ctx.bind_path_matrix_device(d_src.data_ptr(), hp - 1, ep)


@pytest.mark.parametrize("world,block", [(2, 8), (3, 8), (3, 16), (5, 24), (9, 8)])
def test_path_blocks_dealt_round_robin(v2m, tmp_path, world, block):
	"""v2m_upload_path_blocks: every world-th block of `block` chromosome copies on one context; together the contexts
	reproduce every row (local copy l = global first + (l // block) * stride + l % block), and a context refuses copies beyond
	its share."""
	g = synth.build_case(tmp_path, 97, 30000, 500, 43)                 # 86 copies -> 128 rows
	hp, ep = g.paths_by_edge_and_chrom_copy_dims
	n_copies = g.total_chromosome_copies
	expected = [g.output_sequence(g.ref, copy_index=c) for c in range(n_copies)]
	vg = v2m.VariantGraph.from_object(g)
	vg.paths_by_chrom_copy_and_edge = None
	seen = 0
	for rank in range(world):
		with v2m.Context(0) as ctx:
			ctx.upload_graph(vg, g.ref)
			ctx.upload_path_blocks(g.paths_by_edge_and_chrom_copy, hp, ep, block * rank, block, block * world)
			mine = [c for c in range(n_copies) if (c // block) % world == rank]
			local = [(c // (block * world)) * block + c % block for c in mine]
			assert local == sorted(local)
			if mine:
				got = ctx.splice_rows(local)
				assert got == [expected[c] for c in mine]
				assert ctx.splice_rows(local[:3], unaligned=True) == [g.output_sequence(g.ref, copy_index=c, unaligned=True) for c in mine[:3]]
			n_local = len([c for c in range(hp) if (c // block) % world == rank])
			with pytest.raises(v2m.V2MError):
				ctx.splice_rows([64 * ((n_local + 63) // 64)])
			seen += len(mine)
	assert seen == n_copies
	with v2m.Context(0) as ctx:
		ctx.upload_graph(vg, g.ref)
		for bad in ((4, 8, 16), (0, 12, 24), (0, 8, 20), (0, 16, 8), (0, 0, 8)):
			with pytest.raises(v2m.V2MError):
				ctx.upload_path_blocks(g.paths_by_edge_and_chrom_copy, hp, ep, *bad)


def test_path_slice_edges(ctx, v2m, tmp_path):
	g = synth.build_case(tmp_path, 92, 20000, 300, 40)
	hp, ep = g.paths_by_edge_and_chrom_copy_dims
	vg = v2m.VariantGraph.from_object(g)
	vg.paths_by_chrom_copy_and_edge = None
	ctx.upload_graph(vg, g.ref)
	ref_row = g.output_sequence(g.ref)
	# a slice that ends in the middle of a byte: the bits of the following copies must not leak into the padding
	ctx.upload_path_slice(g.paths_by_edge_and_chrom_copy, hp, ep, 8, 13)
	bodies = ctx.splice_rows(list(range(64)))
	for local, body in enumerate(bodies):
		assert body == (g.output_sequence(g.ref, copy_index=8 + local) if local < 13 else ref_row), local
	# the whole matrix in one piece == upload_graph with the (oracle-)transposed matrix
	ctx.upload_path_slice(g.paths_by_edge_and_chrom_copy, hp, ep)
	assert ctx.splice_rows([0, 5, 79]) == [g.output_sequence(g.ref, copy_index=c) for c in (0, 5, 79)]
	# an empty shard binds nothing: only REF rows can be asked for
	ctx.upload_path_slice(g.paths_by_edge_and_chrom_copy, hp, ep, 16, 0)
	assert ctx.splice_rows([v2m.PLOIDY_MAX]) == [ref_row]
	with pytest.raises(v2m.V2MError):
		ctx.splice_rows([0])
	for bad in ((4, 8), (0, hp + 1), (hp, 8)):
		with pytest.raises(v2m.V2MError):
			ctx.upload_path_slice(g.paths_by_edge_and_chrom_copy, hp, ep, *bad)
	with pytest.raises(v2m.V2MError):
		ctx.upload_path_slice(g.paths_by_edge_and_chrom_copy[:hp // 64 * 64], hp, 64, 0, hp)   # fewer edge columns than the graph has edges


def test_output_buffer_allocate_free_allocate(ctx, v2m, tmp_path):
	"""Every row lands in a buffer from v2m_alloc_output, also in one allocated right after another one was freed (the
	address range is typically handed out again).  Round 2 tried buffers mapped from physical chunks (hipMemCreate /
	hipMemMap) here: in the re-mapped buffer 115 of 1074 rows stayed zero on the first launch -- stale address
	translations (profiles/r02/output_buffer_vmm_reuse.txt) -- which is why the library allocates with hipMalloc only."""
	import ctypes
	g = synth.build_case(tmp_path, 93, 3_000_000, 2000, 3)
	vg = v2m.VariantGraph.from_object(g)
	ctx.upload_graph(vg, g.ref)
	L, pitch = g.aligned_length, ctx.min_row_pitch
	n_rows = (3 << 30) // pitch + 1
	rows = [v2m.PLOIDY_MAX if i % 7 == 0 else i % 6 for i in range(n_rows)]
	expected = {r: g.output_sequence(g.ref, copy_index=r) for r in set(rows)}
	want = {r: v2m.checksum_rows_host([body])[0] for r, body in expected.items()}
	want = np.array([want[r] for r in rows], dtype=np.uint64)
	from vcf2multialign_amd import _native
	rt = _native.hip_runtime()                  # (the runtime the process has loaded, not a versioned soname)
	rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
	for _ in range(3):
		out = ctx.alloc_output(n_rows * pitch, 2)
		ctx.splice_rows_device(rows, out, pitch)
		sums = ctx.checksum_rows_device(out, pitch, n_rows, length=L)
		assert np.array_equal(sums, want), np.nonzero(sums != want)[0][:20]
		for r in (0, n_rows // 2, n_rows - 1):
			buf = ctypes.create_string_buffer(L)
			assert rt.hipMemcpy(buf, out + r * pitch, L, 2) == 0
			assert buf.raw == expected[rows[r]], r
		ctx.free_output(out)


# ---- v2m_splice_rows_held: rows a sink may keep (ABI 5) -----------------------------------------------------------------------------

@pytest.mark.parametrize("unaligned", [False, True])
def test_held_rows_stay_valid_until_released(ctx, v2m, tmp_path, monkeypatch, unaligned):
	"""The sink returns at once and a pool of threads reads the rows later, releasing each when done: every row's bytes are still the
	oracle's when its reader gets to it, although the call has gone on through many more slices meanwhile (small slots: a few rows
	each, so slots are reused dozens of times)."""
	import ctypes as C
	import queue
	import threading
	import time
	g = synth.build_case(tmp_path, 97, 40000, 500, 40, mix=(0.6, 0.25, 0.15))          # 80 copies
	_upload(v2m, ctx, g)
	rows = [v2m.PLOIDY_MAX] + list(range(g.total_chromosome_copies))
	want = _oracle_rows(g, rows, unaligned=unaligned)
	monkeypatch.setenv("V2M_RING_SLOT_BYTES", str(3 * ((ctx.max_unaligned_length + 255) // 256 * 256) + 8))   # three rows per slot
	for n_slots, n_readers in ((2, 1), (3, 4), (8, 3)):
		got, todo = {}, queue.Queue()

		def reader():
			for row, ptr, length, hold in iter(todo.get, None):
				time.sleep(0.0005 * (row % 3))                                            # let the pipeline run ahead of the readers
				got[row] = C.string_at(ptr, length) if length else b""
				ctx.release_row(hold)

		readers = [threading.Thread(target=reader) for _ in range(n_readers)]
		for t in readers:
			t.start()
		try:
			ctx.splice_rows_held(rows, lambda row, ptr, length, hold: todo.put((row, ptr, length, hold)), n_slots=n_slots, unaligned=unaligned)
		finally:
			for _ in readers:
				todo.put(None)
			for t in readers:
				t.join()
		assert [got[r] for r in range(len(rows))] == want, (n_slots, n_readers)      # (the call returned: every row had been released)


def test_held_rows_refusals_and_errors(ctx, v2m, tmp_path, monkeypatch):
	g = synth.build_case(tmp_path, 98, 20000, 200, 10)
	_upload(v2m, ctx, g)
	rows = [v2m.PLOIDY_MAX] + list(range(g.total_chromosome_copies))
	monkeypatch.setenv("V2M_RING_SLOT_BYTES", str(2 * ((ctx.aligned_length + 255) // 256 * 256) + 8))
	seen = []

	def refuse_the_seventh(row, ptr, length, hold):
		if row == 7:
			return True                                                                  # not accepted: must not be released
		seen.append(row)
		ctx.release_row(hold)

	with pytest.raises(v2m.V2MError) as e:
		ctx.splice_rows_held(rows, refuse_the_seventh, n_slots=3)
	assert e.value.code == 7 and seen == list(range(7))
	with pytest.raises(v2m.V2MError) as e:                                               # fewer than two slots cannot overlap anything
		ctx.splice_rows_held(rows, lambda *a: None, n_slots=1)
	assert e.value.code == 1
	class Boom(Exception):
		pass
	def raises(row, ptr, length, hold):
		raise Boom()
	with pytest.raises(Boom):
		ctx.splice_rows_held(rows, raises, n_slots=2)
	# the context is still good for the plain form and for another held call
	assert ctx.splice_rows(rows) == _oracle_rows(g, rows)
	out = {}
	def keep(row, ptr, length, hold):
		import ctypes as C
		out[row] = C.string_at(ptr, length)
		ctx.release_row(hold)
	ctx.splice_rows_held(rows, keep, n_slots=2)
	assert [out[r] for r in range(len(rows))] == _oracle_rows(g, rows)
	ctx.splice_rows_held([], keep, n_slots=2)                                            # no row: nothing happens
