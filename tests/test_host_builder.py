"""The C++ host's graph builder (vcf2multialign_amd/csrc/host: readers.cc + graph_builder.cc) against the reference's
golden graph tables and against the oracle's independent builder.  CPU only: the final transpose is the GPU's."""

import json
import os

import numpy as np
import pytest

import oracle
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ARRAYS = ("reference_positions", "aligned_positions", "alt_edge_targets", "alt_edge_count_csum", "label_offsets", "paths_by_edge_and_chrom_copy", "ploidy_csum")


@pytest.fixture(scope="module")
def HostGraph():
	from vcf2multialign_amd import build
	build.build_native()
	from vcf2multialign_amd.host import HostGraph
	return HostGraph


def _same(o, h):
	for k in ARRAYS:
		assert np.array_equal(getattr(o, k), getattr(h, k)), k
	assert o.label_bytes == h.label_bytes
	assert o.sample_names == h.sample_names
	assert o.ref == h.ref
	assert o.paths_by_edge_and_chrom_copy_dims == h.paths_by_edge_and_chrom_copy_dims
	assert o.overlaps() == h.overlaps


def _goldens():
	with open(os.path.join(HERE, "golden", "reference_goldens.json")) as f:
		return json.load(f)["variant_graph"]


@pytest.mark.parametrize("case", _goldens(), ids=lambda c: c["vcf"])
def test_reference_graph_tables(HostGraph, case, fixtures_dir):   # tests/variant_graph.cc:247-339
	d = os.path.join(fixtures_dir, "variant-graph")
	h = HostGraph(os.path.join(d, case["fasta"]), os.path.join(d, case["vcf"]), case["chromosome"])
	assert len(h.reference_positions) == len(case["nodes"])
	o = h.label_offsets
	for n in case["nodes"]:
		i = n["node"]
		assert (int(h.reference_positions[i]), int(h.aligned_positions[i])) == (n["ref_pos"], n["aln_pos"])
		lo, hi = int(h.alt_edge_count_csum[i]), int(h.alt_edge_count_csum[i + 1])
		got = [{"target": int(h.alt_edge_targets[e]), "label": h.label_bytes[int(o[e]):int(o[e + 1])].decode()} for e in range(lo, hi)]
		assert got == n["alt_edges"]
	assert [{k: x[k] for k in ("sample", "chrom_copy_idx", "ref_pos", "var_id", "gt")} for x in h.overlaps] == case["expected_overlaps"]
	_same(oracle.build_variant_graph(os.path.join(d, case["fasta"]), os.path.join(d, case["vcf"]), case["chromosome"]), h)


@pytest.mark.parametrize("name", ["test-1", "test-1-2", "test-2", "test-3", "test-4"])
def test_founder_fixture_graphs(HostGraph, name, fixtures_dir):
	d = os.path.join(fixtures_dir, "founder-sequences")
	vcf = "test-1.vcf" if name == "test-1-2" else name + ".vcf"
	_same(oracle.build_variant_graph(os.path.join(d, name + ".fa"), os.path.join(d, vcf), "1"), HostGraph(os.path.join(d, name + ".fa"), os.path.join(d, vcf), "1"))


@pytest.mark.parametrize("seed,kw", [
	(101, {}), (102, {"mix": (1.0, 0.0, 0.0)}), (103, {"long_every": 25}), (104, {"multi_allelic": 0.4, "density": 0.4}),
	(105, {"ploidy": 1}), (106, {"ploidy": 3}),
])
def test_synthetic_vcfs_match_the_oracle_builder(HostGraph, tmp_path, seed, kw):
	rng = np.random.default_rng(seed)
	ref = synth.random_reference(rng, 30000)
	n_samples = 70 if seed == 101 else 9          # 140 copies: three 64-row words in the path matrix
	recs = synth.random_records(rng, ref, 700, n_samples, **kw)
	fa, vcf = synth.write_inputs(str(tmp_path), ref, recs, n_samples, phased=(seed % 2 == 0))
	_same(oracle.build_variant_graph(fa, vcf, "1"), HostGraph(fa, vcf, "1"))


def test_other_chromosomes_and_missing_alleles(HostGraph, tmp_path):
	ref = b"ACGTACGTACGTACGTACGT"
	vcf = tmp_path / "x.vcf"
	fa = tmp_path / "x.fa"
	fa.write_bytes(b">chrA some description\n" + ref[:10] + b"\n" + ref[10:] + b"\n>chrB\nTTTT\n")
	vcf.write_text(
		"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\n"
		"chrB\t1\tb1\tT\tA\t.\t.\t.\tGT\t1|1\t1|1\n"
		"chrA\t3\ta1;rs1\tG\tT,<DEL>,*\t.\t.\t.\tGT:DP\t1|.:3\t./2:4\n"
		"chrA\t7\ta2\tGT\tG\t.\t.\t.\tDP:GT\t9:0/1\t9:1|0\n"
		"chrB\t2\tb2\tT\tA\t.\t.\t.\tGT\t1|1\t1|1\n")
	o = oracle.build_variant_graph(str(fa), str(vcf), "chrA", seq_id="chrA")
	h = HostGraph(str(fa), str(vcf), "chrA", seq_id="chrA")
	_same(o, h)
	assert h.handled_variants == 2 and h.chr_id_mismatches == 2
	assert len(h.alt_edge_targets) == 3            # T, <DEL>, and the deletion G; '*' makes no edge
	assert h.ref == ref


def test_sample_exclusion(HostGraph, tmp_path):   # variant_graph.cc:215-273
	g = synth.build_case(tmp_path, 107, 8000, 150, 5)
	fa, vcf = os.path.join(str(tmp_path), "synth.fa"), os.path.join(str(tmp_path), "synth.vcf")
	for sample, copy in (("S2", -1), ("S0", 1), ("S4", 0)):
		o = oracle.build_variant_graph(fa, vcf, "1", exclude_sample=sample, exclude_copy=copy)
		h = HostGraph(fa, vcf, "1", exclude_sample=sample, exclude_copy=copy)
		_same(o, h)
		assert int(h.ploidy_csum[-1]) == (8 if copy < 0 else 9)
		assert ("S2" in h.sample_names) == (sample != "S2")
	assert g.total_chromosome_copies == 10


def test_every_copy_excluded(HostGraph, tmp_path):
	"""A haploid one-sample VCF with that sample's only copy excluded (variant_graph.cc:215-273): no chromosome copy is left, the path
	matrix has no rows, and the graph -- nodes, edges, labels -- is still built, on one thread and on several (round 3's chunked merge
	refused it: the parser counts a column per ALT edge, the builder grows none for a matrix without rows)."""
	records = [(2, b"G", [b"T"], np.array([[1]])), (5, b"CG", [b"C"], np.array([[1]])), (9, b"C", [b"G", b"CAA"], np.array([[2]]))]
	fa, vcf = synth.write_inputs(str(tmp_path), b"ACGTACGTACGTACGT", records, 1)      # one genotype column per sample: haploid
	for threads in (1, 3):
		o = oracle.build_variant_graph(fa, vcf, "1", exclude_sample="S0", exclude_copy=0)
		h = HostGraph(fa, vcf, "1", exclude_sample="S0", exclude_copy=0, threads=threads)
		for k in ARRAYS:
			if not k.startswith("paths_"):
				assert np.array_equal(getattr(o, k), getattr(h, k)), k
		assert o.label_bytes == h.label_bytes and o.sample_names == h.sample_names and o.overlaps() == h.overlaps
		# the reference keeps a 1 x 0 matrix here (variant_graph.cc:280, never resized: :445 tests size()), the host builder a 0 x Ep one:
		# either way a matrix that holds no word, and an empty transpose (transpose_matrix.cc:48-49)
		assert o.paths_by_edge_and_chrom_copy_dims == (1, 0) and h.paths_by_edge_and_chrom_copy_dims[0] == 0
		assert o.paths_by_edge_and_chrom_copy.size == 0 and h.paths_by_edge_and_chrom_copy.size == 0
		assert o.paths_by_chrom_copy_and_edge.size == 0          # (the host graph's transpose is the GPU's to make)
		assert int(h.ploidy_csum[-1]) == 0 and len(h.alt_edge_targets) == 4
	full = HostGraph(fa, vcf, "1")
	assert int(full.ploidy_csum[-1]) == 1 and len(full.alt_edge_targets) == 4


def test_malformed_input_is_an_error(HostGraph, tmp_path):
	fa, vcf = synth.write_inputs(str(tmp_path), b"ACGTACGT", [(5, b"C", [b"T"], np.array([[1, 0]])), (2, b"G", [b"T"], np.array([[1, 0]]))], 1)
	with pytest.raises(ValueError):
		HostGraph(fa, vcf, "1")                     # non-increasing positions (variant_graph.cc:292-297)
	fa, vcf = synth.write_inputs(str(tmp_path), b"ACGTACGT", [(2, b"T", [b"A"], np.array([[1, 0]]))], 1)
	with pytest.raises(ValueError):
		HostGraph(fa, vcf, "1")                     # REF column mismatch (delegate decides; the test delegate refuses)


def test_delegate_that_stops_at_a_ref_mismatch(HostGraph, tmp_path):
	"""variant_graph.cc:307-314: a delegate may answer a REF mismatch with "stop here".  The graph then holds the records before it,
	a node at the mismatching record's position and the sink (:437-451), and the chromosome-ID mismatches counted are those
	seen before the stop (:203-207).  A record that goes BACK in position is an error before its REF column is looked at (:293-297)."""
	import ctypes
	from vcf2multialign_amd import host
	L = host._load()
	L.v2mh_set_stop_at_ref_mismatch.argtypes = [ctypes.c_int]
	ref = b"ACGTACGTACGTACGT"
	gt = np.array([[1, 0]])
	def write(records):
		fa, vcf = synth.write_inputs(str(tmp_path), ref, records, 1)
		lines = open(vcf).read().splitlines()
		body = [l for l in lines if not l.startswith("#")]
		other = body[0].split("\t"); other[0] = "2"                       # records of another chromosome, before and after the stop
		open(vcf, "w").write("\n".join([l for l in lines if l.startswith("#")] + ["\t".join(other), body[0], "\t".join(other)] + body[1:] + ["\t".join(other)]) + "\n")
		return fa, vcf
	try:
		L.v2mh_set_stop_at_ref_mismatch(1)
		fa, vcf = write([(1, b"C", [b"T"], gt), (6, b"T", [b"A"], gt), (9, b"C", [b"G"], gt)])   # (0-based) position 6 holds G, not T
		h = HostGraph(fa, vcf, "1")
		assert h.reference_positions.tolist() == [0, 1, 2, 6, 16]            # source, record 1 and its target, the stopping record's node, sink
		assert len(h.alt_edge_targets) == 1 and h.handled_variants == 2 and h.chr_id_mismatches == 2
		fa, vcf = write([(6, b"G", [b"T"], gt), (3, b"A", [b"C"], gt)])     # goes back AND has the wrong REF (position 3 holds T): the position wins
		with pytest.raises(ValueError, match="non-increasing"):
			HostGraph(fa, vcf, "1")
	finally:
		L.v2mh_set_stop_at_ref_mismatch(0)


@pytest.mark.parametrize("threads", [1, 2, 5])
def test_threaded_parse_is_identical_to_sequential(HostGraph, tmp_path, threads):
	"""Workers parse 8-MB chunks, one thread merges in file order: same graph, same overlap reports in the same order."""
	rng = np.random.default_rng(120)
	ref = synth.random_reference(rng, 400000)
	recs = synth.random_records(rng, ref, 9000, 500, multi_allelic=0.1, long_every=300)   # ~19 MB of VCF text: three chunks
	fa, vcf = synth.write_inputs(str(tmp_path), ref, recs, 500)
	assert os.path.getsize(vcf) > 17 * (1 << 20)
	o = oracle.build_variant_graph(fa, vcf, "1")
	h = HostGraph(fa, vcf, "1", threads=threads)
	_same(o, h)
	assert len(h.overlaps) > 10


def test_text_pipeline_equals_direct_generator(HostGraph, tmp_path):
	"""The synthetic generator pushes records straight into graph_builder and derives genotype bits from a hash; writing
	the same dataset as FASTA + VCF text and parsing it back must give the same graph and the same path bits."""
	from vcf2multialign_amd import synth as vsynth
	ds = vsynth.dataset("mini3")
	fa, vcf = str(tmp_path / "m.fa"), str(tmp_path / "m.vcf")
	ds.write_fasta_and_vcf(fa, vcf)
	h = HostGraph(fa, vcf, "1")
	g = ds.graph
	for k in ("reference_positions", "aligned_positions", "alt_edge_targets", "alt_edge_count_csum", "label_offsets"):
		assert np.array_equal(getattr(g, k), getattr(h, k)), k
	assert g.label_bytes == h.label_bytes and h.ref == ds.reference
	hp, ep = h.paths_by_edge_and_chrom_copy_dims
	assert (hp, ep) == (ds.path_cols, ds.path_rows)
	m = h.paths_by_edge_and_chrom_copy.reshape(ep, hp // 64)
	for copy in (0, 1, 63, 64, 199):
		bits = ((m[:, copy // 64] >> np.uint64(copy % 64)) & np.uint64(1)).astype(np.uint8)
		assert np.array_equal(bits, np.unpackbits(ds.copy_column(copy).view(np.uint8), bitorder="little")[:ep]), copy


def test_errors_surface_in_file_order(HostGraph, tmp_path):
	ref = b"ACGTACGTACGTACGTACGT"
	vcf = tmp_path / "bad.vcf"
	fa = tmp_path / "bad.fa"
	fa.write_bytes(b">1\n" + ref + b"\n")
	vcf.write_text(
		"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n"
		"1\t3\ta\tG\tT\t.\t.\t.\tGT\t1|0\n"
		"1\t5\tb\tA\tT\t.\t.\t.\tGT\t1|x\n"
		"1\t2\tc\tC\tT\t.\t.\t.\tGT\t1|0\n")
	with pytest.raises(ValueError) as e:
		HostGraph(str(fa), str(vcf), "1", threads=3)
	assert "line 4" in str(e.value) and "bad GT allele" in str(e.value)


def test_graph_file_round_trip(HostGraph, tmp_path):
	"""--output-graph / --input-graph (our flat format in place of the reference's cereal archive, main.cc:393-426)."""
	g = synth.build_case(tmp_path, 130, 20000, 400, 70, multi_allelic=0.2)   # 140 copies
	fa, vcf = str(tmp_path / "synth.fa"), str(tmp_path / "synth.vcf")
	h = HostGraph(fa, vcf, "1")
	path = tmp_path / "g.graph"
	h.write(path)
	r = HostGraph.read(path)
	for k in ARRAYS:
		assert np.array_equal(getattr(h, k), getattr(r, k)), k
	assert h.label_bytes == r.label_bytes and h.sample_names == r.sample_names
	assert h.paths_by_edge_and_chrom_copy_dims == r.paths_by_edge_and_chrom_copy_dims
	assert g.edge_count == len(r.alt_edge_targets)
	data = bytearray(path.read_bytes())
	data[len(data) // 2] ^= 0x40
	bad = tmp_path / "bad.graph"
	bad.write_bytes(bytes(data))
	with pytest.raises(ValueError, match="checksum"):
		HostGraph.read(bad)
	bad.write_bytes(bytes(data[:len(data) // 3]))
	with pytest.raises(ValueError):
		HostGraph.read(bad)
	bad.write_bytes(b"not a graph file at all, just text\n" * 10)
	with pytest.raises(ValueError, match="V2MGRAF1"):
		HostGraph.read(bad)
	# counts that do not fit the file are refused before anything is allocated from them (here: 2^39 edges claimed)
	good = path.read_bytes()
	huge = bytearray(good)
	huge[8 + 8:8 + 16] = (1 << 39).to_bytes(8, "little")
	bad.write_bytes(bytes(huge))
	with pytest.raises(ValueError, match="do not match the size of the file"):
		HostGraph.read(bad)
	bad.write_bytes(good + b"\0" * 8)
	with pytest.raises(ValueError, match="do not match the size of the file"):
		HostGraph.read(bad)
	# ... also when the claimed sizes wrap 64 bits: 2^40 x 2^30 bits is 2^67 bytes = 0 mod 2^64, so with that matrix's block cut
	# out of the file the unchecked sum would match the file size again and leave a matrix without words (round-2 advisor)
	c = [int.from_bytes(good[8 + 8 * i:16 + 8 * i], "little") for i in range(10)]
	matrix_at = 8 + 80 + 8 * (2 * c[0] + c[1] + (c[0] + 1) + (c[1] + 1)) + c[6] // 64 * c[7] * 8   # paths_by_edge_and_chrom_copy (the host leaves the other one out)
	matrix_bytes = c[8] // 64 * c[9] * 8
	assert matrix_bytes > 0
	wrapping = bytearray(good[:matrix_at] + good[matrix_at + matrix_bytes:])
	wrapping[8 + 64:8 + 72] = (1 << 40).to_bytes(8, "little")
	wrapping[8 + 72:8 + 80] = (1 << 30).to_bytes(8, "little")
	bad.write_bytes(bytes(wrapping))
	with pytest.raises(ValueError, match="do not match the size of the file"):
		HostGraph.read(bad)


def test_graph_file_with_blocks_read_by_several_threads(HostGraph, tmp_path):
	"""A path matrix of 96 MiB: read_graph() fetches and checksums it in 32-MiB parts on several threads; the file, the
	arrays and the verdict on a flipped bit in the last part are what the single-threaded reader gives."""
	from types import SimpleNamespace
	n_nodes, n_edges, n_samples, ploidy = 3, 2, 1024 * 3, 2                   # 6144 copies x 16384-bit columns = 96 MiB
	rng = np.random.default_rng(5)
	hp, ep = n_samples * ploidy, 16384 * 8
	words = rng.integers(0, 2 ** 63, size=ep // 64 * hp, dtype=np.uint64)
	g = SimpleNamespace(reference_positions=[0, 1, 2], aligned_positions=[0, 1, 2], alt_edge_targets=[1, 2], alt_edge_count_csum=[0, 1, 2, 2],
		label_offsets=[0, 1, 2], label_bytes=b"AC")
	h = HostGraph.from_arrays(g, words, hp, ep, n_samples, ploidy)
	path = tmp_path / "big.graph"
	h.write(path)
	r = HostGraph.read(path)
	assert np.array_equal(r.paths_by_edge_and_chrom_copy, words) and r.paths_by_edge_and_chrom_copy_dims == (hp, ep)
	assert list(r.alt_edge_targets) == [1, 2] and r.label_bytes == b"AC" and len(r.sample_names) == n_samples
	size = os.path.getsize(path)
	with open(path, "r+b") as f:
		f.seek(size - 40 * 1024 * 1024)
		b = f.read(1)
		f.seek(size - 40 * 1024 * 1024)
		f.write(bytes([b[0] ^ 1]))
	with pytest.raises(ValueError, match="checksum"):
		HostGraph.read(path)
