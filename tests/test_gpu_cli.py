"""End to end through the C++ host (vcf2multialign_amd/bin/vcf2multialign): FASTA + VCF in, A2M out, every row
spliced on the GPU, compared byte for byte with the oracle / the derived goldens."""

import os
import subprocess

import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLI = os.path.join(ROOT, "vcf2multialign_amd", "bin", "vcf2multialign")
FIX = os.path.join(HERE, "golden", "reference-fixtures", "variant-graph")
DERIVED = os.path.join(HERE, "golden", "derived")


def run(args, cwd=None):
	assert os.path.exists(CLI), "build the host driver first (__graft_entry__.build())"
	return subprocess.run([CLI] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


@pytest.mark.parametrize("stem,fasta", [("test-1a", "test-1.fa"), ("test-1b", "test-1.fa"), ("test-2", "test-2.fa"), ("test-3", "test-3.fa"), ("test-4", "test-4.fa")])
def test_haplotypes_on_reference_fixtures(tmp_path, stem, fasta):
	common = ["--haplotypes", "--input-reference=" + os.path.join(FIX, fasta), "--input-variants=" + os.path.join(FIX, stem + ".vcf"), "--chromosome=1"]
	out = tmp_path / "out.a2m"
	r = run(common + ["--output-sequences-a2m=" + str(out)])
	assert r.returncode == 0, r.stderr.decode()
	assert out.read_bytes() == open(os.path.join(DERIVED, stem + ".haplotypes.a2m"), "rb").read()
	if stem.startswith("test-1"):   # the expected overlap of tests/variant_graph.cc:267, in main.cc:168-170's wording
		assert b"Overlapping alternative alleles. Line number: 8 current variant position: 9 variant identifiers: a5 sample: SAMPLE2 chromosome copy: 0 genotype: 1\n" == r.stdout
	r = run(common + ["-s", str(out), "--unaligned"])
	assert r.returncode == 0, r.stderr.decode()
	assert out.read_bytes() == open(os.path.join(DERIVED, stem + ".haplotypes.unaligned.fa"), "rb").read()
	r = run(common + ["-s", str(out), "--omit-reference", "--dst-chromosome=chrT"])
	assert r.returncode == 0, r.stderr.decode()
	assert out.read_bytes() == open(os.path.join(DERIVED, stem + ".haplotypes.chr.noref.a2m"), "rb").read()


def test_two_diploid_samples_config1(tmp_path):
	"""BASELINE config 1: tiny ref + VCF, --haplotypes, 2 diploid samples (a two-sample column subset of test-1a)."""
	lines = open(os.path.join(FIX, "test-1a.vcf")).read().splitlines()
	vcf = tmp_path / "two.vcf"
	vcf.write_text("\n".join(l if l.startswith("##") else "\t".join(l.split("\t")[:11]) for l in lines) + "\n")
	fa = os.path.join(FIX, "test-1.fa")
	out = tmp_path / "out.a2m"
	r = run(["-H", "-r", fa, "-a", str(vcf), "-c", "1", "-s", str(out), "--output-graph-statistics"])
	assert r.returncode == 0, r.stderr.decode()
	g = oracle.build_variant_graph(fa, str(vcf), "1")
	exp = tmp_path / "exp.a2m"
	g.haplotype_output_a2m(g.ref, str(exp))
	assert out.read_bytes() == exp.read_bytes()
	assert out.read_bytes().count(b">") == 5
	assert b"Total ploidy: 4" in r.stdout


def test_separate_files_and_sample_filter(tmp_path):
	g = synth.build_case(tmp_path, 61, 20000, 300, 4)
	fa, vcf = str(tmp_path / "synth.fa"), str(tmp_path / "synth.vcf")
	excl = tmp_path / "excl.tsv"
	excl.write_text("1\tS1\t0\n1\tS1\t1\n2\tS0\t0\n")
	wd = tmp_path / "sep"
	wd.mkdir()
	r = run(["-H", "-r", fa, "-a", vcf, "-c", "1", "--output-sequences-separate", "-x", str(excl)], cwd=str(wd))
	assert r.returncode == 0, r.stderr.decode()
	go = oracle.build_variant_graph(fa, vcf, "1", exclude_sample="S1")
	names = sorted(os.listdir(wd))
	assert names == sorted(["REF.a2m"] + ["S%d.%d.a2m" % (s, c) for s in (0, 2, 3) for c in (1, 2)])
	# output_sequence_file passes the file name as FASTA id and no newline follows the body (output.cc:36,42)
	assert (wd / "REF.a2m").read_bytes() == b">REF.a2m\n" + go.output_sequence(go.ref)
	assert (wd / "S2.2.a2m").read_bytes() == b">S2.2.a2m\n" + go.output_sequence(go.ref, copy_index=3)
	assert g.total_chromosome_copies == 8 and go.total_chromosome_copies == 6


def test_every_chromosome_copy_excluded(tmp_path):
	"""A haploid one-sample VCF with that sample's only copy excluded: no chromosome copy is left, the path matrix has no rows, and
	the A2M holds the REF row alone -- aligned (gaps where the excluded copy's insertion would go) and unaligned.  (Round 3's reader
	refused this input; the graph and its edges are still built: variant_graph.cc:215-283.)"""
	import numpy as np
	records = [(2, b"G", [b"T"], np.array([[1]])), (5, b"CG", [b"C"], np.array([[1]])), (9, b"C", [b"G", b"CAA"], np.array([[2]]))]
	fa, vcf = synth.write_inputs(str(tmp_path), b"ACGTACGTACGTACGT", records, 1)
	excl = tmp_path / "excl.tsv"
	excl.write_text("1\tS0\t0\n")
	go = oracle.build_variant_graph(fa, vcf, "1", exclude_sample="S0", exclude_copy=0)
	assert go.total_chromosome_copies == 0 and go.edge_count == 4
	out = tmp_path / "out.a2m"
	for extra, unaligned in (([], False), (["--unaligned"], True)):
		r = run(["-H", "-r", fa, "-a", vcf, "-c", "1", "-s", str(out), "-x", str(excl), "--output-graph-statistics"] + extra)
		assert r.returncode == 0, r.stderr.decode()
		assert out.read_bytes() == b">REF\n" + go.output_sequence(go.ref, unaligned=unaligned) + b"\n"
		assert b"Total ploidy: 0" in r.stdout
	assert b"-" in go.output_sequence(go.ref) and b"-" not in go.output_sequence(go.ref, unaligned=True)


def test_one_megabase_end_to_end(tmp_path):
	g = synth.build_case(tmp_path, 62, 1000000, 12000, 40, long_every=500)
	fa, vcf = str(tmp_path / "synth.fa"), str(tmp_path / "synth.vcf")
	out, exp = tmp_path / "out.a2m", tmp_path / "exp.a2m"
	r = run(["-H", "-r", fa, "-a", vcf, "-c", "1", "-s", str(out)])
	assert r.returncode == 0, r.stderr.decode()
	g.haplotype_output_a2m(g.ref, str(exp))
	assert out.stat().st_size == exp.stat().st_size == sum(len(">%s\n" % i) for i in ["REF"] + ["S%d-%d" % (s, c) for s in range(40) for c in (1, 2)]) + 81 * (g.aligned_length + 1)
	assert out.read_bytes() == exp.read_bytes()


def _founder_goldens():
	import json
	with open(os.path.join(HERE, "golden", "reference_goldens.json")) as f:
		return json.load(f)["founder_sequences"]


@pytest.mark.parametrize("case", _founder_goldens(), ids=lambda c: c["vcf"] + "+" + c["fasta"])
def test_founder_sequences_end_to_end(tmp_path, case):
	"""tests/founder_sequences.cc:78-114 through the product: build -> find_cut_positions(0) -> find_matchings(2) on
	the host, rows on the GPU, the complete expected A2M text of the reference's test."""
	d = os.path.join(HERE, "golden", "reference-fixtures", "founder-sequences")
	out = tmp_path / "founders.a2m"
	r = run(["--founder-sequences=2", "--minimum-distance=0", "-r", os.path.join(d, case["fasta"]), "-a", os.path.join(d, case["vcf"]), "-c", "1", "-s", str(out), "--verbose"])
	assert r.returncode == 0, r.stderr.decode()
	assert out.read_bytes().decode() == case["expected_a2m"]
	assert ("Cut positions: " + " ".join(str(c) for c in case["cut_positions"]) + "\n").encode() in r.stdout
	assert b"Maximum segmentation height: " in r.stdout


def test_founders_on_a_larger_graph(tmp_path):
	"""25 founders over a 200 kb synthetic graph: every founder row must be a mosaic of sample rows that switches only
	at the reported cut positions (checked against the oracle's walk with the same cuts)."""
	g = synth.build_case(tmp_path, 63, 200000, 2500, 20)
	fa, vcf = str(tmp_path / "synth.fa"), str(tmp_path / "synth.vcf")
	out = tmp_path / "founders.a2m"
	r = run(["-F", "25", "-d", "50", "-r", fa, "-a", vcf, "-c", "1", "-s", str(out), "--verbose"])
	assert r.returncode == 0, r.stderr.decode()
	stdout = r.stdout.decode().splitlines()
	cuts = [int(x) for x in next(l for l in stdout if l.startswith("Cut positions:")).split()[2:]]
	cols = [[int(x) for x in l.split("\t")[1:]] for l in stdout[stdout.index("Matchings:") + 1:stdout.index("Matchings:") + 26]]
	assert cuts[0] == 0 and cuts[-1] == g.node_count - 1 and all(b - a > 0 for a, b in zip(cuts, cuts[1:]))
	assert all(int(g.aligned_positions[b]) - int(g.aligned_positions[a]) >= 50 for a, b in zip(cuts[:-2], cuts[1:-1]))
	lines = out.read_bytes().split(b"\n")
	assert lines[0] == b">REF" and lines[1] == g.output_sequence(g.ref)
	for f, col in enumerate(cols):
		assert len(col) == len(cuts) - 1
		assert lines[2 + 2 * f] == b">%d" % (1 + f)
		assert lines[3 + 2 * f] == g.output_sequence(g.ref, cuts=list(zip(cuts[:-1], col))), "founder %d" % (1 + f)


def test_sharded_output_over_several_contexts(tmp_path):
	"""--device=0,0,0: three contexts (here on the same GPU), rows sharded over them, every thread writing its rows at
	their final file offsets: the file must equal the single-context one, for haplotypes and for founders."""
	g = synth.build_case(tmp_path, 65, 120000, 1800, 25, long_every=300)
	fa, vcf = str(tmp_path / "synth.fa"), str(tmp_path / "synth.vcf")
	one, three = tmp_path / "one.a2m", tmp_path / "three.a2m"
	assert run(["-H", "-r", fa, "-a", vcf, "-c", "1", "-s", str(one)]).returncode == 0
	r = run(["-H", "-r", fa, "-a", vcf, "-c", "1", "-s", str(three), "--device=0,0,0"])
	assert r.returncode == 0, r.stderr.decode()
	exp = tmp_path / "exp.a2m"
	g.haplotype_output_a2m(g.ref, str(exp))
	assert one.read_bytes() == exp.read_bytes() == three.read_bytes()
	# each context received only its own chromosome copies: 25 diploid samples = 50 copies in blocks of
	# 8, the blocks that do not divide go to the last contexts -- the same boundaries as sharding.py
	from vcf2multialign_amd.sharding import shard_copies
	r = run(["-H", "-r", fa, "-a", vcf, "-c", "1", "-s", str(three), "--device=0,0,0", "--verbose"])
	assert r.returncode == 0, r.stderr.decode()
	for k in range(3):
		c0, c1, _ = shard_copies(50, 3, k)
		assert ("GPU context %d (device 0): chromosome copies [%d, %d)" % (k, c0, c1)).encode() in r.stderr, r.stderr.decode()
	assert three.read_bytes() == exp.read_bytes()
	# more contexts than blocks of copies: some contexts own nothing
	r = run(["-H", "-r", fa, "-a", vcf, "-c", "1", "-s", str(three), "--device=0,0,0,0,0,0,0,0,0", "--omit-reference"])
	assert r.returncode == 0, r.stderr.decode()
	g.haplotype_output_a2m(g.ref, str(exp), output_reference=False)
	assert three.read_bytes() == exp.read_bytes()
	g.haplotype_output_a2m(g.ref, str(exp))
	assert run(["-F", "7", "-r", fa, "-a", vcf, "-c", "1", "-s", str(one), "--dst-chromosome=chrQ"]).returncode == 0
	r = run(["-F", "7", "-r", fa, "-a", vcf, "-c", "1", "-s", str(three), "--dst-chromosome=chrQ", "--device=0,0"])
	assert r.returncode == 0, r.stderr.decode()
	assert one.read_bytes() == three.read_bytes() and one.read_bytes().count(b">chrQ\t") == 8


def test_ordered_outputs_over_several_contexts(tmp_path):
	"""--unaligned, --pipe and --output-sequences-separate with --device=0,0,0: the outputs that have to leave in row order
	(sequence_writer.cc:80, output.cc:47-76, haplotype_output.cc:85-132).  The chromosome copies are dealt to the contexts
	round-robin in blocks of 8 (no context holds the whole matrix), every context splices its rows on its own thread and the
	rows reach the one writer in turns: byte-identical to one context."""
	g = synth.build_case(tmp_path, 66, 150000, 2200, 27, long_every=300)           # 54 copies: 7 blocks of 8, the last one short
	fa, vcf = str(tmp_path / "synth.fa"), str(tmp_path / "synth.vcf")
	common = ["-H", "-r", fa, "-a", vcf, "-c", "1"]
	one, three, exp = tmp_path / "one.fa", tmp_path / "three.fa", tmp_path / "exp.fa"
	env_small_slots = dict(os.environ, V2M_RING_SLOT_BYTES=str(400000))              # 2 rows per slice: many slices per context, turns in the middle of them
	# unaligned A2M file
	g.haplotype_output_a2m(g.ref, str(exp), unaligned=True)
	assert run(common + ["-s", str(one), "--unaligned"]).returncode == 0
	r = run(common + ["-s", str(three), "--unaligned", "--device=0,0,0", "--verbose"])
	assert r.returncode == 0, r.stderr.decode()
	assert one.read_bytes() == exp.read_bytes() == three.read_bytes()
	for k in range(3):
		assert ("GPU context %d (device 0): chromosome copies %d + 24 j + [0, 8), j = 0, 1, ..." % (k, 8 * k)).encode() in r.stderr, r.stderr.decode()
	three.unlink()
	r = subprocess.run([CLI] + common + ["-s", str(three), "--unaligned", "--device=0,0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env_small_slots)
	assert r.returncode == 0 and three.read_bytes() == exp.read_bytes()
	# more contexts than blocks; no REF row
	g.haplotype_output_a2m(g.ref, str(exp), unaligned=True, output_reference=False)
	r = run(common + ["-s", str(three), "--unaligned", "--omit-reference", "--device=0,0,0,0,0,0,0,0,0"])
	assert r.returncode == 0 and three.read_bytes() == exp.read_bytes()
	# aligned, through a pipe
	script = tmp_path / "sink.sh"
	script.write_text("#!/bin/sh\ncat > \"$1.piped\"\n")
	script.chmod(0o755)
	g.haplotype_output_a2m(g.ref, str(exp))
	r = subprocess.run([CLI] + common + ["-s", str(tmp_path / "p.a2m"), "--pipe=" + str(script), "--device=0,0,0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env_small_slots)
	assert r.returncode == 0, r.stderr.decode()
	assert (tmp_path / "p.a2m.piped").read_bytes() == exp.read_bytes()
	# a reader that goes away must end the run (every context's thread), not hang it
	quitter = tmp_path / "quit.sh"
	quitter.write_text("#!/bin/sh\nhead -c 10 > /dev/null\nexit 3\n")
	quitter.chmod(0o755)
	r = run(common + ["-s", str(tmp_path / "x.a2m"), "--pipe=" + str(quitter), "--device=0,0,0"])
	assert r.returncode != 0 and b"exited with status 3" in r.stderr
	# one file per sequence (and the aligned A2M file in the same run: then it is written in turns as well)
	(tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
	assert run(common + ["--output-sequences-separate", "-s", "all.a2m"], cwd=str(tmp_path / "a")).returncode == 0
	r = run(common + ["--output-sequences-separate", "-s", "all.a2m", "--device=0,0,0"], cwd=str(tmp_path / "b"))
	assert r.returncode == 0, r.stderr.decode()
	names = sorted(os.listdir(tmp_path / "a"))
	assert len(names) == 56 and sorted(os.listdir(tmp_path / "b")) == names
	for n in names:
		assert (tmp_path / "b" / n).read_bytes() == (tmp_path / "a" / n).read_bytes(), n
	assert (tmp_path / "b" / "all.a2m").read_bytes() == exp.read_bytes()


def test_graph_checkpoint(tmp_path):
	"""--output-graph then --input-graph: the second run skips the VCF and writes the same A2M (both modes)."""
	g = synth.build_case(tmp_path, 64, 60000, 900, 12, long_every=200)
	fa, vcf = str(tmp_path / "synth.fa"), str(tmp_path / "synth.vcf")
	graph, a, b = tmp_path / "g.graph", tmp_path / "a.a2m", tmp_path / "b.a2m"
	r = run(["-H", "-r", fa, "-a", vcf, "-c", "1", "-s", str(a), "--output-graph=" + str(graph)])
	assert r.returncode == 0, r.stderr.decode()
	r = run(["-H", "-r", fa, "--input-graph=" + str(graph), "-s", str(b)])
	assert r.returncode == 0, r.stderr.decode()
	assert b"Building the variant graph" not in r.stderr
	exp = tmp_path / "exp.a2m"
	g.haplotype_output_a2m(g.ref, str(exp))
	assert a.read_bytes() == exp.read_bytes() == b.read_bytes()
	r1 = run(["-F", "5", "-r", fa, "-a", vcf, "-c", "1", "-s", str(a)])
	r2 = run(["-F", "5", "-r", fa, "-g", str(graph), "-s", str(b)])
	assert r1.returncode == 0 and r2.returncode == 0 and a.read_bytes() == b.read_bytes() and r1.stdout == r2.stdout
	assert run(["-H", "-r", fa, "-g", str(graph), "-a", vcf, "-c", "1", "-s", str(b)]).returncode != 0


def test_pipe(tmp_path):
	"""--pipe=command (output.cc:26-38,49-68): `command <name>` reads what would have been written to <name>."""
	fa, vcf = os.path.join(FIX, "test-2.fa"), os.path.join(FIX, "test-2.vcf")
	script = tmp_path / "sink.sh"
	script.write_text("#!/bin/sh\ncat > \"$1.piped\"\n")
	script.chmod(0o755)
	common = ["-H", "-r", fa, "-a", vcf, "-c", "1"]
	direct = tmp_path / "direct.a2m"
	assert run(common + ["-s", str(direct)]).returncode == 0
	r = run(common + ["-s", str(tmp_path / "out.a2m"), "--pipe=" + str(script)])
	assert r.returncode == 0, r.stderr.decode()
	assert not (tmp_path / "out.a2m").exists()
	assert (tmp_path / "out.a2m.piped").read_bytes() == direct.read_bytes()

	# one subprocess per sequence with --output-sequences-separate
	(tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
	assert run(common + ["--output-sequences-separate"], cwd=str(tmp_path / "a")).returncode == 0
	r = run(common + ["--output-sequences-separate", "--pipe=" + str(script)], cwd=str(tmp_path / "b"))
	assert r.returncode == 0, r.stderr.decode()
	names = sorted(os.listdir(tmp_path / "a"))
	assert len(names) == 7 and sorted(os.listdir(tmp_path / "b")) == [n + ".piped" for n in names]
	for n in names:
		assert (tmp_path / "b" / (n + ".piped")).read_bytes() == (tmp_path / "a" / n).read_bytes()

	# founder mode through a pipe, several MB per row (more than one pipe buffer and more than the stream buffer)
	g = synth.build_case(tmp_path, 64, 3_000_000, 3000, 4)
	big = ["-F", "3", "-r", str(tmp_path / "synth.fa"), "-a", str(tmp_path / "synth.vcf"), "-c", "1"]
	assert run(big + ["-s", str(tmp_path / "f.a2m")]).returncode == 0
	assert run(big + ["-s", str(tmp_path / "g.a2m"), "--pipe=" + str(script)]).returncode == 0
	assert (tmp_path / "g.a2m.piped").read_bytes() == (tmp_path / "f.a2m").read_bytes()
	assert (tmp_path / "f.a2m").stat().st_size > 4 * g.aligned_length

	# failures: a command that cannot be executed, one that exits non-zero, one that stops reading
	r = run(common + ["-s", str(tmp_path / "x.a2m"), "--pipe=/nonexistent/command"])
	assert r.returncode != 0 and b"Unable to execute subprocess" in r.stderr
	r = run(common + ["-s", str(tmp_path / "x.a2m"), "--pipe=false"])
	assert r.returncode != 0 and b"exited with status 1" in r.stderr
	quitter = tmp_path / "quit.sh"
	quitter.write_text("#!/bin/sh\nhead -c 10 > /dev/null\nexit 3\n")
	quitter.chmod(0o755)
	r = run(big + ["-s", str(tmp_path / "x.a2m"), "--pipe=" + str(quitter)])
	assert r.returncode != 0 and b"exited with status 3" in r.stderr


def test_cut_position_files(tmp_path):
	"""--output-cut-positions, then --input-cut-positions instead of the search: same founders."""
	synth.build_case(tmp_path, 65, 100000, 1500, 8)
	common = ["-F", "4", "-r", str(tmp_path / "synth.fa"), "-a", str(tmp_path / "synth.vcf"), "-c", "1"]
	r1 = run(common + ["-d", "40", "-s", str(tmp_path / "a.a2m"), "-t", str(tmp_path / "cuts.bin")])
	assert r1.returncode == 0, r1.stderr.decode()
	r2 = run(common + ["-s", str(tmp_path / "b.a2m"), "--input-cut-positions=" + str(tmp_path / "cuts.bin"), "--output-cut-positions=" + str(tmp_path / "cuts2.bin")])
	assert r2.returncode == 0, r2.stderr.decode()
	assert b"Optimising cut positions" in r1.stderr and b"Optimising cut positions" not in r2.stderr
	assert (tmp_path / "a.a2m").read_bytes() == (tmp_path / "b.a2m").read_bytes()
	assert (tmp_path / "cuts.bin").read_bytes() == (tmp_path / "cuts2.bin").read_bytes()     # min_distance and score are carried over
	assert r1.stdout == r2.stdout                                                             # "Maximum segmentation height: ..."
	from vcf2multialign_amd import host
	cuts, min_distance, score = host.read_cut_positions(tmp_path / "cuts.bin")
	assert min_distance == 40 and cuts[0] == 0 and ("Maximum segmentation height: %d" % (1 + score)).encode() in r1.stdout
	(tmp_path / "bad.bin").write_bytes(b"garbage")
	assert run(common + ["-s", str(tmp_path / "c.a2m"), "-p", str(tmp_path / "bad.bin")]).returncode != 0


def test_options_outside_this_builds_scope_are_refused(tmp_path):
	"""--output-graphviz / --output-memory-breakdown (main.cc:53-120, :449-455) belong to the reference's command-line program, which SURVEY.md
	section 2 leaves out of scope: the driver says so instead of ignoring them."""
	fa, vcf = str(tmp_path / "g.fa"), str(tmp_path / "g.vcf")
	open(fa, "w").write(">1\nACGTTGCA\n")
	open(vcf, "w").write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n1\t3\ta\tG\tT\t.\tPASS\t.\tGT\t0|1\n")
	for option in ("--output-graphviz", "--output-memory-breakdown"):
		r = run(["-H", "-r", fa, "-a", vcf, "-c", "1", option + "=" + str(tmp_path / "x")])
		assert r.returncode != 0 and b"not supported by this build" in r.stderr
	r = run(["-H", "-r", fa, "-a", vcf, "-c", "1", "-v", str(tmp_path / "x")])   # --output-graphviz's short form (cmdline.ggo:38)
	assert r.returncode != 0 and b"not supported by this build" in r.stderr and b"Usage" not in r.stderr


def test_an_unusable_device_ends_the_run(tmp_path):
	"""--device=99: the context cannot be created.  The run ends with the GPU path's error and writes nothing.  (The driver looks at
	the context's fate right after the reference has been read -- by then it is known whenever the FASTA took longer than HIP's
	start-up, 0.2 s -- and again at its first use; with a 64-byte FASTA only the second check can be relied on.)"""
	fa, vcf = os.path.join(FIX, "test-2.fa"), os.path.join(FIX, "test-2.vcf")
	r = run(["-H", "-r", fa, "-a", vcf, "-c", "1", "-s", str(tmp_path / "x.a2m"), "--device=99"])
	assert r.returncode != 0 and b"ERROR (GPU path" in r.stderr
	assert not (tmp_path / "x.a2m").exists()


def test_unsupported_and_bad_arguments():
	assert run(["--founder-sequences=0", "-r", "x", "-a", "y", "-c", "1"]).returncode != 0
	assert run(["-H", "--founder-sequences=2", "-r", "x", "-a", "y", "-c", "1"]).returncode != 0
	r = run(["-H", "--output-memory-breakdown=m.txt", "-r", "x", "-a", "y", "-c", "1"])
	assert r.returncode != 0 and b"not supported" in r.stderr
	assert run(["-H", "-r", "x", "-a", "y"]).returncode != 0
	r = run(["-H", "-r", "/nonexistent.fa", "-a", "/nonexistent.vcf", "-c", "1"])
	assert r.returncode != 0 and b"Unable to read the reference" in r.stderr


def test_plain_c_example(tmp_path):
	"""examples/splice_rows.c: the ABI used from C99 with nothing but include/v2m_hip.h; it checks its own rows."""
	import shutil
	from vcf2multialign_amd import build
	src = os.path.join(ROOT, "examples", "splice_rows.c")
	exe = tmp_path / "splice_rows"
	subprocess.check_call([shutil.which("gcc"), "-std=c99", "-I" + os.path.join(ROOT, "include"), src, "-L" + build.PKG_DIR, "-lv2m_hip",
		"-Wl,-rpath," + build.PKG_DIR, "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)])
	r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
	assert r.returncode == 0, r.stderr.decode()
	assert r.stdout.decode().splitlines() == ["REF     ACG--TACGT", "copy 0  ACGTTTACGT", "copy 1  ACG--TACG-"]


def test_sharded_c_example(tmp_path):
	"""examples/sharded_rows.c: two contexts, each holding only its own slice of the path matrix (v2m_upload_path_slice), from C99."""
	import shutil
	from vcf2multialign_amd import build
	src = os.path.join(ROOT, "examples", "sharded_rows.c")
	exe = tmp_path / "sharded_rows"
	subprocess.check_call([shutil.which("gcc"), "-std=c99", "-I" + os.path.join(ROOT, "include"), src, "-L" + build.PKG_DIR, "-lv2m_hip",
		"-Wl,-rpath," + build.PKG_DIR, "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)])
	r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
	assert r.returncode == 0, r.stderr.decode()
	lines = r.stdout.decode().splitlines()
	assert len(lines) == 17 and lines[0] == "REF     ACG--TACGTAC" and lines[4] == "copy    ACGTTTACG-AC"
