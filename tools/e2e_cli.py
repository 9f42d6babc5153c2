#!/usr/bin/env python3
"""A BASELINE config through the whole text pipeline: synthetic FASTA + VCF (config 2: 400 MB, config 3: 10 GB; put TMPDIR
on /dev/shm for that one) -> bin/vcf2multialign --haplotypes -> A2M (20 GB / 502 GB) into /dev/null, wall-clock per stage
from the driver's own log lines.  Usage: python tools/e2e_cli.py [config2|config3] [output path, default /dev/null; a real file (e.g. on
/dev/shm) shows what the writer costs beyond the link: the file is checked for its size and removed]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcf2multialign_amd import synth, build
cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
dest = sys.argv[2] if len(sys.argv) > 2 else "/dev/null"
tmp = os.environ.get("TMPDIR", "/tmp")
fa, vcf = os.path.join(tmp, cfg + ".fa"), os.path.join(tmp, cfg + ".vcf")
t = time.time(); ds = synth.dataset(cfg); ds.write_fasta_and_vcf(fa, vcf)
print("generated %s: VCF %.0f MB in %.1f s" % (cfg, os.path.getsize(vcf) / 1e6, time.time() - t), flush=True)
t = time.time()
# (overlapping variants go to a file: on stdout, which is read only after the run, they would fill the pipe and stall the driver)
overlaps = os.path.join(tmp, cfg + ".overlaps.tsv")
p = subprocess.Popen([build.CLI_PATH, "-H", "-r", fa, "-a", vcf, "-c", "1", "-s", dest, "--output-graph-statistics", "--output-overlaps=" + overlaps], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
marks = []
for line in p.stderr:
	marks.append((time.time() - t, line.rstrip()))
p.wait()
total = time.time() - t
for m in marks:
	print("  %7.2f s  %s" % m)
print(p.stdout.read().strip())
rows, L = ds.n_copies + 1, ds.graph.aligned_length
print("exit %d; total %.2f s for %d rows x %d bases = %.1f Gbases -> %.2f Gbases/s end to end incl. VCF parsing" % (p.returncode, total, rows, L, rows * L / 1e9, rows * L / total / 1e9))
print("overlap report: %.1f MB" % (os.path.getsize(overlaps) / 1e6))
if dest != "/dev/null":
	size = os.path.getsize(dest)
	# '>' id '\n' body '\n' per row: REF + H rows (haplotype_output.cc:48-81)
	print("output file %s: %.2f GB (%d rows x (%d bases + header + newline)) written in the run above" % (dest, size / 1e9, rows, L))
	assert size >= rows * (L + 1)
	os.remove(dest)
os.remove(vcf); os.remove(fa); os.remove(overlaps)
