#!/usr/bin/env python3
"""VCF text -> variant graph on the host only (reader + builder, no GPU work): wall clock of HostGraph(...) on a synthetic
config's VCF, per thread count.  `V2M_HOST_LIBRARY=path` loads another build of libv2m_host.so (A/B of reader changes).

    TMPDIR=/dev/shm python tools/parse_bench.py [config2|config3] [threads ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcf2multialign_amd import synth, build
if os.environ.get("V2M_HOST_LIBRARY"):
	build.HOST_LIB_PATH = os.environ["V2M_HOST_LIBRARY"]
from vcf2multialign_amd.host import HostGraph
cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
threads = [int(a) for a in sys.argv[2:]] or [1, 16]
tmp = os.environ.get("TMPDIR", "/tmp")
fa, vcf = os.path.join(tmp, cfg + ".fa"), os.path.join(tmp, cfg + ".vcf")
if not os.path.exists(vcf):
	t = time.time(); synth.dataset(cfg).write_fasta_and_vcf(fa, vcf)
	print("generated %s: VCF %.0f MB in %.1f s" % (cfg, os.path.getsize(vcf) / 1e6, time.time() - t), flush=True)
for n in threads:
	for rep in range(3):
		t = time.time()
		hg = HostGraph(fa, vcf, "1", threads=n)
		dt = time.time() - t
		print("%s threads %2d: %.3f s = %.0f MB/s (%d edges, matrix checksum %08x)" % (os.path.basename(build.HOST_LIB_PATH), n, dt, os.path.getsize(vcf) / dt / 1e6, len(hg.alt_edge_targets), int(hg.paths_by_edge_and_chrom_copy.sum() & 0xFFFFFFFF)), flush=True)
		del hg
