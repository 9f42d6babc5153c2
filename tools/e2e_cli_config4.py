#!/usr/bin/env python3
"""BASELINE config 4 through the command-line driver: the config-3 graph (written once as this build's graph checkpoint,
both path matrices included) + FASTA -> bin/vcf2multialign --founder-sequences=25 --minimum-distance=50 -> A2M into
/dev/null, wall clock per stage from the driver's own log lines.  (Config 3's VCF text would be ~10 GB; the text
pipeline is exercised at config 2, tools/e2e_cli.py.)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import synth, build
from vcf2multialign_amd.host import HostGraph

cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
mode = sys.argv[2] if len(sys.argv) > 2 else "founders"    # or "haplotypes": BASELINE config 3 (all 5009 rows, 501 GB of A2M); "unaligned": the same rows with --unaligned
devices = sys.argv[3] if len(sys.argv) > 3 else None        # e.g. "0,0": several contexts, each with its own slice of the path matrix
tmp = os.environ.get("TMPDIR", "/tmp")
fa, gf = os.path.join(tmp, cfg + ".fa"), os.path.join(tmp, cfg + ".v2mgraph")
t = time.time()
ds = synth.dataset(cfg); g = ds.graph
ctx = v2m.Context(0)
dev = torch.device("cuda", 0)
hp, ep = ds.path_cols, ds.path_rows
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ep // 64 * hp, dtype=torch.int64, device=dev); dst = torch.empty_like(src)
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.transpose_bits_device(src.data_ptr(), hp, ep, dst.data_ptr())
ctx.synchronize()
hg = HostGraph.from_arrays(g, src.cpu().numpy().view(np.uint64), hp, ep, ds.samples, ds.ploidy)
hg.set_transposed_paths(dst.cpu().numpy().view(np.uint64), ep, hp)
hg.write(gf)
with open(fa, "wb") as f:
	f.write(b">1\n")
	ref = ds.reference
	for i in range(0, len(ref), 1 << 20):
		f.write(ref[i:i + (1 << 20)] + b"\n")
ctx.close(); del src, dst
print("prepared %s: graph file %.0f MB, FASTA %.0f MB in %.1f s" % (cfg, os.path.getsize(gf) / 1e6, os.path.getsize(fa) / 1e6, time.time() - t), flush=True)

t = time.time()
wrap = os.environ.get("E2E_WRAP", "").split()   # e.g. "rocprofv3 --hip-trace --stats -d DIR -o t --": the driver under a profiler
p = subprocess.Popen(wrap + [build.CLI_PATH] + (["-F", "25", "-d", "50"] if mode == "founders" else ["-H", "--unaligned"] if mode == "unaligned" else ["-H"]) + ["-r", fa, "-g", gf, "-c", "1", "-s", os.environ.get("E2E_DEST", "/dev/null"), "--output-graph-statistics"] + (["--device=" + devices, "--verbose"] if devices else []),
	stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
marks = []
for line in p.stderr:
	marks.append((time.time() - t, line.rstrip()))
p.wait()
total = time.time() - t
for m in marks:
	if not m[1].startswith("Handled "):
		print("  %7.2f s  %s" % m)
print(p.stdout.read().strip())
rows = 26 if mode == "founders" else ds.n_copies + 1
print("exit %d; total %.2f s (%d rows x %d bases = %.1f Gbases -> %.1f Gbases/s end to end)" % (p.returncode, total, rows, g.aligned_length, rows * g.aligned_length / 1e9, rows * g.aligned_length / 1e9 / total))
os.remove(gf); os.remove(fa)
if os.environ.get("E2E_DEST", "/dev/null") != "/dev/null":
	print("output file: %.2f GB" % (os.path.getsize(os.environ["E2E_DEST"]) / 1e9)); os.remove(os.environ["E2E_DEST"])
sys.exit(p.returncode)
