#!/usr/bin/env python3
"""--output-sequences-separate into a directory (tmpfs or disk) with one or several GPU contexts: BASELINE config 2 from a graph checkpoint, 2001 files of
10 Mbases.  Usage: python tools/e2e_separate_files.py DIR [devices, e.g. 0,0,0,0]"""
import os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import synth, build
from vcf2multialign_amd.host import HostGraph
outdir = os.path.join(sys.argv[1], "v2m_separate")
devices = sys.argv[2] if len(sys.argv) > 2 else None
cfg = "config2"
tmp = os.environ.get("TMPDIR", "/tmp")
fa, gf = os.path.join(tmp, cfg + ".fa"), os.path.join(tmp, cfg + ".v2mgraph")
ds = synth.dataset(cfg); g = ds.graph
ctx = v2m.Context(0); dev = torch.device("cuda", 0)
hp, ep = ds.path_cols, ds.path_rows
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ep // 64 * hp, dtype=torch.int64, device=dev); dst = torch.empty_like(src)
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.transpose_bits_device(src.data_ptr(), hp, ep, dst.data_ptr()); ctx.synchronize()
hg = HostGraph.from_arrays(g, src.cpu().numpy().view(np.uint64), hp, ep, ds.samples, ds.ploidy)
hg.set_transposed_paths(dst.cpu().numpy().view(np.uint64), ep, hp)
hg.write(gf)
open(fa, "wb").write(b">1\n" + ds.reference + b"\n")
ctx.close(); del src, dst
shutil.rmtree(outdir, ignore_errors=True); os.makedirs(outdir)
t = time.time()
p = subprocess.run([build.CLI_PATH, "-H", "-r", fa, "-g", gf, "-c", "1", "--output-sequences-separate"] + (["--device=" + devices] if devices else []), cwd=outdir, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
total = time.time() - t
files = os.listdir(outdir)
size = sum(os.path.getsize(os.path.join(outdir, f)) for f in files)
marks = [l for l in p.stderr.splitlines() if "Outputting" in l or l.strip() == "Done."]
print("devices %s -> %s: exit %d, %d files, %.2f GB in %.2f s of process time = %.1f GB/s" % (devices or "0", outdir, p.returncode, len(files), size / 1e9, total, size / total / 1e9), flush=True)
shutil.rmtree(outdir); os.remove(gf); os.remove(fa)
sys.exit(p.returncode)
