#!/usr/bin/env python3
"""Write rate of several simultaneously held 63-GB device buffers (v2m_alloc_output's probe) -- run it under
rocprofv3 --pmc <counters> to see what differs between a fast and a slow buffer; tools/region_probe_summary.py
prints the counters per candidate next to the measured rates."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import vcf2multialign_amd as v2m
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
gb = float(sys.argv[2]) if len(sys.argv) > 2 else 63.0
ctx = v2m.Context(0)
p = ctx.alloc_output(int(gb * 1e9), candidates=n)
print("RATES " + ctx.info, flush=True)
ctx.free_output(p)
ctx.close()
