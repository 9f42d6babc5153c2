#!/bin/bash
# AddressSanitizer + UBSan and ThreadSanitizer over the host C++ (CPU build only: the pool offers no GPU sanitizers): the
# multi-threaded VCF reader, the graph builder, the graph checkpoint, the founder search sequentially and on several threads,
# and -- over a CPU-only mock of the three ABI entry points it calls (tools/sanitize/output_harness.cc) -- the multi-context
# writers of output.cc (sharded pwritev, turnstile, per-context files, pipes; failing sinks, failing contexts, readers that
# exit) together with the worker pool of the bench's checksumming sink (csrc/synth/sink.cc).
# Everything is built and run under a scratch directory.
set -e -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
H="$ROOT/vcf2multialign_amd/csrc/host"
W="${TMPDIR:-/tmp}/v2m_sanitize.$$"
mkdir -p "$W"
trap 'rm -rf "$W"' EXIT
python3 - "$ROOT" "$W" <<'PY'
import sys
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, root + "/tests"); sys.path.insert(0, root)
import numpy as np, synth
rng = np.random.default_rng(5)
ref = synth.random_reference(rng, 300000)
synth.write_inputs(out, ref, synth.random_records(rng, ref, 6000, 30, multi_allelic=0.15, long_every=100), 30)
# What the reader does at scale: an input of several of its 8-MB chunks (14 000 records x 500 samples = 29 MB, 4 chunks), so that
# the worker window, the recycling of consumed chunks' vectors and the merge across chunk boundaries (overlaps reaching from one
# chunk into the next) all run under the sanitizers -- the small input above is ONE chunk.
ref = synth.random_reference(rng, 700000)
synth.write_inputs(out, ref, synth.random_records(rng, ref, 14000, 500, multi_allelic=0.15, long_every=100), 500, name="big")
# ... and the same input with a REF column that contradicts the reference in its SECOND chunk, for a delegate that stops there
# (variant_graph.cc:307-314): the build ends inside a chunk while workers are still parsing the chunks behind it.
import os
text = open(os.path.join(out, "big.vcf"), "rb").read()
assert len(text) > 4 * (8 << 20) - (4 << 20), len(text)
at = text.index(b"\n", 9 << 20) + 1                      # a data line well inside the second chunk
fields = text[at:text.index(b"\n", at)].split(b"\t")
fields[3] = (b"T" if fields[3][:1] != b"T" else b"G") + fields[3][1:]
open(os.path.join(out, "big_mismatch.vcf"), "wb").write(text[:at] + b"\t".join(fields) + text[text.index(b"\n", at):])
print("inputs: synth.vcf %d bytes (1 chunk), big.vcf %d bytes (%d chunks of 8 MB), big_mismatch.vcf: REF mismatch at byte %d" % (os.path.getsize(os.path.join(out, "synth.vcf")), len(text), -(-len(text) // (8 << 20)), at))
PY
for san in address,undefined thread; do
	echo "== -fsanitize=$san"
	g++ -O1 -g -std=c++20 -pthread -fsanitize=$san -fno-omit-frame-pointer -I"$H" -o "$W/reader" "$ROOT/tools/sanitize/reader_harness.cc" "$H/readers.cc" "$H/graph_builder.cc" "$H/graph_file.cc"
	"$W/reader" "$W/synth.fa" "$W/synth.vcf" "$W/graph.bin"
	"$W/reader" "$W/big.fa" "$W/big.vcf" "$W/graph.bin" chunks
	"$W/reader" "$W/big.fa" "$W/big_mismatch.vcf" "$W/graph.bin" stop
	g++ -O1 -g -std=c++20 -pthread -fsanitize=$san -fno-omit-frame-pointer -I"$H" -o "$W/founder" "$ROOT/tools/sanitize/founder_harness.cc" "$H/founder.cc"
	"$W/founder" 3000 600
	g++ -O1 -g -std=c++20 -pthread -fsanitize=$san -fno-omit-frame-pointer -I"$H" -o "$W/output" "$ROOT/tools/sanitize/output_harness.cc" "$H/output.cc" "$ROOT/vcf2multialign_amd/csrc/synth/sink.cc"
	mkdir -p "$W/out"
	for round in 1 2 3; do "$W/output" "$W/out"; done      # (thread interleavings differ from run to run)
done
echo "sanitizers: clean"
