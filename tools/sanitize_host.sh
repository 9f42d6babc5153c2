#!/bin/bash
# AddressSanitizer + UBSan and ThreadSanitizer over the host C++ that has no GPU dependency (CPU build only: the pool
# offers no GPU sanitizers): the multi-threaded VCF reader, the graph builder, the graph checkpoint, and the founder
# search sequentially and on several threads.  Everything is built and run under a scratch directory.
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
PY
for san in address,undefined thread; do
	echo "== -fsanitize=$san"
	g++ -O1 -g -std=c++20 -pthread -fsanitize=$san -fno-omit-frame-pointer -I"$H" -o "$W/reader" "$ROOT/tools/sanitize/reader_harness.cc" "$H/readers.cc" "$H/graph_builder.cc" "$H/graph_file.cc"
	"$W/reader" "$W/synth.fa" "$W/synth.vcf" "$W/graph.bin"
	g++ -O1 -g -std=c++20 -pthread -fsanitize=$san -fno-omit-frame-pointer -I"$H" -o "$W/founder" "$ROOT/tools/sanitize/founder_harness.cc" "$H/founder.cc"
	"$W/founder" 3000 600
done
echo "sanitizers: clean"
