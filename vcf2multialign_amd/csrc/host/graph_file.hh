// graph_file.hh -- flat binary file of a variant graph (--output-graph / --input-graph).
//
// The reference checkpoints its graph with cereal's PortableBinary archive (vcf2multialign/main.cc:393-426, field
// order include/vcf2multialign/variant_graph.hh:197-209); cereal and libbio's serializers are absent here, so that
// byte format cannot be reproduced.  This is our own format with the same content and field order, laid out so that
// every array can be handed to v2m_upload_graph() straight from a read (or an mmap):
//
//   char     magic[8] = "V2MGRAF1"
//   u64      counts[8] = {nodes, edges, label bytes, samples, sample-name bytes, ploidy_csum entries,
//                         paths_by_chrom_copy_and_edge rows, cols}        (+ u64 rows, cols of paths_by_edge_and_chrom_copy)
//   u64      reference_positions[nodes], aligned_positions[nodes], alt_edge_targets[edges],
//            alt_edge_count_csum[nodes + 1], alt_edge_label_offsets[edges + 1]
//   u64      paths_by_chrom_copy_and_edge words, paths_by_edge_and_chrom_copy words
//   u32      ploidy_csum[], padded to 8 bytes
//   char     label bytes, padded to 8;  sample names, NUL-terminated each, padded to 8
//   u64      checksum of everything before it (sum of mix64(word index ^ word))
// Little-endian, as written by the machine (x86-64 / gfx950 hosts only).
#pragma once

#include <string>

#include "variant_graph.hh"

namespace v2m::host {

void write_graph(variant_graph const &graph, char const *path);   // throws std::runtime_error
void read_graph(char const *path, variant_graph &graph);          // throws std::runtime_error on I/O errors, bad magic, bad checksum

} // namespace v2m::host
