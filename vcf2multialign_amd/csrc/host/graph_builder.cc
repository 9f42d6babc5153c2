#include "graph_builder.hh"

#include <algorithm>

namespace v2m::host {

alt_kind classify_alt(std::string_view alt)
{
	if (alt == "<DEL>") return alt_kind::deletion;
	if (alt.empty()) return alt_kind::unhandled;
	for (char const c : alt) {
		switch (c) {
			case 'A': case 'C': case 'G': case 'T': case 'N':
			case 'a': case 'c': case 'g': case 't': case 'n':
				continue;
			default:
				return alt_kind::unhandled;   // ".", "*", "<CNV...>", breakends: no edge (variant_graph.cc:362-363)
		}
	}
	return alt_kind::sequence;
}


graph_builder::graph_builder(variant_graph &graph, bool track_paths, u64 path_alignment) : m_graph(graph), m_track_paths(track_paths), m_path_alignment(path_alignment < 64 ? 64 : (path_alignment + 63) / 64 * 64)
{
	m_graph.alt_edge_count_csum.assign(1, 0);   // variant_graph.cc:147
	add_node(0, 0);                             // variant_graph.cc:148
}


void graph_builder::begin(std::vector<std::string> sample_names, std::vector<u32> const &ploidies)
{
	m_graph.sample_names = std::move(sample_names);
	m_graph.ploidy_csum.assign(1, 0);
	for (u32 const p : ploidies) m_graph.ploidy_csum.push_back(m_graph.ploidy_csum.back() + p);
	u64 const copies(m_graph.ploidy_csum.back());
	if (m_track_paths) {
		u64 const rows(m_path_alignment * ((copies + m_path_alignment - 1) / m_path_alignment));   // variant_graph.cc:277
		m_graph.paths_by_edge_and_chrom_copy = bit_matrix(rows, rows ? 512 : 0);
		m_target_ref_pos_by_copy.assign(copies, 0);
	}
}


u64 graph_builder::add_node(u64 ref_pos, u64 aln_pos)
{
	m_graph.reference_positions.push_back(ref_pos);
	m_graph.aligned_positions.push_back(aln_pos);
	m_graph.alt_edge_count_csum.push_back(m_graph.alt_edge_count_csum.back());
	return m_graph.reference_positions.size() - 1;
}


u64 graph_builder::add_or_update_node(u64 ref_pos, u64 aln_pos)
{
	if (m_graph.reference_positions.back() < ref_pos)
		return add_node(ref_pos, aln_pos);
	m_graph.aligned_positions.back() = std::max(m_graph.aligned_positions.back(), aln_pos);   // variant_graph.cc:94
	return m_graph.reference_positions.size() - 1;
}


u64 graph_builder::add_edge(std::string_view label)
{
	++m_graph.alt_edge_count_csum.back();   // edges always attach to the latest node (variant_graph.cc:99-105)
	m_graph.alt_edge_targets.push_back(0);
	m_graph.alt_edge_label_bytes.append(label);
	m_graph.alt_edge_label_offsets.push_back(m_graph.alt_edge_label_bytes.size());
	return m_graph.alt_edge_targets.size() - 1;
}


// variant_graph.cc:160-179: create the target nodes of every pending ALT that ends at or before ref_pos.
void graph_builder::flush_targets(u64 ref_pos)
{
	while (!m_pending.empty() && m_pending.top().ref_pos <= ref_pos) {
		auto const t(m_pending.top());
		m_pending.pop();
		m_aln_pos = std::max(m_aln_pos + (t.ref_pos - m_prev_ref_pos), t.aln_pos);
		m_graph.alt_edge_targets[t.edge] = add_or_update_node(t.ref_pos, m_aln_pos);
		m_prev_ref_pos = t.ref_pos;
	}
}


bool graph_builder::add_record(u64 ref_pos, u64 ref_allele_length, alt_allele const *alts, std::size_t n_alts)
{
	if (ref_pos < m_prev_ref_pos) return false;                       // variant_graph.cc:293-297
	flush_targets(ref_pos);                                           // :300
	m_aln_pos += ref_pos - m_prev_ref_pos;                            // :303-304
	add_or_update_node(ref_pos, m_aln_pos);                           // :305

	m_edges_by_alt.assign(n_alts, kEdgeMax);
	m_current_edge_targets.clear();
	bool first(true);
	u64 max_edge(0);
	u64 const target_pos(ref_pos + ref_allele_length);                // :333
	for (std::size_t a(0); a < n_alts; ++a) {
		u64 edge;
		switch (alts[a].kind) {
			case alt_kind::sequence:
				edge = add_edge(alts[a].sequence);
				m_pending.push({target_pos, m_seq++, edge, m_aln_pos + alts[a].sequence.size()});   // :338
				break;
			case alt_kind::deletion:
				edge = add_edge({});
				m_pending.push({target_pos, m_seq++, edge, m_aln_pos});                             // :344
				break;
			default:
				continue;
		}
		m_edges_by_alt[a] = edge;
		m_current_edge_targets.push_back(target_pos);
		if (first) { m_min_edge = edge; first = false; }
		max_edge = edge;
	}

	if (m_track_paths) {                                              // :368-376, growth policy is ours (amortised doubling)
		auto &m(m_graph.paths_by_edge_and_chrom_copy);
		if (m.rows && m.cols <= max_edge)
			m.set_column_count(std::max<u64>(2 * m.cols, max_edge + 1));
	}
	m_cur_ref_pos = ref_pos;
	m_prev_ref_pos = ref_pos;                                         // :427
	return true;
}


void graph_builder::add_record_node_only(u64 ref_pos)
{
	if (ref_pos < m_prev_ref_pos) return;
	flush_targets(ref_pos);                                           // :300
	m_aln_pos += ref_pos - m_prev_ref_pos;                            // :303-304
	add_or_update_node(ref_pos, m_aln_pos);                           // :305; :427 (prev_ref_pos = ref_pos) is not reached
	m_edges_by_alt.clear();
}


void graph_builder::set_genotype(u32 copy_row, u32 alt_number)
{
	if (0 == alt_number || m_edges_by_alt.size() < alt_number) return;
	u64 const edge(m_edges_by_alt[alt_number - 1]);
	if (kEdgeMax == edge) return;                                     // :401-403
	if (m_cur_ref_pos < m_target_ref_pos_by_copy[copy_row] && m_overlap_cb)   // :408-418
		m_overlap_cb({m_cur_ref_pos, copy_row, alt_number});
	m_target_ref_pos_by_copy[copy_row] = m_current_edge_targets[edge - m_min_edge];   // :422-423
	m_graph.paths_by_edge_and_chrom_copy.set(copy_row, edge);         // :424
}


void graph_builder::finish(u64 ref_length)
{
	flush_targets(ref_length);                                        // :437-443
	add_or_update_node(ref_length, m_aln_pos + (ref_length - m_prev_ref_pos));
	if (m_track_paths)
		m_graph.paths_by_edge_and_chrom_copy.set_column_count(m_path_alignment * ((m_graph.edge_count() + m_path_alignment - 1) / m_path_alignment));   // :445-451
	// variant_graph.cc:453 (paths_by_chrom_copy_and_edge = transpose_matrix(...)) is the caller's next
	// step and runs on the GPU: see transpose_paths() in gpu_path.cc.  There is no CPU transpose here.
}

} // namespace v2m::host
