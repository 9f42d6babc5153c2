#include "founder.hh"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <exception>
#include <thread>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <numeric>
#include <stdexcept>

namespace v2m::host {

namespace {

// Divergence values are kept BIASED by one in 32 bits: b = d + 1, so that "no match yet" (the reference's
// DIVERGENCE_MAX) is 0 and the order the reference defines -- DIVERGENCE_MAX first (pbwt.hh:25-42) -- is the plain
// integer order.  Edge indices stay below 2^32 - 3 (checked by the callers); the device side narrows them the same way.
constexpr u32 kNoValue = UINT32_MAX;

inline u64 unbiased(u32 b) { return b ? u64(b) - 1 : UINT64_MAX; }

// How often each divergence value occurs (the reference's std::map, pbwt.hh:46): a flat table over the biased values
// plus a bitset hierarchy of the non-zero entries, so that a change costs a few word operations and the values can be
// listed from the largest down.
class divergence_counts {
public:
	explicit divergence_counts(u64 n_values) : m_count(n_values, 0)
	{
		for (u64 n(n_values);;) {
			n = (n + 63) / 64;
			m_level.emplace_back(n, 0);
			if (n <= 1) break;
		}
	}

	u32 operator[](u32 v) const { return m_count[v]; }

	void add(u32 v, u32 n)
	{
		if (0 == m_count[v]) mark(v);
		m_count[v] += n;
	}

	void move(u32 from, u32 to)                             // one copy's value changes
	{
		if (0 == --m_count[from]) unmark(from);
		if (0 == m_count[to]++) mark(to);
	}

	// largest value present below `bound` (exclusive), kNoValue if there is none
	u32 below(u64 bound) const
	{
		u64 pos(bound);
		std::size_t k(0);
		for (;;) {
			u64 const w(pos >> 6), b(pos & 63);
			u64 const m((b && w < m_level[k].size()) ? m_level[k][w] & ((u64(1) << b) - 1) : 0);
			if (m) { pos = (w << 6) + 63 - u64(__builtin_clzll(m)); break; }
			pos = w;                                            // nothing in this word: look for an earlier word one level up
			if (++k == m_level.size()) return kNoValue;
		}
		while (k) { --k; pos = (pos << 6) + 63 - u64(__builtin_clzll(m_level[k][pos])); }
		return u32(pos);
	}

	u32 largest() const { return below(m_count.size()); }

private:
	void mark(u64 v)
	{
		for (auto &level : m_level) {
			u64 &w(level[v >> 6]);
			bool const was_empty(0 == w);
			w |= u64(1) << (v & 63);
			if (!was_empty) break;
			v >>= 6;
		}
	}

	void unmark(u64 v)
	{
		for (auto &level : m_level) {
			u64 &w(level[v >> 6]);
			w &= ~(u64(1) << (v & 63));
			if (w) break;
			v >>= 6;
		}
	}

	std::vector<u32> m_count;
	std::vector<std::vector<u64>> m_level;                      // [0]: bit per value; [k + 1]: bit per word of [k]
};

// byte i of kSpread[x] = bit i of x
struct spread_table {
	u64 v[256];
	spread_table() { for (u32 x(0); x < 256; ++x) { v[x] = 0; for (u32 i(0); i < 8; ++i) v[x] |= u64((x >> i) & 1) << (8 * i); } }
};
spread_table const kSpread;

// Positional BWT over the ALT edges, one binary column per edge (pbwt.hh:21-145): the copies sorted by their
// reversed edge-usage prefixes and the divergence array.  `counts`, when given, follows the divergence values
// (find_cut_positions needs them, find_matchings does not).
class edge_pbwt {
public:
	std::vector<u32> order;        // "permutation": copies in prefix order
	std::vector<u32> divergence;   // biased

	edge_pbwt(u32 copies, divergence_counts *counts) : order(copies), divergence(copies, 0), m_counts(counts)
	{
		std::iota(order.begin(), order.end(), 0u);
		if (copies) {                                       // pbwt.hh:62-70
			divergence[0] = 1;
			if (m_counts) {
				m_counts->add(1, 1);
				if (copies > 1) m_counts->add(0, copies - 1);
			}
		}
		m_prev_order.resize(copies);
		m_prev_divergence.resize(copies);
	}

	// From now on keep `counts` in step with the divergence values (it must already describe them).
	void follow(divergence_counts *counts) { m_counts = counts; }

	// One step of Durbin's algorithm 2 for edge `edge` whose usage bits are `column` (pbwt.hh:77-134).
	void advance(u64 const *column, u64 column_words, u64 edge)
	{
		m_prev_order.swap(order);                           // swap_vectors(), pbwt.hh:137-145
		m_prev_divergence.swap(divergence);
		u32 const n(u32(m_prev_order.size()));

		// usage bits as one byte per copy: the loop below looks them up in prefix order
		m_flag.resize(column_words * 64);
		u32 ones(0);
		for (u64 w(0); w < column_words; ++w) {
			u64 const x(column[w]);
			ones += u32(__builtin_popcountll(x));
			u64 *const dst(reinterpret_cast<u64 *>(m_flag.data()) + 8 * w);
			for (u32 k(0); k < 8; ++k) dst[k] = kSpread.v[(x >> (8 * k)) & 255];
		}

		if (m_counts) step<true>(n, ones, edge);
		else step<false>(n, ones, edge);
	}

private:
	// The copies that follow the edge are few for most edges, so the two cases are ordinary (well predicted) branches.
	template <bool kCounts>
	void step(u32 const n, u32 const ones, u64 const edge)
	{
		u32 const *const __restrict prev_order(m_prev_order.data());
		u32 const *const __restrict prev_div(m_prev_divergence.data());
		unsigned char const *const __restrict flag(m_flag.data());
		u32 *const __restrict out_order(order.data());
		u32 *const __restrict out_div(divergence.data());
		u32 zero_at(0), one_at(n - ones);
		u32 p(u32(edge) + 2), q(p);                         // biased edge + 1
		for (u32 i(0); i < n; ++i) {
			u32 const copy(prev_order[i]);
			u32 const d(prev_div[i]);
			p = std::max(p, d);
			q = std::max(q, d);
			// The reference decrements the old value's count and increments the new one's for every copy
			// (pbwt.hh:110-131); when the value does not change -- the common case, a copy whose predecessor in
			// the order used the same allele -- the two cancel, so the table is only touched on a change.
			if (!flag[copy]) {
				if (kCounts && p != d) m_counts->move(d, p);
				out_order[zero_at] = copy;
				out_div[zero_at] = p;
				++zero_at;
				p = 1;                                          // biased 0
			} else {
				if (kCounts && q != d) m_counts->move(d, q);
				out_order[one_at] = copy;
				out_div[one_at] = q;
				++one_at;
				q = 1;
			}
		}
	}

	divergence_counts *m_counts;
	std::vector<u32> m_prev_order, m_prev_divergence;
	std::vector<unsigned char> m_flag;
};

void check_edge_range(variant_graph const &graph)
{
	if (graph.edge_count() >= u64(UINT32_MAX) - 3)
		throw std::length_error("founder search: edge indices are kept in 32 bits");
}

struct cut_candidate {
	u64 edge;        // first edge of the node
	u64 prev_edge;   // edge of the best predecessor, kEdgeMax = none
	u64 node;
	u32 score;

	void improve(u32 class_count, cut_candidate const &pred)                 // find_cut_positions.cc:55-63
	{
		u32 const candidate(std::max(class_count, pred.score));
		if (candidate < score) { score = candidate; prev_edge = pred.edge; }
	}
};

inline u64 const *edge_column(variant_graph const &g, u64 edge)
{
	auto const &m(g.paths_by_edge_and_chrom_copy);
	return m.words.data() + edge * m.words_per_column();
}

} // namespace


namespace {

// Walks the predecessor links back from the last candidate (find_cut_positions.cc:182-209).
u32 collect_cut_positions(std::vector<cut_candidate> const &cuts, variant_graph const &graph, std::vector<u64> &out)
{
	if (cuts.size() <= 1) return kCutPositionScoreMax;                        // :182-183
	auto const by_edge([](cut_candidate const &c, u64 e) { return c.edge < e; });
	auto it(cuts.cend() - 1);
	u32 const score(it->score);
	for (;;) {
		out.push_back(it->node);
		if (kEdgeMax == it->prev_edge) break;
		it = std::lower_bound(cuts.cbegin(), it, it->prev_edge, by_edge);
	}
	if (0 != out.back()) out.push_back(0);
	std::reverse(out.begin(), out.end());
	if (out.back() != graph.node_count() - 1) out.back() = graph.node_count() - 1;   // the sink usually has no ALT in-edges
	return score;
}

u32 find_cut_positions_sequential(variant_graph const &graph, u64 min_distance, std::vector<u64> &out)
{
	out.clear();
	u32 const copies(graph.total_chromosome_copies());
	u64 const words_per_column(graph.paths_by_edge_and_chrom_copy.words_per_column());
	check_edge_range(graph);
	divergence_counts counts(graph.edge_count() + 2);                          // biased values 0 .. edge_count + 1
	edge_pbwt pbwt(copies, &counts);

	std::vector<cut_candidate> cuts;
	cuts.push_back({0, kEdgeMax, 0, 0});                                      // :111-112
	auto const by_edge([](cut_candidate const &c, u64 e) { return c.edge < e; });

	u64 rightmost_target(0), edge(0), last_cut_edge(kEdgeMax);
	for (u64 node(0); node < graph.node_count(); ++node) {
		// a node no ALT edge jumps over ends a bridge: candidate cut (:126); one candidate per distinct edge index (:129)
		if (rightmost_target <= node && last_cut_edge != edge) {
			cuts.push_back({edge, kEdgeMax, node, copies});
			last_cut_edge = edge;
			cut_candidate &current(cuts.back());

			// Divergence values from the largest down: the number of path classes of the segment that starts at the
			// corresponding edge grows as we go left, so every earlier candidate needs to be looked at once (:134-165).
			auto right_bound(cuts.end());
			u32 value(counts.largest());
			u32 class_count(kNoValue == value ? 0 : counts[value]);
			if (kNoValue != value) {
				for (value = counts.below(value); kNoValue != value; value = counts.below(value)) {   // all values but the largest, descending
					auto const pred(std::lower_bound(cuts.begin(), right_bound, unbiased(value), by_edge));
					if (pred != right_bound) {
						right_bound = pred;
						if (min_distance <= graph.aligned_positions[node] - graph.aligned_positions[pred->node])
							current.improve(class_count, *pred);
					}
					class_count += counts[value];
				}
			}
			if (cuts.begin() != right_bound) {                                // the segment may reach further left still
				--right_bound;
				current.improve(class_count, *right_bound);
			}
		}

		for (u64 e(graph.alt_edge_count_csum[node]); e < graph.alt_edge_count_csum[node + 1]; ++e) {   // :170-176
			pbwt.advance(edge_column(graph, edge), words_per_column, edge);
			++edge;
			rightmost_target = std::max(rightmost_target, graph.alt_edge_targets[e]);
		}
	}

	return collect_cut_positions(cuts, graph, out);
}

} // namespace


namespace {

struct joined_class {                                                         // founder_sequence_greedy_output.cc:48-69
	u32 lhs_rep, rhs_rep, size;
	bool operator<(joined_class const &o) const { return size < o.size; }
};

// The reference keeps "class representative -> founder" in a std::multimap (founder_sequence_greedy_output.cc:174).  It never
// holds more entries than there are founders, every founder is in it at most once, it is refilled in founder order at every cut
// (:431-436, and :248-302 for the first one), and it is only ever asked two things: find(key) -- which for a multimap is the
// first inserted of the entries with that key, i.e. the LOWEST founder on that class -- and begin(), the smallest key (again its
// lowest founder); what is erased is always the entry just found.  So the map is kept as what it is a view of: one key per
// founder, plus, per class, the queue of the founders that sit on it in founder order.  find() is the queue's head, erase() pops
// it, refilling is a pass over the founders -- where two dozen sorted insertions per cut, then scans of the keys, were most of
// the assignment's time at several hundred thousand cuts (a cut makes ~40 lookups: the reference's loop :353-403 goes over the
// joined classes again and again while founders are left, moving one founder per class and pass).
class class_to_founder_map {
public:
	static constexpr u64 kGone = UINT64_MAX;       // (keys are 32-bit class representatives, kPloidyMax included)
	static constexpr u32 kNone = UINT32_MAX;

	void reset(u32 founders, u32 classes)
	{
		m_key.assign(founders, kGone);
		m_next.assign(founders, kNone);
		m_head.assign(classes, kNone);
		m_tail.assign(classes, kNone);
		m_live = 0;
	}
	bool empty() const { return 0 == m_live; }
	void clear()
	{
		for (u64 &k : m_key) { if (k < m_head.size()) m_head[k] = m_tail[k] = kNone; k = kGone; }
		m_live = 0;
	}
	void emplace(u32 key, u32 founder)             // a founder that is not in the map; founders arrive in ascending order
	{
		m_key[founder] = key;
		m_next[founder] = kNone;
		++m_live;
		if (key < m_head.size()) {
			if (kNone == m_head[key]) m_head[key] = founder; else m_next[m_tail[key]] = founder;
			m_tail[key] = founder;
		}
	}
	void erase(u32 founder)                        // the founder find() or first() has just returned: the head of its class's queue
	{
		u64 const key(m_key[founder]);
		if (key < m_head.size()) {
			if (m_head[key] != founder) throw std::logic_error("founder matching: a founder left its class out of turn");
			m_head[key] = m_next[founder];
			if (kNone == m_head[key]) m_tail[key] = kNone;
		}
		m_key[founder] = kGone;
		--m_live;
	}

	u32 find(u32 key) const                        // the lowest founder on class `key`, or kNone
	{
		if (key < m_head.size()) return m_head[key];
		u32 const n(u32(m_key.size()));            // (kPloidyMax: founders without a class)
		for (u32 f(0); f < n; ++f) if (m_key[f] == key) return f;
		return kNone;
	}

	u32 first() const                              // the founder of begin(): smallest key, lowest founder; the map must not be empty
	{
		u32 const n(u32(m_key.size()));
		u64 best(kGone);
		u32 best_f(kNone);
		for (u32 f(0); f < n; ++f) if (m_key[f] < best) { best = m_key[f]; best_f = f; }
		return best_f;
	}

private:
	std::vector<u64> m_key;                        // per founder: the class it sits on, kGone once it has been continued / before it has one
	std::vector<u32> m_next;                       // per founder: the next founder on the same class
	std::vector<u32> m_head, m_tail;               // per class: its queue
	u32 m_live{};
};

// The matching state that survives from one cut to the next.
struct matcher {
	u32 founders;
	std::size_t rows;
	std::vector<u32> &assigned;                    // rows x founders, column-major
	class_to_founder_map founder_by_class;         // class representative of the previous block -> founder
	std::vector<char> reserved;
	std::vector<u32> loose_rhs;
	std::vector<u32> reserved_set;                 // the entries of `reserved` that are 1
	void reserve(u32 rep) { reserved[rep] = 1; reserved_set.push_back(rep); }

	// The assignment is produced row by row and every row touches all founders, so it is kept row-major here (one cache line or
	// two per cut instead of one per founder, each rows * 4 bytes apart) and turned into the reference's column-major matrix
	// (founder_sequence_greedy_output.cc:171) at the end: finish().
	std::vector<u32> by_row = std::vector<u32>(rows * founders, kPloidyMax);
	u32 &slot(std::size_t row, u32 founder) { return by_row[row * founders + founder]; }
	void finish()
	{
		for (std::size_t row(0); row < rows; ++row)
			for (u32 f(0); f < founders; ++f) assigned[f * rows + row] = by_row[row * founders + f];
		std::vector<u32>().swap(by_row);
	}

	// second cut: seed row 0 (:248-302)
	void seed(std::vector<joined_class> const &joined, u32 lhs_distinct)
	{
		founder_by_class.reset(founders, u32(reserved.size()));
		u32 free_founders(founders);
		u32 reserved_left(std::min(free_founders, lhs_distinct));
		free_founders -= reserved_left;
		u32 next_founder(0);
		auto const give([&](joined_class const &c) {
			founder_by_class.emplace(c.lhs_rep, next_founder);
			slot(0, next_founder) = c.lhs_rep;
			++next_founder;
		});
		for (auto c(joined.rbegin()); c != joined.rend(); ++c) {              // largest joined class first
			if (reserved[c->lhs_rep]) {
				if (free_founders) { --free_founders; give(*c); }
			} else if (reserved_left) {
				--reserved_left;
				reserve(c->lhs_rep);
				give(*c);
			}
		}
		while (free_founders && !joined.empty())                              // every founder gets a class (:291-302)
			for (auto c(joined.rbegin()); c != joined.rend() && free_founders; ++c) { --free_founders; give(*c); }
	}

	// every cut from the second on: continue the founders into the block on the right (:304-442)
	void extend(std::size_t row, std::vector<joined_class> const &joined, u32 rhs_distinct)
	{
		// (only what the previous cut set: clearing all of `reserved` -- one byte per chromosome copy -- at each of several hundred
		// thousand cuts was most of the assignment's time)
		for (u32 const rep : reserved_set) reserved[rep] = 0;
		reserved_set.clear();
		loose_rhs.clear();
		u32 free_founders(founders);
		u32 reserved_left(std::min(free_founders, rhs_distinct));
		free_founders -= reserved_left;

		auto const follow([&](joined_class const &c) -> bool {               // a founder currently on c.lhs_rep moves to c.rhs_rep
			u32 const founder(founder_by_class.find(c.lhs_rep));
			if (class_to_founder_map::kNone == founder) return false;
			founder_by_class.erase(founder);
			slot(row, founder) = c.rhs_rep;
			return true;
		});
		auto const place_anywhere([&](u32 rhs_rep) {
			if (founder_by_class.empty()) return;
			u32 const founder(founder_by_class.first());
			founder_by_class.erase(founder);
			slot(row, founder) = rhs_rep;
		});

		// steps 1-3 (:353-403)
		bool first_pass(true), stop(false);
		while (!stop) {
			bool moved(false);
			for (auto c(joined.rbegin()); c != joined.rend(); ++c) {
				if (reserved[c->rhs_rep]) {
					if (free_founders) {
						if (follow(*c)) { moved = true; --free_founders; }
					} else if (!first_pass) {
						stop = true;
						break;
					}
				} else if (reserved_left) {
					--reserved_left;
					if (follow(*c)) reserve(c->rhs_rep);
					else loose_rhs.push_back(c->rhs_rep);
				}
			}
			if (stop || !free_founders) break;
			if (first_pass) { first_pass = false; continue; }
			if (!moved) break;
		}

		// step 4 (:405-416): right-hand classes that nobody could follow into
		for (u32 const rhs_rep : loose_rhs) {
			if (!reserved[rhs_rep]) {
				place_anywhere(rhs_rep);
				reserve(rhs_rep);
			}
		}
		// step 5 (:418-428): founders still without a continuation
		while (!founder_by_class.empty() && !joined.empty())
			for (auto c(joined.rbegin()); c != joined.rend() && !founder_by_class.empty(); ++c) place_anywhere(c->rhs_rep);

		founder_by_class.clear();                                             // :431-436
		for (u32 f(0); f < founders; ++f) founder_by_class.emplace(slot(row, f), f);
	}
};

} // namespace


namespace {

bool find_matchings_sequential(
	variant_graph const &graph, std::vector<u64> const &cut_positions, u32 founder_count, bool keep_ref_edges,
	std::vector<u32> &assigned)
{
	u32 const copies(graph.total_chromosome_copies());
	if (cut_positions.size() < 2 || 0 == copies) return false;               // :163-167

	std::size_t const rows(cut_positions.size() - 1);
	assigned.assign(rows * founder_count, kPloidyMax);                         // :171-172
	matcher m{founder_count, rows, assigned, {}, std::vector<char>(copies, 0), {}, {}};

	auto const &paths(graph.paths_by_edge_and_chrom_copy);
	u64 const words_per_column(paths.words_per_column());
	check_edge_range(graph);
	edge_pbwt pbwt(copies, nullptr);
	std::vector<u32> lhs_class(copies, kPloidyMax), rhs_class(copies, kPloidyMax);
	std::vector<joined_class> joined;
	u32 lhs_distinct(0), rhs_distinct(0), lhs_first_class(0), rhs_first_class(0);
	bool lhs_first_is_ref(true), rhs_first_is_ref(true);
	u64 edge(0), prev_cut_edge(0), cut_pair_edge(0);
	std::size_t next_cut(1), cuts_seen(0);

	for (u64 node(0); node < graph.node_count(); ++node) {
		if (next_cut < cut_positions.size() && node == cut_positions[next_cut]) {
			// classes of the block that ends here, and of the two-block span ending here (:215-251)
			lhs_class.swap(rhs_class);
			std::fill(rhs_class.begin(), rhs_class.end(), kPloidyMax);
			lhs_distinct = rhs_distinct;
			lhs_first_class = rhs_first_class;
			rhs_distinct = 0;
			rhs_first_class = pbwt.order.front();
			u32 rep(kPloidyMax);
			joined.clear();
			for (u32 i(0); i < copies; ++i) {
				u32 const copy(pbwt.order[i]);
				u64 const d(unbiased(pbwt.divergence[i]));
				if (prev_cut_edge < d) { rep = copy; ++rhs_distinct; }        // plain integer comparison: "no match yet" starts a class
				rhs_class[copy] = rep;
				if (cuts_seen) {
					if (cut_pair_edge < d) joined.push_back({lhs_class[copy], rep, 0});
					++joined.back().size;
				}
			}

			if (cuts_seen) {
				std::sort(joined.begin(), joined.end());                      // by size, smallest first; ties as std::sort leaves them (:256)
				if (!keep_ref_edges && lhs_first_is_ref && rhs_first_is_ref)  // :258-264
					std::erase_if(joined, [&](joined_class const &c) { return c.lhs_rep == lhs_first_class && c.rhs_rep == rhs_first_class; });
				if (1 == cuts_seen) m.seed(joined, lhs_distinct);
				m.extend(cuts_seen, joined, rhs_distinct);
			}

			++cuts_seen;                                                      // :444-451
			++next_cut;
			cut_pair_edge = prev_cut_edge;
			prev_cut_edge = edge;
			lhs_first_is_ref = rhs_first_is_ref;
			rhs_first_is_ref = true;
		}

		for (u64 e(graph.alt_edge_count_csum[node]); e < graph.alt_edge_count_csum[node + 1]; ++e) {   // :454-462
			pbwt.advance(edge_column(graph, edge), words_per_column, edge);
			rhs_first_is_ref = rhs_first_is_ref && !paths.test(pbwt.order.front(), edge);
			++edge;
		}
	}

	m.finish();
	if (1 == cuts_seen) {                                                     // a single block (:468-507)
		u32 rep(kPloidyMax);
		joined.clear();
		for (u32 i(0); i < copies; ++i) {
			if (0 < unbiased(pbwt.divergence[i])) { rep = pbwt.order[i]; ++rhs_distinct; joined.push_back({kPloidyMax, rep, 0}); }
			rhs_class[pbwt.order[i]] = rep;
			++joined.back().size;
		}
		std::sort(joined.begin(), joined.end());
		if (!keep_ref_edges && rhs_first_is_ref)
			std::erase_if(joined, [&](joined_class const &c) { return c.rhs_rep == rhs_first_class; });
		u32 founder(0);
		for (auto c(joined.rbegin()); c != joined.rend() && founder < founder_count; ++c, ++founder)
			assigned[founder * rows + 0] = c->rhs_rep;
	}
	return true;
}


// Chunks produced by `threads` workers in any order, consumed by the calling thread in chunk order; a worker does not
// start chunk c before chunk c - window has been consumed, so at most `window` produced chunks are alive at a time.
template <typename Produce, typename Consume>
void run_chunk_pipeline(std::size_t n_chunks, unsigned threads, std::size_t window, Produce &&produce, Consume &&consume)
{
	std::mutex mutex;
	std::condition_variable changed;
	std::size_t next_claim(0), consumed(0);
	std::vector<char> done(n_chunks, 0);
	std::exception_ptr error;
	auto const work([&] {
		for (;;) {
			std::size_t c;
			{
				std::unique_lock<std::mutex> lock(mutex);
				changed.wait(lock, [&] { return error || next_claim >= n_chunks || next_claim < consumed + window; });
				if (error || next_claim >= n_chunks) return;
				c = next_claim++;
			}
			try {
				produce(c);
			} catch (...) {
				std::lock_guard<std::mutex> lock(mutex);
				if (!error) error = std::current_exception();
			}
			{
				std::lock_guard<std::mutex> lock(mutex);
				done[c] = 1;
			}
			changed.notify_all();
		}
	});
	std::vector<std::thread> pool;
	for (unsigned t(0); t < threads; ++t) pool.emplace_back(work);
	for (std::size_t c(0); c < n_chunks; ++c) {
		{
			std::unique_lock<std::mutex> lock(mutex);
			changed.wait(lock, [&] { return error || done[c]; });
			if (error) break;
		}
		try {
			consume(c);
		} catch (...) {
			std::lock_guard<std::mutex> lock(mutex);
			if (!error) error = std::current_exception();
			break;
		}
		{
			std::lock_guard<std::mutex> lock(mutex);
			consumed = c + 1;
		}
		changed.notify_all();
	}
	{
		std::lock_guard<std::mutex> lock(mutex);
		if (error) next_claim = n_chunks;                                       // nobody claims anything any more
	}
	changed.notify_all();
	for (auto &t : pool) t.join();
	if (error) std::rethrow_exception(error);
}



// ---------------------------------------------------------------------------------------------------------------------
// The same matching with the expensive part -- the pBWT over all edges and the path classes at every cut -- spread over
// threads.  What makes that possible: the pBWT state after the first k edges does not have to be reached step by
// step.  The order is the copies sorted by their reversed k-edge prefixes (read as a k-bit integer with edge k - 1 as the
// most significant bit; ties keep the copy order), and a copy's divergence value is one past the highest edge on which
// it differs from its predecessor in that order (0 if they agree everywhere, k for the first copy) -- exactly what k
// steps of pbwt.hh:77-134 leave behind.  Both are read off paths_by_chrom_copy_and_edge, where a copy's prefix is a run
// of words.  A chunk of consecutive cuts therefore starts from a state built from scratch, walks its own edges and
// leaves one record per cut; the greedy assignment itself (cheap, strictly sequential) then consumes the records in
// cut order.
// ---------------------------------------------------------------------------------------------------------------------
void pbwt_state_at(variant_graph const &graph, u64 k, edge_pbwt &pbwt)
{
	if (0 == k) return;                                                       // the constructor's state
	auto const &m(graph.paths_by_chrom_copy_and_edge);                        // rows = edges, cols = copies
	u64 const wpc(m.words_per_column());
	u64 const top((k - 1) >> 6);
	u64 const top_mask((k & 63) ? (u64(1) << (k & 63)) - 1 : ~u64(0));
	u64 const *const words(m.words.data());
	u32 const n(u32(pbwt.order.size()));
	auto const prefix_word([&](u32 copy, u64 w) { u64 const x(words[copy * wpc + w]); return w == top ? x & top_mask : x; });

	std::iota(pbwt.order.begin(), pbwt.order.end(), 0u);
	std::stable_sort(pbwt.order.begin(), pbwt.order.end(), [&](u32 a, u32 b) {
		for (u64 w(top + 1); w-- > 0;) {
			u64 const x(prefix_word(a, w)), y(prefix_word(b, w));
			if (x != y) return x < y;
		}
		return false;
	});
	pbwt.divergence[0] = u32(k) + 1;                                          // biased k
	for (u32 i(1); i < n; ++i) {
		u32 const a(pbwt.order[i - 1]), b(pbwt.order[i]);
		u32 d(1);                                                             // biased 0: the two agree on every edge so far
		for (u64 w(top + 1); w-- > 0;) {
			u64 const diff(prefix_word(a, w) ^ prefix_word(b, w));
			if (diff) { d = u32(64 * w + 63 - u64(__builtin_clzll(diff))) + 2; break; }   // one past the edge, biased
		}
		pbwt.divergence[i] = d;
	}
}


// What the sequential loop knows at one cut position (founder_sequence_greedy_output.cc:208-264), minus the assignment.
struct cut_record {
	std::size_t joined_begin, joined_end;   // into the chunk's pool; sorted by size
	u32 rhs_distinct;
	u32 rhs_first_class;
	bool rhs_first_is_ref;                  // over the edges since the previous cut
};

struct cut_chunk {
	std::size_t first_cut{}, end_cut{};     // cut indices [first_cut, end_cut), first_cut >= 1
	std::vector<joined_class> pool;
	std::vector<cut_record> records;
};

void scan_cut_chunk(variant_graph const &graph, std::vector<u64> const &cut_positions, cut_chunk &chunk)
{
	u32 const copies(graph.total_chromosome_copies());
	auto const &paths(graph.paths_by_edge_and_chrom_copy);
	u64 const words_per_column(paths.words_per_column());
	auto const edge_at([&](std::size_t cut) { return graph.alt_edge_count_csum[cut_positions[cut]]; });   // edges before the cut node

	edge_pbwt pbwt(copies, nullptr);
	std::vector<u32> lhs_class(copies, kPloidyMax), rhs_class(copies, kPloidyMax);
	std::size_t const start_cut(chunk.first_cut - 1);
	u64 edge(edge_at(start_cut));
	pbwt_state_at(graph, edge, pbwt);
	if (start_cut >= 1) {                                                     // the classes the previous cut left behind
		u64 const threshold(edge_at(start_cut - 1));
		u32 rep(kPloidyMax);
		for (u32 i(0); i < copies; ++i) {
			if (threshold < unbiased(pbwt.divergence[i])) rep = pbwt.order[i];
			rhs_class[pbwt.order[i]] = rep;
		}
	}

	bool rhs_first_is_ref(true);
	std::size_t next_cut(chunk.first_cut);
	u64 const last_node(cut_positions[chunk.end_cut - 1]);
	for (u64 node(cut_positions[start_cut]); node <= last_node; ++node) {
		if (next_cut < chunk.end_cut && node == cut_positions[next_cut]) {
			u64 const prev_cut_edge(edge_at(next_cut - 1));
			u64 const cut_pair_edge(next_cut >= 2 ? edge_at(next_cut - 2) : 0);
			lhs_class.swap(rhs_class);
			cut_record rec{chunk.pool.size(), 0, 0, pbwt.order.front(), rhs_first_is_ref};
			u32 rep(kPloidyMax);
			for (u32 i(0); i < copies; ++i) {
				u32 const copy(pbwt.order[i]);
				u64 const d(unbiased(pbwt.divergence[i]));
				if (prev_cut_edge < d) { rep = copy; ++rec.rhs_distinct; }
				rhs_class[copy] = rep;
				if (next_cut >= 2) {
					if (cut_pair_edge < d) chunk.pool.push_back({lhs_class[copy], rep, 0});
					++chunk.pool.back().size;
				}
			}
			rec.joined_end = chunk.pool.size();
			std::sort(chunk.pool.begin() + std::ptrdiff_t(rec.joined_begin), chunk.pool.end());   // :256; same input order as the sequential loop
			chunk.records.push_back(rec);
			++next_cut;
			rhs_first_is_ref = true;
			if (node == last_node) break;
		}
		for (u64 e(graph.alt_edge_count_csum[node]); e < graph.alt_edge_count_csum[node + 1]; ++e) {
			pbwt.advance(edge_column(graph, edge), words_per_column, edge);
			rhs_first_is_ref = rhs_first_is_ref && !paths.test(pbwt.order.front(), edge);
			++edge;
		}
	}
}


bool find_matchings_chunked(
	variant_graph const &graph, std::vector<u64> const &cut_positions, u32 founder_count, bool keep_ref_edges,
	std::vector<u32> &assigned, unsigned threads)
{
	u32 const copies(graph.total_chromosome_copies());
	std::size_t const n_cuts(cut_positions.size());
	std::size_t const rows(n_cuts - 1);
	assigned.assign(rows * founder_count, kPloidyMax);
	matcher m{founder_count, rows, assigned, {}, std::vector<char>(copies, 0), {}, {}};

	// chunks of consecutive cuts with about the same number of edges each, several per thread
	u64 const n_edges(graph.edge_count());
	std::size_t const wanted(std::min<std::size_t>(std::max<std::size_t>(std::size_t(threads) * 8, std::size_t(n_edges / 8192) + 1), n_cuts - 1));
	std::vector<cut_chunk> chunks;
	{
		std::size_t k(1);
		for (std::size_t j(1); j <= wanted && k < n_cuts; ++j) {
			u64 const edge_goal(j == wanted ? n_edges + 1 : n_edges * j / wanted);
			std::size_t end(k + 1);
			while (end < n_cuts && graph.alt_edge_count_csum[cut_positions[end]] < edge_goal) ++end;
			if (j == wanted) end = n_cuts;
			chunks.emplace_back();
			chunks.back().first_cut = k;
			chunks.back().end_cut = end;
			k = end;
		}
	}

	// the assignment, in cut order (founder_sequence_greedy_output.cc:254-457), while the workers scan the chunks ahead
	u32 lhs_distinct(0), rhs_distinct(0), lhs_first_class(0), rhs_first_class(0);
	bool lhs_first_is_ref(true);
	std::vector<joined_class> joined;
	std::size_t cuts_seen(0);
	run_chunk_pipeline(chunks.size(), threads, std::size_t(threads) * 2 + 2,
		[&](std::size_t c) { scan_cut_chunk(graph, cut_positions, chunks[c]); },
		[&](std::size_t c) {
			auto &chunk(chunks[c]);
			for (auto const &rec : chunk.records) {
				lhs_distinct = rhs_distinct;
				lhs_first_class = rhs_first_class;
				rhs_distinct = rec.rhs_distinct;
				rhs_first_class = rec.rhs_first_class;
				if (cuts_seen) {
					joined.assign(chunk.pool.begin() + std::ptrdiff_t(rec.joined_begin), chunk.pool.begin() + std::ptrdiff_t(rec.joined_end));
					if (!keep_ref_edges && lhs_first_is_ref && rec.rhs_first_is_ref)      // :258-264
						std::erase_if(joined, [&](joined_class const &jc) { return jc.lhs_rep == lhs_first_class && jc.rhs_rep == rhs_first_class; });
					if (1 == cuts_seen) m.seed(joined, lhs_distinct);
					m.extend(cuts_seen, joined, rhs_distinct);
				}
				++cuts_seen;
				lhs_first_is_ref = rec.rhs_first_is_ref;
			}
			std::vector<joined_class>().swap(chunk.pool);
			std::vector<cut_record>().swap(chunk.records);
		});
	m.finish();
	return true;
}

} // namespace


namespace {

// V2M_FOUNDER_TIMING=1: where the walked searches spend their time, on stderr.
struct phase_timer {
	bool const on{nullptr != std::getenv("V2M_FOUNDER_TIMING")};
	std::chrono::steady_clock::time_point last{std::chrono::steady_clock::now()};
	void mark(char const *what)
	{
		if (!on) return;
		auto const now(std::chrono::steady_clock::now());
		std::fprintf(stderr, "[founder] %-40s %7.3f s\n", what, std::chrono::duration<double>(now - last).count());
		last = now;
	}
};

// The walkers' output areas: hundreds of MB that are written once (by a device-to-host copy) and read once.  As ordinary
// pages, touching them for the first time and giving them back costs more than using them (config 4: 800 MB of pairs, 0.15 s to
// release alone); mapped with 2-MB pages both costs all but vanish.
class huge_u32_buffer {
public:
	explicit huge_u32_buffer(std::size_t count)
	{
		std::size_t const huge(std::size_t(2) << 20);
		m_bytes = (std::max<std::size_t>(count, 1) * sizeof(u32) + huge - 1) / huge * huge;
		void *const p(::mmap(nullptr, m_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
		if (MAP_FAILED == p) throw std::bad_alloc();
		(void) ::madvise(p, m_bytes, MADV_HUGEPAGE);             // (advice: without transparent huge pages it is an ordinary mapping)
		m_p = static_cast<u32 *>(p);
	}
	~huge_u32_buffer() { if (m_p) ::munmap(m_p, m_bytes); }
	huge_u32_buffer(huge_u32_buffer const &) = delete;
	huge_u32_buffer &operator=(huge_u32_buffer const &) = delete;
	u32 *get() const { return m_p; }
private:
	u32 *m_p{};
	std::size_t m_bytes{};
};

// pBWT states after the given numbers of edges, built from scratch on `threads` threads (pbwt_state_at).
void build_states(variant_graph const &graph, std::vector<u32> const &edges, unsigned threads, std::unique_ptr<u32[]> &order, std::unique_ptr<u32[]> &divergence)
{
	u32 const copies(graph.total_chromosome_copies());
	order.reset(new u32[edges.size() * copies]);
	divergence.reset(new u32[edges.size() * copies]);
	std::atomic<std::size_t> next(0);
	std::exception_ptr error;
	std::mutex error_mutex;
	auto const work([&] {
		try {
			for (std::size_t c; (c = next.fetch_add(1)) < edges.size();) {
				edge_pbwt pbwt(copies, nullptr);
				pbwt_state_at(graph, edges[c], pbwt);
				std::copy(pbwt.order.begin(), pbwt.order.end(), order.get() + c * copies);
				std::copy(pbwt.divergence.begin(), pbwt.divergence.end(), divergence.get() + c * copies);
			}
		} catch (...) {
			std::lock_guard<std::mutex> const lock(error_mutex);
			if (!error) error = std::current_exception();
		}
	});
	std::vector<std::thread> pool;
	for (unsigned t(1); t < std::max(1u, threads); ++t) pool.emplace_back(work);
	work();
	for (auto &t : pool) t.join();
	if (error) std::rethrow_exception(error);
}


// find_matchings_chunked() with the chunk walks done by `walker`: it leaves, per cut, the joined classes in pBWT order; they
// are sorted here (the reference's std::sort on the reference's input order, founder_sequence_greedy_output.cc:256) on the
// threads, and the assignment consumes them in cut order.
bool find_matchings_walked(
	variant_graph const &graph, std::vector<u64> const &cut_positions, u32 founder_count, bool keep_ref_edges,
	std::vector<u32> &assigned, unsigned threads, founder_walker &walker)
{
	u32 const copies(graph.total_chromosome_copies());
	std::size_t const n_cuts(cut_positions.size());
	std::size_t const rows(n_cuts - 1);
	u64 const n_edges(graph.edge_count());
	check_edge_range(graph);
	assigned.assign(rows * founder_count, kPloidyMax);
	matcher m{founder_count, rows, assigned, {}, std::vector<char>(copies, 0), {}, {}};

	phase_timer timer;
	std::vector<u32> cut_edge(n_cuts);
	for (std::size_t j(0); j < n_cuts; ++j) cut_edge[j] = u32(graph.alt_edge_count_csum[cut_positions[j]]);

	// start states: the cut search's if it has just run on this graph, else one every 8192 edges
	if (walker.state_edge.empty() || walker.state_copies != copies || 0 != walker.state_edge.front()
		|| walker.state_edges != n_edges || walker.state_matrix != static_cast<void const *>(graph.paths_by_chrom_copy_and_edge.words.data())) {
		walker.state_edge.clear();
		for (u64 e(0); e < std::max<u64>(1, n_edges); e += 8192) walker.state_edge.push_back(u32(e));
		build_states(graph, walker.state_edge, threads, walker.state_order, walker.state_divergence);
		walker.state_copies = copies;
		walker.state_edges = n_edges;
		walker.state_matrix = graph.paths_by_chrom_copy_and_edge.words.data();
	}
	timer.mark("matching: start states");
	std::size_t const n_chunks(walker.state_edge.size());
	// chunk k: from the first cut at or past its state (the walker steps on to it), the cuts after that one up to and including
	// the first cut at or past the next state
	std::vector<u64> chunk_first_cut(n_chunks + 1);
	for (std::size_t k(0); k < n_chunks; ++k) {
		std::size_t const start_cut(std::size_t(std::lower_bound(cut_edge.begin(), cut_edge.end(), walker.state_edge[k]) - cut_edge.begin()));
		chunk_first_cut[k] = std::min<u64>(start_cut + 1, n_cuts);
	}
	chunk_first_cut[n_chunks] = n_cuts;
	for (std::size_t k(n_chunks); k-- > 0;) chunk_first_cut[k] = std::min(chunk_first_cut[k], chunk_first_cut[k + 1]);

	u64 max_chunk_cuts(0);
	for (std::size_t k(0); k < n_chunks; ++k) max_chunk_cuts = std::max<u64>(max_chunk_cuts, chunk_first_cut[k + 1] - chunk_first_cut[k]);
	u64 pool_capacity(4096 + 128 * max_chunk_cuts);
	if (char const *const e = std::getenv("V2M_FOUNDER_POOL_CAPACITY")) if (*e) pool_capacity = std::max<u64>(1, std::strtoull(e, nullptr, 10));   // test knob
	huge_u32_buffer const pool_lhs(n_chunks * pool_capacity), pool_rhs(n_chunks * pool_capacity), pool_size(n_chunks * pool_capacity);
	std::vector<u64> rec_pool_end(n_cuts, 0);
	std::vector<u32> rec_distinct(n_cuts, 0), rec_first_class(n_cuts, 0), rec_first_is_ref(n_cuts, 0), status(n_chunks, 1);
	walker.records(copies, cut_edge, chunk_first_cut, walker.state_edge, walker.state_order.get(), walker.state_divergence.get(),
		pool_capacity, pool_lhs.get(), pool_rhs.get(), pool_size.get(), rec_pool_end.data(), rec_distinct.data(), rec_first_class.data(), rec_first_is_ref.data(), status.data());
	// The states are good for ONE matching after the search that left them: they are keyed by the graph's shape and the address of its
	// matrix, which a matrix rewritten in place (or a new one allocated where the old one was) would still match.  Used once, then dropped.
	walker.state_edge.clear();
	walker.state_order.reset();
	walker.state_divergence.reset();
	walker.state_matrix = nullptr;
	timer.mark("matching: chunk walks (walker)");
	if (walker.on_last_walk) walker.on_last_walk();

	// chunks left undone: the host's own chunk scan; the others: joined classes gathered and sorted per cut, on the threads
	std::vector<cut_chunk> chunks(n_chunks);
	walker.chunks_walked = walker.chunks_left = 0;
	for (std::size_t k(0); k < n_chunks; ++k) {
		chunks[k].first_cut = chunk_first_cut[k];
		chunks[k].end_cut = chunk_first_cut[k + 1];
		if (chunks[k].first_cut < chunks[k].end_cut) ++(0 == status[k] ? walker.chunks_walked : walker.chunks_left);
	}
	// The threads take the chunks in order and leave each one ready (joined classes gathered, sorted per cut); this thread runs the
	// assignment, in cut order (founder_sequence_greedy_output.cc:254-457), over every chunk as soon as it is ready: the sorting
	// disappears behind the assignment.
	{
		std::atomic<std::size_t> next(0);
		std::unique_ptr<std::atomic<unsigned char>[]> ready(new std::atomic<unsigned char>[n_chunks]);
		for (std::size_t k(0); k < n_chunks; ++k) ready[k].store(0, std::memory_order_relaxed);
		std::atomic<bool> failed(false);
		std::exception_ptr error;
		std::mutex error_mutex;
		auto const work([&] {
			try {
				for (std::size_t k; (k = next.fetch_add(1)) < n_chunks;) {
					cut_chunk &chunk(chunks[k]);
					if (chunk.first_cut < chunk.end_cut) {
						if (0 != status[k]) scan_cut_chunk(graph, cut_positions, chunk);
						else {
							u32 const *const lhs(pool_lhs.get() + k * pool_capacity), *const rhs(pool_rhs.get() + k * pool_capacity), *const size(pool_size.get() + k * pool_capacity);
							chunk.pool.resize(rec_pool_end[chunk.end_cut - 1]);
							for (std::size_t i(0); i < chunk.pool.size(); ++i) chunk.pool[i] = {lhs[i], rhs[i], size[i]};
							u64 begin(0);
							for (std::size_t cut(chunk.first_cut); cut < chunk.end_cut; ++cut) {
								u64 const end(rec_pool_end[cut]);
								std::sort(chunk.pool.begin() + std::ptrdiff_t(begin), chunk.pool.begin() + std::ptrdiff_t(end));   // :256
								chunk.records.push_back({std::size_t(begin), std::size_t(end), rec_distinct[cut], rec_first_class[cut], 0 != rec_first_is_ref[cut]});
								begin = end;
							}
						}
					}
					ready[k].store(1, std::memory_order_release);
				}
			} catch (...) {
				std::lock_guard<std::mutex> const lock(error_mutex);
				if (!error) error = std::current_exception();
				failed.store(true);
			}
		});
		std::vector<std::thread> pool;
		for (unsigned t(0); t < std::max(1u, threads); ++t) pool.emplace_back(work);

		u32 lhs_distinct(0), rhs_distinct(0), lhs_first_class(0), rhs_first_class(0);
		bool lhs_first_is_ref(true);
		std::vector<joined_class> joined;
		std::size_t cuts_seen(0);
		try {
			for (std::size_t k(0); k < n_chunks && !failed.load(); ++k) {
				while (!ready[k].load(std::memory_order_acquire)) { if (failed.load()) break; std::this_thread::yield(); }
				if (failed.load()) break;
				cut_chunk &chunk(chunks[k]);
				for (auto const &rec : chunk.records) {
					lhs_distinct = rhs_distinct;
					lhs_first_class = rhs_first_class;
					rhs_distinct = rec.rhs_distinct;
					rhs_first_class = rec.rhs_first_class;
					if (cuts_seen) {
						joined.assign(chunk.pool.begin() + std::ptrdiff_t(rec.joined_begin), chunk.pool.begin() + std::ptrdiff_t(rec.joined_end));
						if (!keep_ref_edges && lhs_first_is_ref && rec.rhs_first_is_ref)      // :258-264
							std::erase_if(joined, [&](joined_class const &jc) { return jc.lhs_rep == lhs_first_class && jc.rhs_rep == rhs_first_class; });
						if (1 == cuts_seen) m.seed(joined, lhs_distinct);
						m.extend(cuts_seen, joined, rhs_distinct);
					}
					++cuts_seen;
					lhs_first_is_ref = rec.rhs_first_is_ref;
				}
				std::vector<joined_class>().swap(chunk.pool);
				std::vector<cut_record>().swap(chunk.records);
			}
		} catch (...) {
			std::lock_guard<std::mutex> const lock(error_mutex);
			if (!error) error = std::current_exception();
			failed.store(true);
		}
		if (failed.load()) next.store(n_chunks);                                  // nobody starts another chunk
		for (auto &t : pool) t.join();
		if (error) std::rethrow_exception(error);
	}
	m.finish();
	timer.mark("matching: sort per cut + greedy assignment");
	return true;
}

} // namespace


bool find_matchings(
	variant_graph const &graph, std::vector<u64> const &cut_positions, u32 founder_count, bool keep_ref_edges,
	std::vector<u32> &assigned, unsigned threads, founder_walker *walker)
{
	u32 const copies(graph.total_chromosome_copies());
	if (cut_positions.size() < 2 || 0 == copies) return false;               // :163-167
	if (0 == threads) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
	auto const &transposed(graph.paths_by_chrom_copy_and_edge);
	bool const have_transposed(transposed.cols >= copies && transposed.rows >= graph.edge_count() && !transposed.words.empty());
	if (!walker || cut_positions.size() <= 2 || copies > walker->max_copies_records() || 0 == graph.edge_count() || !have_transposed)
		return find_matchings(graph, cut_positions, founder_count, keep_ref_edges, assigned, threads);
	return find_matchings_walked(graph, cut_positions, founder_count, keep_ref_edges, assigned, threads, *walker);
}


bool find_matchings(
	variant_graph const &graph, std::vector<u64> const &cut_positions, u32 founder_count, bool keep_ref_edges,
	std::vector<u32> &assigned, unsigned threads)
{
	u32 const copies(graph.total_chromosome_copies());
	if (cut_positions.size() < 2 || 0 == copies) return false;               // :163-167
	if (0 == threads) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
	auto const &transposed(graph.paths_by_chrom_copy_and_edge);
	bool const have_transposed(transposed.cols >= copies && transposed.rows >= graph.edge_count() && !transposed.words.empty());
	// a single block (two cut positions) takes the reference's separate path (:475-508); so does a search without the
	// copy-major matrix, or on one thread
	if (threads <= 1 || cut_positions.size() <= 2 || !have_transposed)
		return find_matchings_sequential(graph, cut_positions, founder_count, keep_ref_edges, assigned);
	return find_matchings_chunked(graph, cut_positions, founder_count, keep_ref_edges, assigned, threads);
}


namespace {

// ---------------------------------------------------------------------------------------------------------------------
// The cut search over threads.  Which nodes are candidates, and which earlier candidate a divergence value points to,
// depend on the graph alone; the scores do not, but they are cheap.  So the edges are cut into chunks; a worker builds
// the pBWT state at the start of its chunk from scratch (pbwt_state_at), walks its edges and, at every candidate node,
// writes down the (earlier candidate, class count) pairs the reference would try, in its order; the calling thread
// consumes the chunks in order and does nothing but the score updates (find_cut_positions.cc:55-63).  At most
// `window` finished chunks wait for it at any time.
// ---------------------------------------------------------------------------------------------------------------------
struct cut_trial { u32 pred; u32 class_count; };

struct cut_search_chunk {
	std::size_t first{}, end{};            // candidates [first, end), indices into the candidate list (>= 1)
	std::vector<cut_trial> trials;
	std::vector<std::size_t> trial_end;    // per candidate
};

void scan_cut_search_chunk(
	variant_graph const &graph, u64 min_distance, std::vector<cut_candidate> const &cuts, std::vector<u32> const &first_candidate_from_edge,
	cut_search_chunk &chunk)
{
	u32 const copies(graph.total_chromosome_copies());
	u64 const words_per_column(graph.paths_by_edge_and_chrom_copy.words_per_column());
	u64 const n_edges(graph.edge_count());
	divergence_counts counts(n_edges + 2);
	edge_pbwt pbwt(copies, nullptr);
	u64 edge(cuts[chunk.first].edge);
	pbwt_state_at(graph, edge, pbwt);
	if (0 == edge) { counts.add(1, 1); if (copies > 1) counts.add(0, copies - 1); }   // the initial state (pbwt.hh:62-70)
	else for (u32 const d : pbwt.divergence) counts.add(d, 1);
	pbwt.follow(&counts);

	std::size_t next(chunk.first);
	u64 const last_node(cuts[chunk.end - 1].node);
	for (u64 node(cuts[chunk.first].node); node <= last_node; ++node) {
		if (next < chunk.end && node == cuts[next].node) {
			std::size_t right_bound(next + 1);                                 // the candidate itself is already in the list (:129,135)
			u32 value(counts.largest());
			u32 class_count(counts[value]);
			for (value = counts.below(value); kNoValue != value; value = counts.below(value)) {
				u64 const v(unbiased(value));
				std::size_t const pred(v <= n_edges ? std::min<std::size_t>(first_candidate_from_edge[v], right_bound) : right_bound);
				if (pred != right_bound) {
					right_bound = pred;
					// (trying the candidate itself never changes anything: its score is the number of copies)
					if (pred != next && min_distance <= graph.aligned_positions[node] - graph.aligned_positions[cuts[pred].node])
						chunk.trials.push_back({u32(pred), class_count});
				}
				class_count += counts[value];
			}
			if (0 != right_bound && right_bound - 1 != next) chunk.trials.push_back({u32(right_bound - 1), class_count});
			chunk.trial_end.push_back(chunk.trials.size());
			++next;
			if (node == last_node) break;
		}
		for (u64 e(graph.alt_edge_count_csum[node]); e < graph.alt_edge_count_csum[node + 1]; ++e) {
			pbwt.advance(edge_column(graph, edge), words_per_column, edge);
			++edge;
		}
	}
}

u32 find_cut_positions_chunked(variant_graph const &graph, u64 min_distance, std::vector<u64> &out, unsigned threads)
{
	out.clear();
	u32 const copies(graph.total_chromosome_copies());
	u64 const n_edges(graph.edge_count());
	check_edge_range(graph);

	// the candidates: bridge nodes, one per distinct edge index (find_cut_positions.cc:111-112,126-131)
	std::vector<cut_candidate> cuts;
	cuts.push_back({0, kEdgeMax, 0, 0});
	{
		u64 rightmost_target(0), edge(0), last_cut_edge(kEdgeMax);
		for (u64 node(0); node < graph.node_count(); ++node) {
			if (rightmost_target <= node && last_cut_edge != edge) {
				cuts.push_back({edge, kEdgeMax, node, copies});
				last_cut_edge = edge;
			}
			for (u64 e(graph.alt_edge_count_csum[node]); e < graph.alt_edge_count_csum[node + 1]; ++e, ++edge)
				rightmost_target = std::max(rightmost_target, graph.alt_edge_targets[e]);
		}
	}
	if (cuts.size() >= u64(UINT32_MAX)) throw std::length_error("founder search: candidate indices are kept in 32 bits");
	// first candidate whose edge index is >= e (what std::lower_bound over the whole list returns)
	std::vector<u32> first_candidate_from_edge(n_edges + 1);
	{
		std::size_t c(0);
		for (u64 e(0); e <= n_edges; ++e) {
			while (c < cuts.size() && cuts[c].edge < e) ++c;
			first_candidate_from_edge[e] = u32(c);
		}
	}

	// chunks of candidates with about the same number of edges each
	std::size_t const n_cand(cuts.size());
	std::size_t const wanted(std::min<std::size_t>(std::max<std::size_t>(std::size_t(threads) * 8, std::size_t(n_edges / 8192) + 1), n_cand - 1));
	std::vector<cut_search_chunk> chunks;
	for (std::size_t j(1), k(1); j <= wanted && k < n_cand; ++j) {
		u64 const edge_goal(n_edges * j / wanted);
		std::size_t end(k + 1);
		while (end < n_cand && cuts[end].edge < edge_goal) ++end;
		if (j == wanted) end = n_cand;
		chunks.emplace_back();
		chunks.back().first = k;
		chunks.back().end = end;
		k = end;
	}

	run_chunk_pipeline(chunks.size(), threads, std::size_t(threads) * 2 + 2,
		[&](std::size_t c) { scan_cut_search_chunk(graph, min_distance, cuts, first_candidate_from_edge, chunks[c]); },
		[&](std::size_t c) {
			auto &chunk(chunks[c]);
			std::size_t t(0);
			for (std::size_t j(chunk.first); j < chunk.end; ++j) {
				cut_candidate &current(cuts[j]);
				for (std::size_t const t_end(chunk.trial_end[j - chunk.first]); t < t_end; ++t)
					current.improve(chunk.trials[t].class_count, cuts[chunk.trials[t].pred]);
			}
			std::vector<cut_trial>().swap(chunk.trials);
			std::vector<std::size_t>().swap(chunk.trial_end);
		});
	return collect_cut_positions(cuts, graph, out);
}

} // namespace


namespace {

// find_cut_positions_chunked() with the chunk walks done by `walker`: candidates and chunks as there, every chunk's start
// state built on the host threads first, one call for all the walks, then the score updates in candidate order.
u32 find_cut_positions_walked(variant_graph const &graph, u64 min_distance, std::vector<u64> &out, unsigned threads, founder_walker &walker)
{
	out.clear();
	u32 const copies(graph.total_chromosome_copies());
	u64 const n_edges(graph.edge_count());
	check_edge_range(graph);

	std::vector<cut_candidate> cuts;
	cuts.push_back({0, kEdgeMax, 0, 0});
	{
		u64 rightmost_target(0), edge(0), last_cut_edge(kEdgeMax);
		for (u64 node(0); node < graph.node_count(); ++node) {
			if (rightmost_target <= node && last_cut_edge != edge) {
				cuts.push_back({edge, kEdgeMax, node, copies});
				last_cut_edge = edge;
			}
			for (u64 e(graph.alt_edge_count_csum[node]); e < graph.alt_edge_count_csum[node + 1]; ++e, ++edge)
				rightmost_target = std::max(rightmost_target, graph.alt_edge_targets[e]);
		}
	}
	if (cuts.size() >= u64(UINT32_MAX)) throw std::length_error("founder search: candidate indices are kept in 32 bits");
	std::size_t const n_cand(cuts.size());
	if (n_cand <= 1) return collect_cut_positions(cuts, graph, out);

	// chunks of candidates with about the same number of edges each: one workgroup walks one chunk, so the walker says how many
	// it wants (a start state costs the host well under a millisecond of one thread at 5000 copies), but never fewer than a few
	// hundred edges each
	std::size_t const by_size(std::max<std::size_t>(std::size_t(threads) * 8, std::size_t(n_edges / 8192) + 1));
	std::size_t const by_walker(std::min<std::size_t>(walker.preferred_chunks(), std::size_t(n_edges / 512) + 1));
	std::size_t const wanted(std::min<std::size_t>(std::max(by_size, by_walker), n_cand - 1));
	std::vector<u64> chunk_first;
	for (std::size_t j(1), k(1); j <= wanted && k < n_cand; ++j) {
		u64 const edge_goal(n_edges * j / wanted);
		std::size_t end(k + 1);
		while (end < n_cand && cuts[end].edge < edge_goal) ++end;
		if (j == wanted) end = n_cand;
		chunk_first.push_back(k);
		k = end;
	}
	chunk_first.push_back(n_cand);
	std::size_t const n_chunks(chunk_first.size() - 1);

	phase_timer timer;
	std::vector<u32> cand_edge(n_cand);
	std::vector<u64> cand_aligned(n_cand);
	for (std::size_t c(0); c < n_cand; ++c) { cand_edge[c] = u32(cuts[c].edge); cand_aligned[c] = graph.aligned_positions[cuts[c].node]; }

	// start states, on the host threads
	std::unique_ptr<u32[]> start_order(new u32[n_chunks * copies]), start_div(new u32[n_chunks * copies]);
	{
		std::atomic<std::size_t> next_chunk(0);
		std::exception_ptr error;
		std::mutex error_mutex;
		auto const work([&] {
			try {
				edge_pbwt pbwt(copies, nullptr);
				for (std::size_t c; (c = next_chunk.fetch_add(1)) < n_chunks;) {
					pbwt = edge_pbwt(copies, nullptr);
					pbwt_state_at(graph, cuts[chunk_first[c]].edge, pbwt);
					std::copy(pbwt.order.begin(), pbwt.order.end(), start_order.get() + c * copies);
					std::copy(pbwt.divergence.begin(), pbwt.divergence.end(), start_div.get() + c * copies);
				}
			} catch (...) {
				std::lock_guard<std::mutex> const lock(error_mutex);
				if (!error) error = std::current_exception();
			}
		});
		std::vector<std::thread> pool;
		for (unsigned t(1); t < std::max(1u, threads); ++t) pool.emplace_back(work);
		work();
		for (auto &t : pool) t.join();
		if (error) std::rethrow_exception(error);
	}

	timer.mark("cut search: start states");
	// the walks
	u64 max_chunk_candidates(0);
	for (std::size_t c(0); c < n_chunks; ++c) max_chunk_candidates = std::max<u64>(max_chunk_candidates, chunk_first[c + 1] - chunk_first[c]);
	u64 capacity(std::max<u64>(4096, 192 * max_chunk_candidates));            // (about 100 pairs per candidate on 1KG-like input)
	if (char const *const e = std::getenv("V2M_FOUNDER_TRIAL_CAPACITY")) if (*e) capacity = std::max<u64>(1, std::strtoull(e, nullptr, 10));   // test knob: forces chunks back to the host
	std::vector<u64> trial_end(n_cand, 0);
	std::vector<u32> status(n_chunks, 1);

	// The walker hands every chunk's pairs over in chunk order (the GPU walker: while the next chunks' pairs are still on their way
	// back); here the score updates, in candidate order (find_cut_positions.cc:55-63).  Chunks the walker left undone are walked here.
	std::vector<u32> first_candidate_from_edge;
	walker.chunks_walked = walker.chunks_left = 0;
	walker.walk_streamed(copies, min_distance, cand_edge, cand_aligned, chunk_first, start_order.get(), start_div.get(), capacity, trial_end.data(), status.data(),
		[&](std::size_t c, u32 chunk_status, u32 const *pred, u32 const *cls, u64 n_pairs) {
			++(0 == chunk_status ? walker.chunks_walked : walker.chunks_left);
			if (0 == chunk_status) {
				u64 t(0);
				for (std::size_t j(chunk_first[c]); j < chunk_first[c + 1]; ++j) {
					cut_candidate &current(cuts[j]);
					u64 const t_end(trial_end[j]);
					if (t_end > n_pairs) throw std::runtime_error("founder search: a walked chunk's pair counts exceed what it handed over");
					for (; t < t_end; ++t) current.improve(cls[t], cuts[pred[t]]);
				}
				return;
			}
			if (first_candidate_from_edge.empty()) {
				first_candidate_from_edge.resize(n_edges + 1);
				std::size_t k(0);
				for (u64 e(0); e <= n_edges; ++e) { while (k < cuts.size() && cuts[k].edge < e) ++k; first_candidate_from_edge[e] = u32(k); }
			}
			cut_search_chunk chunk;
			chunk.first = chunk_first[c];
			chunk.end = chunk_first[c + 1];
			scan_cut_search_chunk(graph, min_distance, cuts, first_candidate_from_edge, chunk);
			std::size_t t(0);
			for (std::size_t j(chunk.first); j < chunk.end; ++j)
				for (std::size_t const t_end(chunk.trial_end[j - chunk.first]); t < t_end; ++t) cuts[j].improve(chunk.trials[t].class_count, cuts[chunk.trials[t].pred]);
		});
	timer.mark("cut search: chunk walks + score updates");
	// (the states serve the matching that follows as well: any state at or before a chunk's first cut will do there)
	walker.state_edge.resize(n_chunks);
	for (std::size_t c(0); c < n_chunks; ++c) walker.state_edge[c] = cand_edge[chunk_first[c]];
	walker.state_order = std::move(start_order);
	walker.state_divergence = std::move(start_div);
	walker.state_copies = copies;
	walker.state_edges = graph.edge_count();
	walker.state_matrix = graph.paths_by_chrom_copy_and_edge.words.data();
	return collect_cut_positions(cuts, graph, out);
}

} // namespace


void founder_walker::walk_streamed(u64 n_copies, u64 min_distance, std::vector<u32> const &cand_edge, std::vector<u64> const &cand_aligned,
	std::vector<u64> const &chunk_first, u32 const *start_order, u32 const *start_divergence,
	u64 capacity, u64 *trial_end, u32 *status, chunk_taker const &take)
{
	std::size_t const n_chunks(chunk_first.size() - 1);
	huge_u32_buffer const trial_pred(n_chunks * capacity), trial_class(n_chunks * capacity);
	walk(n_copies, min_distance, cand_edge, cand_aligned, chunk_first, start_order, start_divergence, capacity, trial_pred.get(), trial_class.get(), trial_end, status);
	for (std::size_t c(0); c < n_chunks; ++c) {
		u64 const n((0 == status[c] && chunk_first[c] < chunk_first[c + 1]) ? trial_end[chunk_first[c + 1] - 1] : 0);
		take(c, status[c], trial_pred.get() + c * capacity, trial_class.get() + c * capacity, n);
	}
}


namespace {

class host_founder_walker final : public founder_walker {
public:
	host_founder_walker(variant_graph const &graph, u64 max_copies) : m_graph(graph), m_max_copies(max_copies) {}
	u64 max_copies() const override { return m_max_copies; }

	void walk(u64 n_copies, u64 min_distance, std::vector<u32> const &cand_edge, std::vector<u64> const &cand_aligned,
		std::vector<u64> const &chunk_first, u32 const *start_order, u32 const *start_divergence,
		u64 capacity, u32 *trial_pred, u32 *trial_class, u64 *trial_end, u32 *status) override
	{
		u64 const n_edges(m_graph.edge_count()), words_per_column(m_graph.paths_by_edge_and_chrom_copy.words_per_column());
		std::vector<u32> first_candidate_from_edge(n_edges + 1);
		for (u64 e(0), c(0); e <= n_edges; ++e) { while (c < cand_edge.size() && cand_edge[c] < e) ++c; first_candidate_from_edge[e] = u32(c); }
		for (std::size_t k(0); k + 1 < chunk_first.size(); ++k) {
			status[k] = 0;
			if (chunk_first[k] == chunk_first[k + 1]) continue;
			divergence_counts counts(n_edges + 2);
			edge_pbwt pbwt(u32(n_copies), nullptr);
			std::copy(start_order + k * n_copies, start_order + (k + 1) * n_copies, pbwt.order.begin());
			std::copy(start_divergence + k * n_copies, start_divergence + (k + 1) * n_copies, pbwt.divergence.begin());
			for (u32 const d : pbwt.divergence) counts.add(d, 1);
			pbwt.follow(&counts);
			u64 edge(cand_edge[chunk_first[k]]), n_trials(0);
			u32 *const pred_out(trial_pred + k * capacity), *const class_out(trial_class + k * capacity);
			auto const emit([&](u32 pred, u32 class_count) { if (n_trials < capacity) { pred_out[n_trials] = pred; class_out[n_trials] = class_count; } ++n_trials; });
			for (u64 next(chunk_first[k]); next < chunk_first[k + 1]; ++next) {
				for (; edge < cand_edge[next]; ++edge) pbwt.advance(edge_column(m_graph, edge), words_per_column, edge);   // find_cut_positions.cc:170-176
				u64 right_bound(next + 1);                                                                           // :134-165
				u32 value(counts.largest());
				u32 class_count(counts[value]);
				for (value = counts.below(value); kNoValue != value; value = counts.below(value)) {
					u64 const v(unbiased(value));
					u64 const pred(v <= n_edges ? std::min<u64>(first_candidate_from_edge[v], right_bound) : right_bound);
					if (pred != right_bound) {
						right_bound = pred;
						if (pred != next && min_distance <= cand_aligned[next] - cand_aligned[pred]) emit(u32(pred), class_count);
					}
					class_count += counts[value];
				}
				if (0 != right_bound && right_bound - 1 != next) emit(u32(right_bound - 1), class_count);
				trial_end[next] = n_trials;
			}
			if (n_trials > capacity) status[k] = 1;
		}
	}

	void records(u64 n_copies, std::vector<u32> const &cut_edge, std::vector<u64> const &chunk_first_cut, std::vector<u32> const &start_edge,
		u32 const *start_order, u32 const *start_divergence, u64 pool_capacity, u32 *pool_lhs, u32 *pool_rhs, u32 *pool_size,
		u64 *rec_pool_end, u32 *rec_distinct, u32 *rec_first_class, u32 *rec_first_is_ref, u32 *status) override
	{
		auto const &paths(m_graph.paths_by_edge_and_chrom_copy);
		u64 const words_per_column(paths.words_per_column());
		u32 const copies = u32(n_copies);
		for (std::size_t k(0); k + 1 < chunk_first_cut.size(); ++k) {
			status[k] = 0;
			u64 const cut_begin(chunk_first_cut[k]), cut_end(chunk_first_cut[k + 1]);
			if (cut_begin >= cut_end) continue;
			edge_pbwt pbwt(copies, nullptr);
			std::copy(start_order + k * n_copies, start_order + (k + 1) * n_copies, pbwt.order.begin());
			std::copy(start_divergence + k * n_copies, start_divergence + (k + 1) * n_copies, pbwt.divergence.begin());
			u64 edge(start_edge[k]), n_pool(0);
			u64 const start_cut(cut_begin - 1);
			for (; edge < cut_edge[start_cut]; ++edge) pbwt.advance(edge_column(m_graph, edge), words_per_column, edge);
			std::vector<u32> lhs_class(copies, kPloidyMax), rhs_class(copies, kPloidyMax);
			if (start_cut >= 1) {                                                 // the classes the previous cut left behind
				u32 rep(kPloidyMax);
				for (u32 i(0); i < copies; ++i) {
					if (cut_edge[start_cut - 1] < unbiased(pbwt.divergence[i])) rep = pbwt.order[i];
					rhs_class[pbwt.order[i]] = rep;
				}
			}
			bool first_is_ref(true);
			u32 *const lhs_out(pool_lhs + k * pool_capacity), *const rhs_out(pool_rhs + k * pool_capacity), *const size_out(pool_size + k * pool_capacity);
			for (u64 cut(cut_begin); cut < cut_end; ++cut) {
				for (; edge < cut_edge[cut]; ++edge) {                               // founder_sequence_greedy_output.cc:454-462
					pbwt.advance(edge_column(m_graph, edge), words_per_column, edge);
					first_is_ref = first_is_ref && !paths.test(pbwt.order.front(), edge);
				}
				lhs_class.swap(rhs_class);                                           // :215-251
				u32 rep(kPloidyMax), distinct(0);
				for (u32 i(0); i < copies; ++i) {
					u32 const copy(pbwt.order[i]);
					u64 const d(unbiased(pbwt.divergence[i]));
					if (cut_edge[cut - 1] < d) { rep = copy; ++distinct; }
					rhs_class[copy] = rep;
					if (cut >= 2) {
						if (cut_edge[cut - 2] < d) {
							if (n_pool < pool_capacity) { lhs_out[n_pool] = lhs_class[copy]; rhs_out[n_pool] = rep; size_out[n_pool] = 0; }
							++n_pool;
						}
						if (n_pool && n_pool <= pool_capacity) ++size_out[n_pool - 1];
					}
				}
				rec_pool_end[cut] = n_pool;
				rec_distinct[cut] = distinct;
				rec_first_class[cut] = pbwt.order.front();
				rec_first_is_ref[cut] = first_is_ref ? 1 : 0;
				first_is_ref = true;
			}
			if (n_pool > pool_capacity) status[k] = 1;
		}
	}

private:
	variant_graph const &m_graph;
	u64 m_max_copies;
};

} // namespace


std::unique_ptr<founder_walker> make_host_founder_walker(variant_graph const &graph, u64 max_copies)
{
	return std::unique_ptr<founder_walker>(new host_founder_walker(graph, max_copies));
}


u32 find_cut_positions(variant_graph const &graph, u64 min_distance, std::vector<u64> &out, unsigned threads, founder_walker *walker)
{
	u32 const copies(graph.total_chromosome_copies());
	if (0 == threads) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
	auto const &transposed(graph.paths_by_chrom_copy_and_edge);
	bool const have_transposed(transposed.cols >= copies && transposed.rows >= graph.edge_count() && !transposed.words.empty());
	if (!walker || 0 == copies || copies > walker->max_copies() || 0 == graph.edge_count() || !have_transposed)
		return find_cut_positions(graph, min_distance, out, threads);
	return find_cut_positions_walked(graph, min_distance, out, threads, *walker);
}


u32 find_cut_positions(variant_graph const &graph, u64 min_distance, std::vector<u64> &out, unsigned threads)
{
	u32 const copies(graph.total_chromosome_copies());
	if (0 == threads) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
	auto const &transposed(graph.paths_by_chrom_copy_and_edge);
	bool const have_transposed(transposed.cols >= copies && transposed.rows >= graph.edge_count() && !transposed.words.empty());
	if (threads <= 1 || 0 == copies || 0 == graph.edge_count() || !have_transposed)
		return find_cut_positions_sequential(graph, min_distance, out);
	return find_cut_positions_chunked(graph, min_distance, out, threads);
}


namespace {
	template <typename T> void put_le(std::ostream &os, T v)
	{
		unsigned char b[sizeof(T)];
		for (std::size_t i(0); i < sizeof(T); ++i) b[i] = (unsigned char) (v >> (8 * i));
		os.write(reinterpret_cast<char const *>(b), sizeof(T));
	}

	template <typename T> T get_le(std::istream &is, char const *path)
	{
		unsigned char b[sizeof(T)];
		if (!is.read(reinterpret_cast<char *>(b), sizeof(T))) throw std::runtime_error(std::string(path) + ": truncated cut position file");
		T v(0);
		for (std::size_t i(0); i < sizeof(T); ++i) v |= T(b[i]) << (8 * i);
		return v;
	}
}


void write_cut_positions(cut_position_file const &cuts, char const *path)
{
	std::ofstream os(path, std::ios::binary | std::ios::trunc);
	if (!os) throw std::runtime_error(std::string("unable to open ") + path + " for writing");
	os.put(1);                                                                // archive header: little-endian
	put_le<std::uint32_t>(os, 0);                                             // class version of cut_positions (first use of the type)
	put_le<u64>(os, cuts.min_distance);                                       // serialize(): min_distance, cut_positions, score (output.hh:133-139)
	put_le<u64>(os, cuts.cut_positions.size());
	for (u64 const c : cuts.cut_positions) put_le<u64>(os, c);
	put_le<u32>(os, cuts.score);
	os.flush();
	if (!os) throw std::runtime_error(std::string("error while writing ") + path);
}


cut_position_file read_cut_positions(char const *path)
{
	std::ifstream is(path, std::ios::binary);
	if (!is) throw std::runtime_error(std::string("unable to open ") + path);
	int const marker(is.get());
	if (1 != marker) throw std::runtime_error(std::string(path) + ": not a little-endian portable-binary cut position file");
	if (0 != get_le<std::uint32_t>(is, path)) throw std::runtime_error(std::string(path) + ": unknown cut position file version");
	cut_position_file out;
	out.min_distance = get_le<u64>(is, path);
	u64 const count(get_le<u64>(is, path));
	is.seekg(0, std::ios::end);
	if (count > u64(is.tellg()) / 8) throw std::runtime_error(std::string(path) + ": cut position count exceeds the file size");
	is.seekg(1 + 4 + 8 + 8, std::ios::beg);
	out.cut_positions.resize(count);
	for (u64 &c : out.cut_positions) c = get_le<u64>(is, path);
	out.score = get_le<u32>(is, path);
	return out;
}

} // namespace v2m::host
