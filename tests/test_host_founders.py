"""The host's cut-position search and greedy matching (vcf2multialign_amd/csrc/host/founder.cc) against the values the
reference's own test pins (tests/founder_sequences.cc:118-188: find_cut_positions(graph, 0), find_matchings(graph, 2))."""

import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
	with open(os.path.join(HERE, "golden", "reference_goldens.json")) as f:
		return json.load(f)["founder_sequences"]


@pytest.fixture(scope="module")
def HostGraph():
	from vcf2multialign_amd import build
	build.build_native()
	from vcf2multialign_amd.host import HostGraph
	return HostGraph


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["vcf"] + "+" + c["fasta"])
def test_cut_positions_and_matchings(HostGraph, case, fixtures_dir):
	d = os.path.join(fixtures_dir, "founder-sequences")
	h = HostGraph(os.path.join(d, case["fasta"]), os.path.join(d, case["vcf"]), case["chromosome"])
	cuts, assigned, score = h.find_founders(case["founder_count"], case["minimum_distance"], keep_ref_edges=False)
	assert cuts == case["cut_positions"]                               # REQUIRE(expected_cut_positions == output.cut_positions())
	assert assigned == case["assigned_samples_column_major"]           # REQUIRE(expected_matchings == output.assigned_samples())
	assert len(assigned) == case["assigned_samples_rows"] * case["founder_count"]


def test_minimum_distance_and_founder_count_are_honoured(HostGraph, fixtures_dir):
	d = os.path.join(fixtures_dir, "founder-sequences")
	h = HostGraph(os.path.join(d, "test-1.fa"), os.path.join(d, "test-1.vcf"), "1")
	cuts0, _, score0 = h.find_founders(2, 0)
	cuts_far, assigned, score_far = h.find_founders(3, 100)            # no two cuts can be 100 aligned positions apart: one block
	assert cuts_far == [0, len(h.reference_positions) - 1]
	assert score_far >= score0
	assert len(assigned) == 3 and len(set(assigned)) == 3             # the three largest path classes of the single block
	for k in (1, 2, 5, 14):
		cuts, assigned, _ = h.find_founders(k, 0)
		assert cuts == cuts0 and len(assigned) == (len(cuts) - 1) * k
