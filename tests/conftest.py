import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
	if p not in sys.path:
		sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")
FIXTURES = os.path.join(GOLDEN, "reference-fixtures")


def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
	# the in-tree native libraries (HIP product, C++ host, synthetic-input helper) and the CPU oracle; no-ops when fresh
	from vcf2multialign_amd import build
	build.build_native()
	import oracle
	oracle.build_oracle()


@pytest.fixture(scope="session")
def goldens():
	import json
	with open(os.path.join(GOLDEN, "reference_goldens.json")) as f:
		return json.load(f)


@pytest.fixture(scope="session")
def fixtures_dir():
	return FIXTURES
