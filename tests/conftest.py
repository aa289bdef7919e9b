"""pytest configuration: path setup, the `gpu` marker, golden-fixture loader."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "junction-tree_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The C-ABI library is required by most tests (planner checks run it in plan-only mode on
    # CPU): build it with hipcc if it is missing or older than its sources (~30 s, cross-compiles
    # without a GPU).  On the GPU box the prebuilt file travels with the snapshot.
    sys.path.insert(0, os.path.join(ROOT, "junction-tree_amd"))
    try:
        import build as jt_build
        jt_build.build(force=False, verbose=False)
    except Exception as exc:                         # noqa: BLE001 - report, let the tests fail loudly
        print("WARNING: could not build libjtprop.so: %r" % (exc,), file=sys.stderr)


class Golden:
    """A fixture bundle written by oracle/gen_golden.py: JSON `meta` + numbered arrays."""

    def __init__(self, name):
        data = np.load(os.path.join(GOLDEN, name))
        self.meta = json.loads(str(data["meta"]))
        self._data = data

    def arr(self, key):
        return self._data[key]

    def arrs(self, keys):
        return [self._data[k] for k in keys]


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load


def as_tree(obj):
    """JSON tree ([c, [sep, subtree], ...]) -> the reference's list/tuple form."""
    return [obj[0]] + [(e[0], as_tree(e[1])) for e in obj[1:]]
