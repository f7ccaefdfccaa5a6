"""The reference's docstring examples (tests/golden/reference_doctests.json: statements, expected
repr, citation) hold for pyopal_amd: what src/pyopal/tests/test_doctest.py harvests there.
Examples that search run in the GPU tier."""
import json
import os

import pytest

import pyopal_amd

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_doctests.json")) as f:
    EXAMPLES = json.load(f)["examples"]


def run(example):
    scope = dict(pyopal=pyopal_amd, **{k: getattr(pyopal_amd, k) for k in pyopal_amd.__all__})
    for line in example["lines"][:-1]:
        exec(line, scope)
    assert repr(eval(example["lines"][-1], scope)) == example["expect"], example["cite"]


@pytest.mark.parametrize("example", [e for e in EXAMPLES if not e["gpu"]], ids=lambda e: e["cite"])
def test_docstring_examples(example):
    run(example)


@pytest.mark.gpu
@pytest.mark.parametrize("example", [e for e in EXAMPLES if e["gpu"]], ids=lambda e: e["cite"])
def test_docstring_examples_that_search(example):
    run(example)
