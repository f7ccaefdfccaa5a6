"""One-off fuzzing run (not collected by pytest: file name without test_ prefix is collected only when named)."""
import os
import pytest
from hypothesis import HealthCheck, given, seed, settings

import test_gpu_property as P

pytestmark = pytest.mark.gpu
capi = P.capi


@seed(int(os.environ.get("FUZZ_SEED", "12345")))
@settings(max_examples=int(os.environ.get("FUZZ_N", "600")), deadline=None, derandomize=False, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow, HealthCheck.data_too_large])
@given(case=P.long_cases())
def test_fuzz_long(capi, case):
    P.test_random_cases_match_the_checker.hypothesis.inner_test(capi, case)


@seed(int(os.environ.get("FUZZ_SEED", "12345")) + 1)
@settings(max_examples=int(os.environ.get("FUZZ_N", "600")) * 4, deadline=None, derandomize=False, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow, HealthCheck.data_too_large])
@given(case=P.cases())
def test_fuzz_small(capi, case):
    P.test_random_cases_match_the_checker.hypothesis.inner_test(capi, case)
