"""The shipped configuration: the rest of the GPU suite runs with the size dispatch of the drop-in entry points switched
off (tests/conftest.py sets M4RI_HIP_HOST_SMALL_WORK=0 so that every parity test exercises the HIP path).  Users get the
default -- products and echelon forms of at most 2^20 word operations go to the library's own host routines
(m4ri-rust_amd/csrc/gf2_small_host.cpp), everything larger to the device, also INSIDE composite routines (solve_left =
elimination + products, inverse, rank, the friendly layer's vector products).  This module re-runs the host-ABI parity
tests of the other modules under that default, against the same oracle answers and committed fixtures."""
import pytest

from test_gpu_elim import (block_words, pkg,  # noqa: F401  (fixtures)
                           test_elimination_fixtures, test_inverse, test_inverse_singular, test_rank_and_variants,
                           test_rref_random, test_rref_rank_deficient, test_rref_short_and_wide, test_rref_structured,
                           test_solve_left, test_solve_left_underdetermined_rows_and_inconsistent,
                           test_solve_left_wide_rhs_small_kernel, test_upper_echelon_form)
from test_gpu_parity import (test_addmul_and_prealloc, test_golden_full, test_identity_products_all_strategies,  # noqa: F401
                             test_random_vs_oracle, test_ref_mul_identity, test_ref_vecmul)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def shipped_size_dispatch(monkeypatch):
    monkeypatch.delenv("M4RI_HIP_HOST_SMALL_WORK", raising=False)
    yield


def test_the_default_really_dispatches(pkg):  # noqa: F811
    """Guard for this module's premise: with the variable unset a tiny product is counted by the host routines and a large
    one is not."""
    import numpy as np
    import gf2util as g
    L = pkg._lib.lib()
    before = L.gf2_host_small_calls()
    a, b = g.random_words(10, 10, 1), g.random_words(10, 10, 2)
    c = (pkg.BinMatrix.from_words(a, 10) * pkg.BinMatrix.from_words(b, 10)).to_words()
    assert np.array_equal(c, g.o_mul_naive(a, b, 10, 10, 10)) and L.gf2_host_small_calls() == before + 1
    a, b = g.random_words(3000, 3000, 1), g.random_words(3000, 3000, 2)
    c = (pkg.BinMatrix.from_words(a, 3000) * pkg.BinMatrix.from_words(b, 3000)).to_words()
    assert np.array_equal(c, g.o_mul_m4rm(a, b, 3000, 3000, 3000)) and L.gf2_host_small_calls() == before + 1
