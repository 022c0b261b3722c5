"""Randomised unstructured meshes (Delaunay of random points; 2D / 3D, linear / quadratic, constant isotropic / orthotropic field / isotropic
field): K against the oracle where it is small, the matrix-free operator against the assembled K to rounding, block-Jacobi, two-level and
multigrid PCG to the same displacements. 350 seeds ran clean in round 5 (scripts/fuzz_unstructured.py); six of them are kept here."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [3, 9, 10, 14, 25, 31])
def test_random_unstructured_mesh(seed):
    from fuzz_unstructured_util import check
    ok, line = check(seed)
    assert ok, line
