"""Shared test helpers: the parity tolerance of the path and the comparison that applies it."""

import numpy as np

# Parity tolerance of the path (BASELINE.json north_star: 1e-4 relative fp32; SURVEY.md §8c:
# pointwise-relative is ill-posed where LayerNorm outputs cross zero, hence the atol term).
RTOL, ATOL = 1e-4, 1e-5
REL_L2 = 1e-5


def assert_close(got, ref, what="", rtol=RTOL, atol=ATOL, rel_l2=REL_L2):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} != {ref.shape}"
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    err = np.abs(got - ref)
    bound = atol + rtol * np.abs(ref)
    worst = int(np.argmax(err - bound))
    assert (err <= bound).all(), (f"{what}: max abs err {err.max():.3e}; worst idx {np.unravel_index(worst, got.shape)} "
                                  f"got {got.flat[worst]:.7g} ref {ref.flat[worst]:.7g}")
    l2 = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
    assert l2 <= rel_l2, f"{what}: relative L2 error {l2:.3e} > {rel_l2:.1e}"
