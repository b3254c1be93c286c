import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle as orc

    orc.build()
    orc.api()
    return orc


def assert_image_close(img_d, img_o, spp, frac=2e-4, rel=1e-4, what=""):
    """radiance: `rel` of the image scale per pixel (fp32 reassociation of the sample sum, ulp-level libm
    differences) for all but `frac` of the pixels; and NO pixel may be off by more than what one or two
    samples whose hit or shadow test flips at an edge can cause: a sample carries 1/spp of its pixel, and
    a single sample is at most ~1.5x the brightest pixel mean, so the bound is 1.5 * scale / spp (it was a
    flat 0.2 * scale: at 64 spp this is 8x tighter)."""
    scale = float(img_o.max())
    assert scale > 0, what
    err = np.abs(img_d.astype(np.float64) - img_o.astype(np.float64))
    bad = float((err > rel * scale).mean())
    assert bad <= frac, f"{what}: {bad:.2e} of the pixel channels differ by more than {rel:g} of the scale"
    assert err.max() <= 1.5 * scale / spp, f"{what}: worst pixel off by {err.max() / scale:.3f} of the scale (bound {1.5 / spp:.3f})"
    return scale, err
