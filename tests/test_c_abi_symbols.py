"""The C-ABI library loads and exports every symbol include/pareben_hip.h declares (no compute
calls: there is no GPU here), and argument checking works without a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "pareben_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    dotc = re.findall(r"\bvoid\s+([A-Za-z]+)\s*\(", txt)          # the reference's own .C symbols
    return sorted(set(re.findall(r"\b(pareben_[a-z_]+)\s*\(", txt)) | set(dotc))


def test_header_symbols_exported():
    import pareben_amd
    L = pareben_amd.load_library()
    names = _declared()
    assert {"pareben_ctx_create", "pareben_ctx_run", "pareben_ctx_destroy", "pareben_cv_grid", "pareben_cv_grid_multi",
            "pareben_lambda_max_pairs", "pareben_fit_gaussian", "pareben_fit_gaussian_epis", "pareben_fit_binomial",
            "pareben_fit_binomial_epis", "pareben_ctx_gram", "pareben_version", "pareben_last_error", "pareben_set_trace",
            "elasticNetLinearNeMainEff", "elasticNetLinearNeEpisEff", "ElasticNetBinaryNEmainEff", "ElasticNetBinaryNEfull"} <= set(names)
    for n in names:
        assert hasattr(L, n), n
    assert b"gfx950" in L.pareben_version()
    assert isinstance(L.pareben_device_count(), int)


def test_no_cpu_fallback_and_loud_failure():
    import numpy as np
    import pareben_amd
    L = pareben_amd.load_library()
    if L.pareben_device_count() > 0:
        pytest.skip("GPU present: covered by the gpu tests")
    X = np.asfortranarray(np.random.default_rng(0).standard_normal((20, 5)))
    y = np.zeros(20); fid = (np.arange(20) % 2 + 1).astype(np.int32)
    with pytest.raises(pareben_amd.ParebenError):
        pareben_amd.Context(X, y, fid, 2)                 # no device -> error, never a CPU path
    with pytest.raises(pareben_amd.ParebenError):
        pareben_amd.fit_gaussian(X, y, 0.1, 0.5)
    with pytest.raises(pareben_amd.ParebenError):
        pareben_amd.fit_binomial(X, (y > 0).astype(float), 0.1, 0.5, epis=True)
    with pytest.raises(pareben_amd.ParebenError):
        pareben_amd.cv_grid_multi(X, y, fid, 2, np.array([1.0]), np.array([0.1]), n_gpu=0)     # no device: error, not a host loop
    with pytest.raises(pareben_amd.ParebenError):
        pareben_amd._lib.lambda_max_pairs(X, y)


def test_product_does_not_import_oracle():
    """Nothing under pareben_amd/ may reference oracle/ or the emulation harness."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "pareben_amd")):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle_lib|liboracle|eben_oracle|emul_lib|gm_emul", t):
                    bad.append(f)
    assert not bad, bad
