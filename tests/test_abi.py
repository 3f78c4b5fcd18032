"""CPU: the C-ABI library builds, loads and exports every symbol include/mzr.h declares; the
product path refuses to run without a GPU instead of falling back to a CPU path."""
import os
import re

import pytest

import mizuroute_amd as m

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mzr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mzr_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip_lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(hip_lib, s)]
    assert not missing, missing
    assert sorted(syms) == sorted(m.api.EXPORTS)


def test_no_cpu_fallback(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = m.make_network(20)
    with pytest.raises(m.MzrError) as e:
        m.RoutingDomain(net, 3600.0, [m.KWT], frac_future=[1.0])
    assert e.value.ierr == 90 and "no CPU path" in e.value.message


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "mizuroute_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".f90")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("parity oracle", "").replace("CPU oracle", "").lower() or f == "synthetic.py", (dp, f)
