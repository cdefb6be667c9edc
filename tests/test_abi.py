"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/ma_amd.h declares (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

from ma_testlib import ROOT


def declared_functions():
    src = open(os.path.join(ROOT, "include", "ma_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(ma_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    import ma_amd
    assert os.path.exists(ma_amd.lib_path()), "libma_amd.so not built: run __graft_entry__.build()"
    L = C.CDLL(ma_amd.lib_path())
    names = declared_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, "symbols declared in include/ma_amd.h but not exported: %s" % missing


def test_abi_version_and_params():
    import ma_amd
    L = ma_amd.lib()
    assert L.ma_abi_version() == 1
    p = ma_amd.Params.preset("default")
    assert (p.seeding_technique, p.min_seed_len, p.max_ambiguity, p.max_num_soc, p.min_num_soc) == (0, 16, 100, 30, 1)
    assert (p.match, p.mismatch, p.gap, p.extend, p.gap2, p.extend2) == (2, 4, 4, 2, 24, 1)
    assert (p.padding, p.bandwidth_ext, p.zdrop, p.max_gap_area) == (1000, 512, 200, 20)
    q = ma_amd.Params.preset("illumina")
    assert (q.seeding_technique, q.max_ambiguity, q.min_num_soc, q.max_num_soc) == (1, 500, 10, 20)


def test_params_layout_matches_oracle_struct():
    import ma_amd
    from ma_testlib import OrParams
    assert C.sizeof(ma_amd.Params) == C.sizeof(OrParams)
    # same fields in the same order; the product's diagnostics knob libm_probe sits where the oracle's block has padding
    assert [f[0].replace("libm_probe", "pad_") for f in ma_amd.Params._fields_] == [f[0] for f in OrParams._fields_]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import ma_amd.api as api
    monkeypatch.setattr(api, "_lib", None)
    monkeypatch.setattr(api, "lib_path", lambda: str(tmp_path / "nope.so"))
    try:
        api.lib()
    except api.MaError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("expected MaError")
    monkeypatch.setattr(api, "_lib", None)
