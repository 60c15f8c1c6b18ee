"""bench.py's counter legs cannot go stale: every committed rocprofv3 profile (profiles/pmc/<name>.json) names the kernel sources it
was taken on by sha256(kernels.hip + pt_device.h + ptrt_internal.h), and bench.py uses a profile only while workload, kernel and that hash
all match. CPU-only checks of the guard itself, and of the committed profiles being the current sources' (re-profile after touching
the kernels: tools/profile_session.sh + tools/make_profiles.py)."""
import glob
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_committed_counter_profiles_belong_to_the_current_kernels():
    b = _bench()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc", "*.json")))
    assert {os.path.basename(f) for f in files} >= {"tess.json", "cornell.json", "soup.json", "glass.json", "tess4k.json", "sphere.json"}
    for f in files:
        j = json.load(open(f))
        assert j["source_sha256"] == b.source_hash(), f"{os.path.relpath(f, ROOT)} was taken on other kernel sources: re-run the profiling session"
        assert j["per_launch"] and j["hbm_bytes_per_launch"] > 0 and len(j["workload_key"]) == 9


def test_counter_profiles_are_dropped_when_anything_differs(monkeypatch):
    b = _bench()
    j = json.load(open(os.path.join(ROOT, "profiles", "pmc", "tess.json")))
    key, kern = j["workload_key"], j["kernel"].split("<")[0]
    pmc, why = b.load_pmc("tess", key, kern)
    assert pmc is not None and why.startswith("fresh")
    other_spp = list(key); other_spp[4] = 1024          # per-ray figures: the sample count may differ
    assert b.load_pmc("tess", other_spp, kern)[0] is not None
    other_size = list(key); other_size[2] = 3840
    assert b.load_pmc("tess", other_size, kern)[0] is None
    assert b.load_pmc("tess", key, "k_extend_packed")[0] is None
    other_tree = list(key); other_tree[8] += 1            # another builder, another tree: other counters per ray
    assert b.load_pmc("tess", other_tree, kern)[0] is None
    assert b.load_pmc("no_such_profile", key, kern)[0] is None
    monkeypatch.setattr(b, "source_hash", lambda: "0" * 64)  # the kernels changed since the counters were taken
    pmc, why = b.load_pmc("tess", key, kern)
    assert pmc is None and "stale" in why
