"""End-to-end snapshot tests: the reference's own `.sgcl` -> `.expect` pairs
(tests/integration.rs protocol: `--no-timing` + the file's `# flags:` line), copied as data fixtures into
tests/golden/sgcl/.  The host interpreter (genfer_amd/csrc/host) is backend-agnostic over the C ABI:

  * oracle backend (CPU, always run): the report must match the `.expect` file BYTE FOR BYTE — this pins
    the oracle on 109 whole programs (moments, probability masses, supports to 16-17 digits);
  * hip backend (-m gpu): the same programs through libgftaylor; primary quantities (Z, E, raw moments,
    probability masses) within 1e-10 relative, derived central/standardised moments within an absolute
    tolerance scaled by the raw moments they are differences of.
"""
import glob
import os
import re

import pytest

from conftest import GOLDEN, ROOT

SGCL = os.path.join(GOLDEN, "sgcl")
FILES = sorted(glob.glob(os.path.join(SGCL, "**", "*.sgcl"), recursive=True))
FILES = [f for f in FILES if os.path.exists(f[:-5] + ".expect")]
SLOW = [f for f in FILES if os.sep + "slow" + os.sep in f]
FAST = [f for f in FILES if f not in SLOW]


def flags_of(path):
    first = open(path).readline()
    return first[len("# flags:"):].strip() if first.startswith("# flags:") else ""


def run(path, backend, prefix):
    import genfer_amd

    rc, text, _ = genfer_amd.run_sgcl_with_backend(open(path).read(), "--no-timing " + flags_of(path), backend, prefix)
    return rc, text


@pytest.fixture(scope="module")
def oracle_path(oracle_lib):
    return os.path.join(ROOT, "oracle", "liborc.so")


def rel(path):
    return os.path.relpath(path, SGCL)


@pytest.mark.parametrize("path", FAST, ids=rel)
def test_snapshot_oracle_byte_exact(path, oracle_path):
    rc, text = run(path, oracle_path, "orc_")
    assert rc == 0, text
    assert text == open(path[:-5] + ".expect").read()


@pytest.mark.parametrize("path", SLOW, ids=rel)
def test_snapshot_oracle_byte_exact_slow(path, oracle_path):
    """All six `slow/` fixtures of the reference (tests/integration.rs runs them behind `--ignored`): mixture,
    nested_infer_expensive and the multi-variable population_* / two_populations2000 programs SURVEY §8c-iii singles
    out.  0.0-2.0 s each on the oracle."""
    rc, text = run(path, oracle_path, "orc_")
    assert rc == 0, text
    assert text == open(path[:-5] + ".expect").read()


def test_example_sgcl_matches_readme_and_closed_form(oracle_path):
    """BASELINE configs[0]: example.sgcl --limit 26 has no .expect; it is pinned by README.md:92-109 and by
    the closed form p(n) = Pois(n;10) * n * 0.2 * 0.8^(n-1) (posterior calls-1 ~ Poisson(8))."""
    from math import exp, factorial

    rc, text = run(os.path.join(SGCL, "example.sgcl"), oracle_path, "orc_")
    assert rc == 0, text
    assert "Total measure:             Z = 0.27067056647322557" in text  # README.md:94
    assert "Unnormalized: p(1)     = 0.00009079985952496972" in text     # README.md:100 (older print layout: "p(1) = ...")
    assert "p(n)     <= 3.1727834072246485e-7 for all n >= 26" in text     # README.md:109
    seen = 0
    for line in text.splitlines():
        m = re.match(r"Unnormalized: p\((\d+)\)\s+= (\S+)", line)
        if m:
            n, p = int(m.group(1)), float(m.group(2))
            want = exp(-10) * 10**n / factorial(n) * n * 0.2 * 0.8 ** (n - 1) if n >= 1 else 0.0
            assert abs(p - want) <= 1e-12 * max(want, 1e-300) or p == want
            seen += 1
    assert seen == 26


from genfer_amd.reports import compare_reports  # noqa: E402  (the parity contract of SURVEY §4; shared with bench.py)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FAST, ids=rel)
def test_snapshot_hip_within_tolerance(path):
    import genfer_amd

    genfer_amd.lib()
    rc, text = run(path, genfer_amd.LIB_PATH, "gft_")
    assert rc == 0, text
    compare_reports(text, open(path[:-5] + ".expect").read())


@pytest.mark.gpu
@pytest.mark.parametrize("path", SLOW, ids=rel)
def test_snapshot_hip_within_tolerance_slow(path):
    import genfer_amd

    genfer_amd.lib()
    rc, text = run(path, genfer_amd.LIB_PATH, "gft_")
    assert rc == 0, text
    compare_reports(text, open(path[:-5] + ".expect").read())


# ---- `--bounds` (BASELINE configs[4]): Interval<F64> tensors through the same interpreter -------------
BOUNDS_PROGRAMS = [
    "example.sgcl",
    "test_expect/sample/poisson.sgcl",
    "test_expect/sample/geometric.sgcl",
    "test_expect/sample/binomial-var.sgcl",
    "test_expect/observe/poisson-var.sgcl",
    "test_expect/observe/negbinomial.sgcl",
    "test_expect/sample/exponential.sgcl",
    "test_expect/real_world/50_2vars.sgcl",
]
BOUNDS_PROGRAMS = [p for p in BOUNDS_PROGRAMS if os.path.exists(os.path.join(SGCL, p))]


def run_flags(path, backend, prefix, flags):
    import genfer_amd

    return genfer_amd.run_sgcl_with_backend(open(path).read(), flags, backend, prefix)[:2]


def intervals_and_points(text):
    out = []
    for line in text.splitlines():
        m = re.search(r"∈ \[(\S+), (\S+)\]", line)
        if m:
            out.append((line.split("∈")[0].strip(), float(m.group(1)), float(m.group(2))))
        else:
            m = re.search(r"= (\S+)$", line)
            if m and ("p(" in line or ":" in line):
                try:
                    v = float(m.group(1))
                    out.append((line.split("=")[0].strip(), v, v))
                except ValueError:
                    pass
    return out


@pytest.mark.parametrize("prog", BOUNDS_PROGRAMS)
def test_bounds_oracle_encloses_point_results(prog, oracle_path):
    """The reference's tests never use --bounds (interval parity is unpinned by it), so the
    interval path is checked for SOUNDNESS: every reported interval must contain the plain-f64 value."""
    path = os.path.join(SGCL, prog)
    flags = "--no-timing --limit 12 " + " ".join(t for t in flags_of(path).split() if t.startswith("--no-probs"))
    rc, pt = run_flags(path, oracle_path, "orc_", flags)
    assert rc == 0, pt
    rc, iv = run_flags(path, oracle_path, "orci_", flags + " --bounds")
    assert rc == 0, iv
    p, i = intervals_and_points(pt), intervals_and_points(iv)
    assert len(p) == len(i) and len(p) > 10
    for (kp, lo_p, hi_p), (ki, lo, hi) in zip(p, i):
        assert kp == ki
        if lo_p != lo_p:  # NaN
            continue
        slack = 1e-9 * max(abs(lo_p), 1.0) if "Normalized" in kp or "p(n)" in kp else 0.0
        assert lo - slack <= lo_p <= hi + slack, (kp, lo, lo_p, hi)


@pytest.mark.gpu
@pytest.mark.parametrize("prog", BOUNDS_PROGRAMS)
def test_bounds_hip_matches_oracle(prog, oracle_path):
    """Interval tensors on the GPU (two planes, same kernels over the interval element functor) vs the
    oracle's Interval<f64>: same op order and widening => bounds agree to 1e-10 (device libm seeds aside)."""
    import genfer_amd

    genfer_amd.lib()
    path = os.path.join(SGCL, prog)
    flags = "--no-timing --bounds --limit 12 " + " ".join(t for t in flags_of(path).split() if t.startswith("--no-probs"))
    rc, want = run_flags(path, oracle_path, "orci_", flags)
    assert rc == 0, want
    rc, got = run_flags(path, genfer_amd.LIB_PATH, "gfti_", flags)
    assert rc == 0, got
    compare_reports(got, want)


# ---- BASELINE configs[2] at its benchmarked size: `--limit 100` (tensors of side ~ 100 + sum of observations) ----------
# The `.expect` snapshots were generated at the programs' auto-limit, so at --limit 100 the checker is the oracle
# itself, run here on the host beside the HIP run of the same program (the shipped configuration: default
# size-threshold dispatch, products above the crossover on the tiled kernel).
C3_F64 = ["approx/hmm/hmm", "approx/two_populations/two_populations", "approx/mixture/mixture", "approx/switchpoint/switchpoint",
          "approx/population/population", "approx/population_modified/population_modified", "exact/alarm/alarm",
          "exact/clickGraph/clickGraph", "exact/clinicalTrial2/clinicalTrial2", "exact/evidence1/evidence1", "exact/evidence2/evidence2",
          "exact/grass/grass", "exact/murderMystery/murderMystery", "exact/twoCoins/twoCoins"]  # the whole suite
C3_BOUNDS = ["approx/hmm/hmm", "approx/two_populations/two_populations", "approx/mixture/mixture", "approx/switchpoint/switchpoint",
             "approx/population/population"]
C3_LIMIT100 = [(p, False) for p in C3_F64] + [(p, True) for p in C3_BOUNDS]
# oracle runs too long to repeat inside the GPU test run are committed by tests/golden/make_c3_limit100_golden.py
# (mixture --bounds: ~12 minutes of one host core)
C3_STORED = os.path.join(GOLDEN, "c3_limit100")


def c3_stored(prog, bounds):
    return os.path.join(C3_STORED, prog.split("/")[-1] + ("-bounds" if bounds else "") + ".oracle.txt")


def test_oracle_c3_stored_reports_are_wellformed():
    """The committed oracle reports (generated by make_c3_limit100_golden.py) have the benchmarked size."""
    files = glob.glob(os.path.join(C3_STORED, "*.oracle.txt"))
    assert files
    for f in files:
        text = open(f).read()
        # (--limit 100 for the NeurIPS programs and three_populations; four_populations is timed at --limit 24)
        assert text.count("p(") >= (24 if "four_populations" in f else 100) and "Total measure" in text


@pytest.mark.gpu
@pytest.mark.parametrize("prog,bounds", C3_LIMIT100, ids=[p.split("/")[-1] + ("-bounds" if b else "") for p, b in C3_LIMIT100])
def test_c3_limit100_hip_matches_oracle(prog, bounds, oracle_path):
    import conftest
    import genfer_amd

    genfer_amd.lib()
    conftest._set_tier("host")  # the library's default dispatch
    path = os.path.join(SGCL, "neurips2023", prog + ".sgcl")
    flags = "--no-timing --limit 100" + (" --bounds" if bounds else "")
    try:
        rc, got = run_flags(path, genfer_amd.LIB_PATH, "gfti_" if bounds else "gft_", flags)
        assert rc == 0, got
    finally:
        conftest._set_tier("device")
    if os.path.exists(c3_stored(prog, bounds)):
        want = open(c3_stored(prog, bounds)).read()
    else:
        rc, want = run_flags(path, oracle_path, "orci_" if bounds else "orc_", flags)
        assert rc == 0, want
    assert "Total measure" in want and (want.count("p(") >= 100 or "clickGraph" in prog)  # clickGraph's result is continuous
    compare_reports(got, want)


# ---- this repo's product-dominated programs (bench.py's e2e rows three_populations / four_populations) ----------------
# Every statement couples two variables, so `eval` runs general Horner loops whose steps are rank-3 / rank-4 general
# products (SURVEY §8 A3 / A7 on the path the headline kernel serves).  Checked against the oracle at limits the oracle
# finishes in about a second.
PRODUCT_PROGRAMS = [("bench/three_populations", "--limit 40"), ("bench/four_populations", "--limit 12")]


@pytest.mark.gpu
@pytest.mark.parametrize("prog,limit", PRODUCT_PROGRAMS, ids=[p.split("/")[-1] for p, _ in PRODUCT_PROGRAMS])
def test_product_programs_hip_matches_oracle(prog, limit, oracle_path):
    import genfer_amd

    genfer_amd.lib()
    path = os.path.join(SGCL, prog + ".sgcl")
    rc, got = run_flags(path, genfer_amd.LIB_PATH, "gft_", "--no-timing " + limit)
    assert rc == 0, got
    rc, want = run_flags(path, oracle_path, "orc_", "--no-timing " + limit)
    assert rc == 0, want
    assert "Total measure" in want
    compare_reports(got, want)


# ... and at exactly the sizes bench.py's e2e rows time them, against the committed oracle reports
# (tests/golden/make_c3_limit100_golden.py: 6-30 s of one host core each)
PRODUCT_PROGRAMS_BENCH = [("bench/three_populations", "--limit 100"), ("bench/four_populations", "--limit 24")]


@pytest.mark.gpu
@pytest.mark.parametrize("prog,limit", PRODUCT_PROGRAMS_BENCH, ids=[p.split("/")[-1] + "@bench" for p, _ in PRODUCT_PROGRAMS_BENCH])
def test_product_programs_at_bench_size_match_committed_oracle(prog, limit):
    import conftest
    import genfer_amd

    genfer_amd.lib()
    stored = c3_stored(prog, False)
    assert os.path.exists(stored), stored
    conftest._set_tier("host")  # the library's default dispatch — the configuration bench.py times
    try:
        rc, got = run_flags(os.path.join(SGCL, prog + ".sgcl"), genfer_amd.LIB_PATH, "gft_", "--no-timing " + limit)
        assert rc == 0, got
    finally:
        conftest._set_tier("device")
    want = open(stored).read()
    assert "Total measure" in want and want.count("p(") >= int(limit.split()[-1])
    compare_reports(got, want)


@pytest.mark.parametrize("prog,limit", [("bench/three_populations", "--limit 12"), ("bench/four_populations", "--limit 6")],
                         ids=["three_populations", "four_populations"])
def test_product_programs_run_on_the_oracle(prog, limit, oracle_path):
    """The programs parse and evaluate (CPU): Z is a probability, the masses are printed."""
    rc, text = run_flags(os.path.join(SGCL, prog + ".sgcl"), oracle_path, "orc_", "--no-timing " + limit)
    assert rc == 0, text
    z = float(re.search(r"Z = (\S+)", text).group(1))
    assert 0.0 < z < 1.0 and text.count("p(") >= 2 * int(limit.split()[-1])
