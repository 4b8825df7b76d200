"""A16 (`impl Display / Debug for TaylorPoly`, fmt_polynomial, src/multivariate_taylor.rs:632-636,694-730) and §8f-4 (the
reference's command line: `--print-gf`, `--json`, the executable its harness spawns, src/main.rs:22-131,595-645 and
benchmarks/neurips2023/exact/bench.py:44-105).

The expected strings are written out by hand from the reference's formatting rules: coefficients in row-major order,
zeros skipped, `Display for F64` = ryu (f64.rs:41-45), variables a, b, ... (ppl.rs:107-117), "^e" above 1, " + "
between terms, "0" for the zero polynomial; GF nodes per generating_function.rs:330-432 with the precedences of
:451-470."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

GENFER = os.path.join(ROOT, "genfer_amd", "csrc", "host", "genfer")
ORACLE = os.path.join(ROOT, "oracle", "liborc.so")
SGCL = os.path.join(GOLDEN, "sgcl")

# Debug = "TaylorPoly({:?}, {})" of degrees_p1 and of the coefficient ndarray (mt:632-636): ndarray 0.15.6's Display —
# nested brackets with EVERY stored element through `Display for F64`, rows separated by ",\n" + (ndim-2) blank lines +
# one space per depth, usize::MAX degrees in full.  Literal strings, written out from arrayformat.rs' rules.
UMAX = 18446744073709551615
DISPLAY_CASES = [
    (np.array([[1.0, 2.0, 0.0], [3.0, 0.0, 4.5e-7]]), [4, 5], "1.0 + 2.0b + 3.0a + 4.5e-7ab^2",
     "TaylorPoly([4, 5], [[1.0, 2.0, 0.0],\n [3.0, 0.0, 4.5e-7]])"),
    (np.zeros((2, 2)), [2, 2], "0", "TaylorPoly([2, 2], [[0.0, 0.0],\n [0.0, 0.0]])"),
    (np.array([0.1 + 0.2, 1e22, -1e-7, 123456789012345680.0]), [UMAX], "0.30000000000000004 + 1e22a + -1e-7a^2 + 1.2345678901234568e17a^3",
     "TaylorPoly([18446744073709551615], [0.30000000000000004, 1e22, -1e-7, 1.2345678901234568e17])"),
    (np.array([[[0.0, 0.5], [0.0, 0.0]], [[0.0, 0.0], [0.0, -2.0]]]), [2, 2, 2], "0.5c + -2.0abc",
     "TaylorPoly([2, 2, 2], [[[0.0, 0.5],\n  [0.0, 0.0]],\n\n [[0.0, 0.0],\n  [0.0, -2.0]]])"),
]


def _check_debug_large(T):
    """From 500 elements on ndarray abbreviates: 11 items on the last two axes, 6 on the others, half from each end."""
    p = T.new(np.arange(600, dtype=np.float64), [600])
    want = "[" + ", ".join(f"{i}.0" for i in range(5)) + ", ..., " + ", ".join(f"{i}.0" for i in range(595, 600)) + "]"
    assert repr(p) == f"TaylorPoly([600], {want})"
    a = np.arange(8 * 12 * 13, dtype=np.float64).reshape(8, 12, 13)
    p = T.new(a, [8, 12, 13])

    def row(r):
        return "[" + ", ".join(f"{v:.1f}" for v in r[:5]) + ", ..., " + ", ".join(f"{v:.1f}" for v in r[-5:]) + "]"

    def plane(m):
        rows = [row(r) for r in m[:5]] + ["..."] + [row(r) for r in m[-5:]]
        return "[" + ",\n  ".join(rows) + "]"

    planes = [plane(m) for m in a[:3]] + ["..."] + [plane(m) for m in a[-3:]]
    assert repr(p) == "TaylorPoly([8, 12, 13], [" + ",\n\n ".join(planes) + "])"


def _check_display(T):
    for arr, deg, disp, dbg in DISPLAY_CASES:
        p = T.new(arr, deg)
        assert str(p) == disp
        if dbg:
            assert repr(p) == dbg
    assert str(T.from_scalar(0.0)) == "0" and str(T.from_scalar(2.5)) == "2.5"
    assert repr(T.from_scalar(2.5)) == "TaylorPoly([], 2.5)"  # 0-dimensional array: the element itself
    _check_debug_large(T)
    assert str(T.var(27, 0.0, 3)) == "1.0x_27"  # ppl.rs:113: variables beyond z


def test_display_oracle(OTP, OTPI):
    _check_display(OTP)
    iv = OTPI.new(np.stack([np.array([0.0, 1.5]), np.array([0.0, 1.75])]), [3])
    assert str(iv) == "[1.5, 1.75]a"  # interval.rs:243-247
    assert repr(iv) == "TaylorPoly([3], [[0.0, 0.0], [1.5, 1.75]])"


@pytest.mark.gpu
def test_display_hip(GTP, GTPI, OTP):
    _check_display(GTP)
    iv = GTPI.new(np.stack([np.array([0.0, 1.5]), np.array([0.0, 1.75])]), [3])
    assert str(iv) == "[1.5, 1.75]a" and repr(iv) == "TaylorPoly([3], [[0.0, 0.0], [1.5, 1.75]])"
    rng = np.random.default_rng(5)
    for shape in [(7,), (3, 4), (2, 3, 2), (40, 60)]:  # the last one lives on the device under the default dispatch
        a = rng.standard_normal(shape) * 10.0 ** rng.integers(-12, 12, size=shape)
        a[rng.random(shape) < 0.3] = 0.0
        assert str(GTP.new(a, list(shape))) == str(OTP.new(a, list(shape)))


def _run(args, backend="oracle"):
    env = dict(os.environ)
    if backend == "oracle":
        env["GENFER_BACKEND"] = ORACLE + ":orc"
    else:
        env.pop("GENFER_BACKEND", None)
    return subprocess.run([GENFER] + args, capture_output=True, text=True, env=env, timeout=600)


def _ensure_cli():
    if not os.path.exists(GENFER):
        subprocess.check_call(["make", "-C", os.path.dirname(GENFER)])


EXAMPLE_GF = ("coeff_at_zero([a -> a * (0.2 * b + 0.8) in [b -> 1.0 in exp(10.0 * (a + -1.0)) * [a -> 1.0 in 1.0]]] of b^1) * b^1 + 0.0")


def test_cli_reports_match_expect_files(oracle_lib):
    """`genfer --no-timing <flags> file` (tests/integration.rs protocol) prints the `.expect` file byte for byte."""
    _ensure_cli()
    for rel in ("test_expect/sample/poisson.sgcl", "test_expect/observe/negbinomial.sgcl", "neurips2023/exact/alarm/alarm.sgcl"):
        path = os.path.join(SGCL, rel)
        first = open(path).readline()
        flags = first[len("# flags:"):].split() if first.startswith("# flags:") else []
        r = _run(["--no-timing"] + flags + [path])
        assert r.returncode == 0, r.stderr
        assert r.stdout == open(path[:-5] + ".expect").read()


def test_cli_print_gf_and_json(oracle_lib, tmp_path):
    _ensure_cli()
    js = tmp_path / "out.json"
    r = _run(["--limit", "5", "--print-gf", "--json", str(js), os.path.join(SGCL, "example.sgcl")])
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("Generating function:\n" + EXAMPLE_GF + "\n\nRemaining mass:\nmax(0.0, 0.0)\n\nTime to construct the generating function: ")
    assert re.search(r"^Total inference time: [0-9.]+s$", r.stdout, re.M)  # what exact/bench.py:36 parses
    text = js.read_text()
    # the reference's literal layout (main.rs:617-633): leading newline, 4-space indent, trailing commas
    assert text.startswith('\n{\n    "model": "example",\n    "system": "genfer",\n    "time_gf_translation": ')
    assert text.endswith(",\n}\n")
    data = json.loads(re.sub(r",(\s*[\]}])", r"\1", text))
    assert data["total"] == 0.27067056647322557 and data["mean"] == 9.0 and len(data["masses"]) == 5
    assert data["masses"][1] == 0.00009079985952496972
    for k in ("time_gf_translation", "time_moments", "time_probs", "time_infer"):
        assert isinstance(data[k], float) and 0 < data[k] < 60 and "e" not in re.search(rf'"{k}": ([^,]+),', text).group(1)


def test_cli_errors_like_the_reference(oracle_lib):
    _ensure_cli()
    r = _run(["--rational", os.path.join(SGCL, "example.sgcl")])
    assert r.returncode != 0 and "out of scope" in r.stderr
    r = _run([os.path.join(SGCL, "does_not_exist.sgcl")])
    assert r.returncode != 0


@pytest.mark.gpu
def test_cli_on_the_gpu_backend():
    """The executable with its default backend (libgftaylor next to it): the report of example.sgcl within 1e-10."""
    from test_e2e_snapshots import compare_reports

    _ensure_cli()
    path = os.path.join(SGCL, "example.sgcl")
    got = _run(["--no-timing", "--limit", "26", path], backend="hip")
    assert got.returncode == 0, got.stderr
    want = _run(["--no-timing", "--limit", "26", path], backend="oracle")
    compare_reports(got.stdout, want.stdout)
    bounds = _run(["--no-timing", "--limit", "12", "--bounds", path], backend="hip")
    assert bounds.returncode == 0 and "∈ [" in bounds.stdout
