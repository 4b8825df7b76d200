"""Comparing two Genfer reports (the text `genfer file.sgcl` prints, src/main.rs) under the parity contract of
SURVEY §4: identical text once the numbers are masked; primary quantities (Z, E, raw moments, probability masses)
within 1e-10 relative; derived central / standardised moments and tail bounds within an absolute tolerance scaled by
the raw moments they are differences of.  Used by tests/test_e2e_snapshots.py and by bench.py's e2e rows (which compare
the timed GPU report with the oracle report of the same program at the same size)."""
import re

NUM = re.compile(r"[-+]?(?:\d+\.\d+(?:e-?\d+)?|\d+e-?\d+|inf|NaN)")


def numbers(line):
    return [float(x) for x in NUM.findall(line)]


def compare_reports(got, want):
    """Raises AssertionError naming the first difference."""
    gl, wl = got.splitlines(), want.splitlines()
    assert len(gl) == len(wl), "different number of report lines"
    raw = {}
    for g, w in zip(gl, wl):
        gs, ws = NUM.sub("#", g), NUM.sub("#", w)
        assert gs == ws, f"report text differs:\n{g}\n{w}"
        gn, wn = numbers(g), numbers(w)
        if not wn:
            continue
        primary = any(k in w for k in ("Total measure", "Expected value", "raw moment", "p(")) and "<=" not in w
        if "Expected value" in w:
            raw["E"] = abs(wn[-1])
        if "4th raw moment" in w:
            raw["m4"] = abs(wn[-1])
        for a, b in zip(gn, wn):
            if a != a and b != b:  # NaN in both reports (e.g. skewness of a point mass)
                continue
            if primary:
                assert abs(a - b) <= 1e-10 * abs(b) or a == b or abs(b) < 1e-300, f"{g} vs {w}"
            else:
                # central / standardised moments and tail bounds: differences of raw moments
                scale = max(abs(b), raw.get("m4", 1.0), 1.0)
                assert abs(a - b) <= 1e-9 * scale or a == b, f"{g} vs {w}"


def first_difference(got, want):
    """None if the reports agree under the contract, else a one-line description of the first difference."""
    try:
        compare_reports(got, want)
    except AssertionError as e:
        return " / ".join(str(e).splitlines())[:300]
    return None


_TIMING = re.compile(r"^(Time to |Total inference time|Total time)")


def strip_timing(text):
    """The report without its wall-clock lines (what `--no-timing` prints)."""
    return "\n".join(l for l in text.splitlines() if not _TIMING.match(l.strip())) + "\n"
