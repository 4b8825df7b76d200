"""Host-interpreter invariants that the GPU-side numbers rest on, checked on the CPU over the oracle backend.

The interpreter keeps the observation chains of the Poisson recognisers (generating_function.rs:684-706) for the length of
one top-level eval() instead of rebuilding them at every input point (gfh_genfun.hpp ChainTable).  That must not change
WHICH TaylorPoly operations are issued: the report and the per-(operation, size bucket) histogram of a run are the same
with the table on and off (GFH_CHAIN_TABLE=0 = the reference's rebuild-every-time form)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "genfer_amd", "csrc", "host", "genfer")
ORACLE = os.path.join(ROOT, "oracle", "liborc.so")
SGCL = os.path.join(ROOT, "tests", "golden", "sgcl")

CASES = [  # (sized for seconds on the CPU oracle; the full-size programs were compared the same way by hand: profiles/r05/ab_interpreter_chain_table.txt)
    ("neurips2023/approx/population/population.sgcl", ""),
    ("neurips2023/approx/two_populations/two_populations.sgcl", "--limit 40"),
    ("neurips2023/approx/switchpoint/switchpoint.sgcl", "--limit 30"),
    ("neurips2023/approx/hmm/hmm.sgcl", "--limit 40"),
    ("neurips2023/approx/hmm/hmm.sgcl", "--limit 16 --bounds"),
]


def run(path, flags, table):
    env = dict(os.environ, GENFER_BACKEND=ORACLE + ":orc", GFH_TRACE_SIZES="1", GFH_CHAIN_TABLE="1" if table else "0")
    r = subprocess.run([CLI] + flags.split() + [os.path.join(SGCL, path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    report = "\n".join(l for l in r.stdout.splitlines() if "ime" not in l)  # timing lines differ run to run
    hist = sorted(l for l in r.stderr.splitlines() if l.startswith("[gfh sizes]"))
    return report, hist


@pytest.mark.parametrize("path,flags", CASES, ids=[c[0].split("/")[-1] + (" " + c[1] if c[1] else "") for c in CASES])
def test_kept_observation_chains_issue_the_same_operations(path, flags):
    if not (os.path.exists(CLI) and os.path.exists(ORACLE)):
        pytest.skip("genfer / liborc.so not built")
    rep_on, hist_on = run(path, flags, True)
    rep_off, hist_off = run(path, flags, False)
    assert hist_on, "GFH_TRACE_SIZES printed nothing"
    assert rep_on == rep_off
    assert hist_on == hist_off
