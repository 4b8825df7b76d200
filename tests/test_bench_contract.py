"""bench.py's static contract, checked without a GPU: the programs its `e2e` rows time exist as committed fixtures, the
clock sampler is a child that is started before anything can have initialised the GPU, and the JSON keys the driver
reads are spelled in the source."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_source():
    with open(os.path.join(ROOT, "bench.py")) as f:
        return f.read()


def test_e2e_programs_are_committed_fixtures():
    tree = ast.parse(_bench_source())
    progs = None
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(getattr(t, "id", None) == "E2E_PROGRAMS" for t in node.targets):
            progs = ast.literal_eval(node.value)
    assert progs, "E2E_PROGRAMS not found"
    names = set()
    for name, rel, flags, cpu_runs, stored, recorded in progs:
        assert name not in names
        names.add(name)
        assert os.path.isfile(os.path.join(ROOT, "tests", "golden", "sgcl", rel)), rel
        assert isinstance(flags, str) and isinstance(cpu_runs, int) and cpu_runs >= 0
        # every row has a checker at the size it is timed at: an oracle run in the same call, or a committed oracle report
        assert cpu_runs > 0 or stored, name
        if stored:
            assert os.path.isfile(os.path.join(ROOT, "tests", "golden", "c3_limit100", stored)), stored
        if cpu_runs == 0:  # no CPU seconds measured in the run: the recorded figure names its source
            assert recorded and recorded[0] > 0 and "profiles/" in recorded[1]
    assert {"hmm", "mixture", "three_populations", "four_populations", "hmm_bounds", "mixture_bounds"} <= names


def test_clock_helper_starts_before_the_gpu_is_touched():
    src = _bench_source()
    main = src[src.index("def main():"):]
    assert main.index("start_clock_helper()") < main.index("import torch"), "the rocm-smi helper must be forked before HIP can be initialised"
    assert "subprocess.Popen" not in src[src.index("def sclk_under_load"):src.index("def main():")], "no fork + exec after GPU initialisation"


def test_nothing_spawns_a_process_after_gpu_initialisation():
    """A process that has initialised HIP must not fork + exec on this pool: the oracle library is built (if missing)
    by ensure_oracle() BEFORE `import torch`; cpu_baseline / e2e_seconds / main's body after that point spawn nothing."""
    src = _bench_source()
    main = src[src.index("def main():"):]
    assert main.index("ensure_oracle()") < main.index("import torch")
    for fn in ("def cpu_baseline(", "def cpu_baseline_all_cores(", "def e2e_seconds("):
        body = src[src.index(fn):]
        body = body[:body.index("\n\n\n")]
        assert "subprocess" not in body and "os.system" not in body and "Popen" not in body, fn
    after = main[main.index("import torch"):]
    assert "subprocess" not in after and "os.system" not in after and "Popen(" not in after


def test_smoke_builds_the_oracle_before_it_initialises_the_gpu():
    """__graft_entry__.smoke(): _oracle_cls() may spawn `make`; it must run before genfer_amd.init() (the same rule)."""
    with open(os.path.join(ROOT, "__graft_entry__.py")) as f:
        src = f.read()
    smoke = src[src.index("def smoke()"):]
    smoke = smoke[:smoke.index("\n\n\n")]
    assert smoke.index("_oracle_cls()") < smoke.index("genfer_amd.init("), "build the oracle before the GPU is initialised"
    after = smoke[smoke.index("genfer_amd.init("):]
    assert "subprocess" not in after and "os.system" not in after and "_oracle_cls" not in after


def test_e2e_rows_carry_a_parity_verdict():
    src = _bench_source()
    body = src[src.index("def e2e_seconds("):src.index("_CLOCK_HELPER")]
    assert "first_difference" in body and '"parity"' in body and "failed = True" in body


def test_contract_keys_are_present():
    src = _bench_source()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"', '"scaling"',
                '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"', '"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"'):
        assert key in src, key
