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
    for name, rel, flags, cpu_runs in progs:
        assert name not in names
        names.add(name)
        assert os.path.isfile(os.path.join(ROOT, "tests", "golden", "sgcl", rel)), rel
        assert isinstance(flags, str) and isinstance(cpu_runs, int) and cpu_runs >= 0
        if "--bounds" in flags.split():
            assert cpu_runs == 0  # the CPU oracle needs minutes for these
    assert {"hmm", "mixture", "three_populations", "four_populations", "hmm_bounds", "mixture_bounds"} <= names


def test_clock_helper_starts_before_the_gpu_is_touched():
    src = _bench_source()
    main = src[src.index("def main():"):]
    assert main.index("start_clock_helper()") < main.index("import torch"), "the rocm-smi helper must be forked before HIP can be initialised"
    assert "subprocess.Popen" not in src[src.index("def sclk_under_load"):src.index("def main():")], "no fork + exec after GPU initialisation"


def test_contract_keys_are_present():
    src = _bench_source()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"', '"scaling"',
                '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"', '"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"'):
        assert key in src, key
