"""One-off end-to-end check of `--bounds`: the report text of the HIP backend against the CPU oracle (timing lines
aside) on NeurIPS'23 programs at limits the oracle finishes in minutes.  Run on the GPU box from the repo root."""
import sys, os
sys.path.insert(0, os.getcwd())
import genfer_amd
genfer_amd.init(0)
oracle = os.path.join(os.getcwd(), "oracle", "liborc.so")
for prog, lim in (("mixture", 30), ("switchpoint", 40), ("population", 100), ("hmm", 40)):
    src = open(f"tests/golden/sgcl/neurips2023/approx/{prog}/{prog}.sgcl").read()
    flags = f"--limit {lim} --bounds"
    rc1, t1, _ = genfer_amd.run_sgcl_with_backend(src, flags, genfer_amd.LIB_PATH, "gfti_")
    rc2, t2, _ = genfer_amd.run_sgcl_with_backend(src, flags, oracle, "orci_")
    strip = lambda t: "\n".join(l for l in t.splitlines() if "time" not in l.lower())
    print(prog, lim, rc1, rc2, "IDENTICAL" if strip(t1) == strip(t2) else "DIFFERENT", len(t1))
    if strip(t1) != strip(t2):
        a, b = strip(t1).splitlines(), strip(t2).splitlines()
        for x, y in zip(a, b):
            if x != y:
                print("  gpu:", x[:150]); print("  cpu:", y[:150]); break
