#!/usr/bin/env python3
"""Symbolises a tools/profile_host.py sample file: per shared object and per function, share of the calling thread's CPU time.
Usage: symbolize_samples.py <file.samples> [top N]"""
import bisect
import collections
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
maps, samples, stacks = [], [], []
for line in open(path):
    if line.startswith("M "):
        p = line[2:].split()
        lo, hi = (int(x, 16) for x in p[0].split("-"))
        maps.append((lo, hi, int(p[2], 16), p[5] if len(p) > 5 else "[anon]", p[1]))
    elif line.startswith("S "):
        fr = [int(x, 16) for x in line[2:].split()]
        samples.append(fr[0])
        stacks.append(fr)
base = {}
for lo, hi, off, name, perm in maps:
    if off == 0 and name not in base:
        base[name] = lo
exec_maps = sorted(m for m in maps if "x" in m[4])
starts = [m[0] for m in exec_maps]
symtabs = {}


def local_path(name):
    if os.path.exists(name):
        return name
    for sub in ("genfer_amd/csrc", "genfer_amd/csrc/host", "oracle", "tools/sampler"):
        c = os.path.join(ROOT, sub, os.path.basename(name))
        if os.path.exists(c):
            return c
    return None


def symtab(name):
    if name in symtabs:
        return symtabs[name]
    tab = []
    lp = local_path(name)
    if lp:
        for args in (["nm", "-n", "-C", "--defined-only", lp], ["nm", "-n", "-C", "-D", "--defined-only", lp]):
            try:
                out = subprocess.run(args, capture_output=True, text=True).stdout
            except OSError:
                out = ""
            for l in out.splitlines():
                q = l.split(None, 2)
                if len(q) == 3 and q[1] in "tTwWiu":
                    tab.append((int(q[0], 16), q[2]))
    tab = sorted(set(tab))
    symtabs[name] = ([a for a, _ in tab], [n for _, n in tab])
    return symtabs[name]


def resolve(pc):
    i = bisect.bisect_right(starts, pc) - 1
    if i < 0 or pc >= exec_maps[i][1]:
        return "?", "?"
    lo, hi, off, name, _ = exec_maps[i]
    vaddr = pc - base.get(name, lo - off)
    addrs, names = symtab(name)
    j = bisect.bisect_right(addrs, vaddr) - 1
    return os.path.basename(name), (names[j] if j >= 0 else "?")[:150]


by_obj, by_fn, by_caller = collections.Counter(), collections.Counter(), collections.Counter()
for fr in stacks:
    short, fn = resolve(fr[0])
    by_obj[short] += 1
    by_fn[(short, fn)] += 1
    if len(fr) > 1 and short.startswith("libc.so"):  # SAMPLE_STACKS=1: who called into libc
        for pc in fr[1:]:
            o, f = resolve(pc - 1)
            if not o.startswith("libc.so") and not o.startswith("libstdc++"):
                by_caller[(fn[:28], o, f[:110])] += 1
                break
n = len(samples)
print(f"{n} samples")
for k, v in by_obj.most_common():
    print(f"  {100.0 * v / n:5.1f} %  {k}")
print()
for (o, f), v in by_fn.most_common(top):
    print(f"  {100.0 * v / n:5.1f} %  {o:18s} {f}")
if by_caller:
    print("\nlibc samples by their first caller outside libc / libstdc++:")
    for (fn, o, f), v in by_caller.most_common(top):
        print(f"  {100.0 * v / n:5.1f} %  {fn:28s} <- {o:16s} {f}")
