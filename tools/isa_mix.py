#!/usr/bin/env python3
"""Static opcode mix of the hot path of the LK kernels, for the VALU roofline of bench.py.

hipcc -S of velocity_amd/csrc/vh_lk.hip (build container, same flags as the library) -> per kernel the opcode histogram of
  * the Newton-iteration blocks (innermost loop, interior-window path) and
  * the template set-up blocks (interior-window path: staging / V rows / gradients / window sums),
selected by loop depth and size (rules below, per kernel); the border-window and byte-load fallback blocks are left out -- they are not what
the measured configuration executes (> 95 % interior windows at C2).  bench.py weights the two histograms with the LIVE set-up and iteration
counters of the run (x the fitted wave-instructions per set-up / iteration where a PMC fit exists) and prices every opcode with the issue
rate measured by tools/ubench/valu_rate (profiles/rNN_valu_rate.json).

usage: python tools/isa_mix.py r04      -> profiles/r04_lk_isa_mix.json"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from velocity_amd import _build  # noqa: E402

# kernel -> (symbol, rule for iteration blocks, rule for set-up blocks); a rule = (loop depths, min VALU, max VALU)
KERNELS = {
    "k_lk3<51,1,4>": ("_Z5k_lk3ILi51ELi1ELi4EEvPKvm", ((4, 5), 150, 10 ** 6), ((3,), 60, 10 ** 6)),  # (depths count the launch-slot loop of round 5)
    "k_lk_o<15>": ("_Z6k_lk_oILi15EEvPKvm", ((2,), 150, 400), ((1,), 60, 800)),
    "k_lk_q<15>": ("_Z6k_lk_qILi15EEvPKvm", ((2,), 120, 300), ((1,), 60, 450)),
}


def blocks_of(lines, sym):
    start = next(i for i, l in enumerate(lines) if re.match(re.escape(sym) + r"\w*:", l))  # (prefix: the parameter list may grow)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    out, cur = [], dict(name="entry", depth=0, ops=collections.Counter(), byte_loads=0)
    for i in range(start + 1, end):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?\s*(.*)", l)
        if m:
            out.append(cur)
            d = re.search(r"Depth=(\d+)", m.group(2))
            cur = dict(name=m.group(1), depth=int(d.group(1)) if d else 0, ops=collections.Counter(), byte_loads=0)
            continue
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        op = t[0]
        if op.startswith(("flat_load_ubyte", "global_load_ubyte")):
            cur["byte_loads"] += 1
        if "branch" in op:  # a branch ends the basic block: what follows runs only on fall-through
            out.append(cur)
            cur = dict(name=cur["name"].split("+")[0] + "+%d" % i, depth=cur["depth"], ops=collections.Counter(), byte_loads=0)
            continue
        if not op.startswith("v_"):
            continue
        op = re.sub(r"_e32$|_e64$", "", op)
        if ("row_" in l or "quad_perm" in l) and not op.endswith("_dpp"):
            op += "_dpp"
        cur["ops"][op] += 1
    out.append(cur)
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "vh_lk.s")
        flags = [f for f in _build.FLAGS if f != "-fPIC"]
        subprocess.run([_build._hipcc()] + flags + ["--offload-device-only", "-S", "-o", asm, os.path.join(_build.CSRC, "vh_lk.hip")], check=True, capture_output=True)
        lines = open(asm).read().split("\n")
    res = dict(_comment=__doc__.split("usage:")[0].strip(), source_hash=_build.source_hash(), kernels={})
    for kname, (sym, r_it, r_su) in KERNELS.items():
        bl = blocks_of(lines, sym)

        def pick(rule):
            depths, lo, hi = rule
            tot, names = collections.Counter(), []
            for b in bl:
                n = sum(b["ops"].values())
                reflect = b["ops"]["v_min_i32"] + b["ops"]["v_max_i32"]  # REFLECT_101 index arithmetic of the byte-load fallback: not the interior path
                d2 = sum(c for o, c in b["ops"].items() if "dot2" in o)
                # k_lk3's border-window strips (strip_setup<false>: v_mul_lo_u32 is 30 % of the block) and the strips of its err pass (66-79 instructions,
                # 16 dot2; KLTmain discards err, the measured launches never run them) are not the interior path either -- rounds 4 and 5 counted both
                # into the set-up mix (v_mul_lo_u32 was 12 % of it; the class shares barely move: it is a half-rate opcode like the dot products)
                not_interior = kname.startswith("k_lk3") and (b["ops"]["v_mul_lo_u32"] * 5 >= n or (n < 80 and d2 > 0 and b["depth"] == 3))
                if b["depth"] in depths and lo <= n <= hi and b["byte_loads"] <= 2 and reflect * 10 <= n and not not_interior:
                    tot += b["ops"]
                    names.append((b["name"], n))
            return tot, names

        it, it_names = pick(r_it)
        su, su_names = pick(r_su)
        res["kernels"][kname] = dict(symbol=sym, static_valu_total=sum(sum(b["ops"].values()) for b in bl),
                                     iteration=dict(blocks=it_names, static_valu=sum(it.values()), opcodes=dict(it.most_common())),
                                     setup=dict(blocks=su_names, static_valu=sum(su.values()), opcodes=dict(su.most_common())))
        print(kname, "iteration blocks", it_names, "set-up blocks", su_names)
    path = os.path.join(ROOT, "profiles", f"{tag}_lk_isa_mix.json")
    json.dump(res, open(path, "w"), indent=1)
    print(path)


if __name__ == "__main__":
    main()
