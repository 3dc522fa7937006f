#!/usr/bin/env python3
"""LDS bank-conflict profile of the kernels on the CPU model (TEST / DEVELOPMENT TOOL; nothing of the product uses it).

    python tools/hipemu/ldsprof.py [--tag T --extra "-DDALIAMD_X=1 ..."] [--top N] -- <pytest arguments>

runs the given gpu-marked tests on the racecheck build of the model (make RACE=1: every memory access of the kernel
sources calls a hook) with HIPEMU_LDSPROF set: the runtime groups the LDS accesses of a wave's lanes into wave
instructions (same code address, same visit count since the wave's last wave-wide event) and prices each after the
table in /opt/skills/guides/MI355X_MICROARCH.md, section LDS (lane groups per instruction width, 32 or 64 banks,
broadcast of equal dwords).  Output per kernel: wave instructions, conflict-free LDS cycles, extra (conflict) cycles - the
model's counterpart of SQ_LDS_BANK_CONFLICT against SQ_LDS_IDX_ACTIVE - and the source lines that pay most.

What it is not: the access widths are what clang makes of the source for x86-64, not the gfx950 instruction stream
(neighbouring dword accesses the device merges into ds_read2 / b64 stay separate here); the ratio between two LAYOUTS of
the same code is what the tool is for.  Validation against the hardware counters: profiles/r04_ldsprof_model.md.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"


def symbolize(lib, offsets):
    """offset -> (function, file:line of the innermost frame, file:line of the outermost frame)"""
    offsets = sorted(set(offsets))
    out = subprocess.run([SYMBOLIZER, "-e", lib, "--inlines", "--functions=short"] + [hex(o) for o in offsets],
                         capture_output=True, text=True).stdout.strip("\n").split("\n\n")
    res = {}
    for off, blk in zip(offsets, out):
        lines = blk.strip().splitlines()
        frames = [(lines[i], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]
        if not frames:
            res[off] = ("?", "?", "?")
            continue
        short = lambda loc: re.sub(r":\d+$", "", loc.replace(ROOT + "/", "").replace("tools/hipemu/_build", "_build"))  # noqa: E731
        res[off] = (frames[-1][0], short(frames[0][1]), short(frames[-1][1]))
    return res


def collect(path, lib):
    rows = []
    for ln in open(path):
        k, pc, size, rw, instr, base, conflict, lanes = ln.split()
        rows.append((int(k, 16), int(pc, 16), int(size), rw, int(instr), int(base), int(conflict), int(lanes)))
    sym = symbolize(lib, [r[0] for r in rows] + [r[1] for r in rows])
    kernels = collections.defaultdict(lambda: {"instr": 0, "base": 0, "conflict": 0, "sites": collections.defaultdict(lambda: [0, 0, 0, 0])})
    for k, pc, size, rw, instr, base, conflict, lanes in rows:
        name = sym[k][0]
        K = kernels[name]
        K["instr"] += instr
        K["base"] += base
        K["conflict"] += conflict
        site = K["sites"][(sym[pc][1], size, rw)]
        site[0] += instr
        site[1] += base
        site[2] += conflict
        site[3] += lanes
    return kernels


def source_line(loc):
    m = re.match(r"(.*):(\d+)$", loc)
    if not m:
        return ""
    path = m.group(1)
    for cand in (os.path.join(ROOT, path), os.path.join(ROOT, "tools", "hipemu", path.replace("_build", "_build", 1)), path):
        if os.path.exists(cand):
            lines = open(cand).read().splitlines()
            n = int(m.group(2))
            return lines[n - 1].strip()[:110] if 0 < n <= len(lines) else ""
    return ""


def main():
    ap = argparse.ArgumentParser(usage=__doc__)
    ap.add_argument("--tag", default="")
    ap.add_argument("--extra", default="")
    ap.add_argument("--top", type=int, default=8)
    ap.add_argument("--kernel", default="", help="only kernels whose name contains this")
    ap.add_argument("pytest_args", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    pytest_args = [a for a in args.pytest_args if a != "--"]
    env = dict(os.environ, DALI_AMD_HIPEMU="race", HIPEMU_TAG=args.tag, HIPEMU_EXTRA=args.extra)
    env.pop("LD_PRELOAD", None)
    with tempfile.TemporaryDirectory() as tmp:
        env["HIPEMU_LDSPROF"] = os.path.join(tmp, "lds.txt")
        out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-p", "no:xdist"] + pytest_args,
                             cwd=ROOT, env=env, capture_output=True, text=True)
        print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-2000:])
        if not os.path.exists(env["HIPEMU_LDSPROF"]):
            sys.exit("no LDS accesses were recorded")
        lib = os.path.join(ROOT, "tools", "hipemu", "_build_race" + (f"_{args.tag}" if args.tag else ""), "lib", "libdali_amd_kernels.so")
        kernels = collect(env["HIPEMU_LDSPROF"], lib)
    for name, K in sorted(kernels.items(), key=lambda kv: -kv[1]["base"] - kv[1]["conflict"]):
        if args.kernel and args.kernel not in name:
            continue
        tot = K["base"] + K["conflict"]
        print(f"\n{name}: {K['instr']} LDS wave instructions, {K['base']} conflict-free cycles + {K['conflict']} conflict cycles "
              f"(conflict / all = {K['conflict'] / max(1, tot):.3f}, cycles per instruction {tot / max(1, K['instr']):.2f})")
        for (loc, size, rw), (instr, base, conflict, lanes) in sorted(K["sites"].items(), key=lambda kv: -kv[1][2])[:args.top]:
            print(f"  {conflict:>10} conflict / {base:>10} base  {rw}{size:<2} x{instr:<9} lanes/instr {lanes / max(1, instr):5.1f}  {loc}  | {source_line(loc)}")


if __name__ == "__main__":
    main()
