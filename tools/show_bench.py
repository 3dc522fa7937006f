"""Development aid: prints the bench JSON line of a file as an indented tree (python tools/show_bench.py FILE [key-prefix ...])."""
import json
import sys


def show(d, ind=0, only=None):
    for k, v in d.items():
        if only and ind == 0 and not any(k.startswith(o) for o in only):
            continue
        if isinstance(v, dict):
            print(" " * ind + k + ":")
            show(v, ind + 2)
        else:
            print(" " * ind + f"{k}: {str(v)[:150]}")


line = [ln for ln in open(sys.argv[1]).read().strip().splitlines() if ln.startswith("{")][-1]
show(json.loads(line), 0, sys.argv[2:])
