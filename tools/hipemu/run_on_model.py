#!/usr/bin/env python3
"""Runs a python script of the repository with the TESTS' loader pointed at the CPU model of the kernels:

    python tools/hipemu/run_on_model.py bench.py --steps 2 --batch 8 ...
    python tools/hipemu/run_on_model.py -c "import __graft_entry__ as g; g.smoke()"

(test infrastructure: the script itself - bench.py, __graft_entry__.py - knows nothing about the model; what runs is its
whole host path with the kernels executed by the model, i.e. a dry run of the script in a container without a GPU.  The
numbers it prints mean nothing.)"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import hipemu_env  # noqa: E402

hipemu_env.activate(os.environ.get("DALI_AMD_HIPEMU_BUILD", ""))
if sys.argv[1] == "-c":
    sys.argv = sys.argv[1:]
    exec(compile(sys.argv[1], "<-c>", "exec"), {"__name__": "__main__"})
    sys.exit(0)
script = sys.argv[1]
sys.argv = sys.argv[1:]
runpy.run_path(os.path.join(ROOT, script) if not os.path.isabs(script) else script, run_name="__main__")
