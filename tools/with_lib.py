#!/usr/bin/env python3
"""Run a script of this tree against ANOTHER build of libmpx.so without touching the product library:
    python tools/with_lib.py /tmp/libmpx_B.so bench.py --steps 2 ...
The process binds `_lib.LIB_PATH` to the given file before the script starts (the way tools/probes/*.py bind the
diagnostic build); network_interpretation_imagenet_amd/libmpx.so and its stamp are never written, so a failed or interrupted
A/B run cannot leave a probe build (possibly a timing-only, wrong-result one) installed as the product."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) < 3:
    raise SystemExit(__doc__)
lib, script = os.path.abspath(sys.argv[1]), sys.argv[2]
if not os.path.exists(lib):
    raise SystemExit("with_lib: %s does not exist" % lib)
from network_interpretation_imagenet_amd import _lib  # noqa: E402

_lib.LIB_PATH = lib
os.environ["MPX_LIB_PATH"] = lib         # child processes (bench.py --gpus N starts its own ranks) bind the same file
sys.stderr.write("with_lib: this process binds %s (NOT the product library)\n" % lib)
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
