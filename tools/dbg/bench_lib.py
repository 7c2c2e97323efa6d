"""Run bench.py in-process against another build of the library: python tools/dbg/bench_lib.py <lib.so> [bench args]"""
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from pulselib_amd import _native  # noqa: E402

_native._SO = Path(sys.argv[1]).resolve()
sys.argv = [str(ROOT / "bench.py"), "--inproc"] + sys.argv[2:]
runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
