"""bench.py against another build of the library (A/B runs): python tools/dev/bench_with_lib.py <lib.so> [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd import _native as nv
nv.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import __graft_entry__ as ge
ge.build_native = lambda: None            # (the library under test is already built)
import bench
bench.main()
