"""A/B of two builds of the library on one box: python tools/probes/ab_lib.py <path/to/lib.so> -- <bench_configs args>"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ndrustfft_amd import _lib
_lib._default = _lib.Library(os.path.abspath(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "tools", "bench_configs.py")] + sys.argv[3:]
runpy.run_path(sys.argv[0], run_name="__main__")
