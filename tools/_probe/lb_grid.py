import sys, runpy
sys.path.insert(0, ".")
from nsol_amd import _lib
blocks = int(sys.argv[1])
sys.argv = [sys.argv[0]]
_lib.set_param("max_grid_blocks", blocks)
runpy.run_path("tools/bench_lbfgsb_kernels.py", run_name="__main__")
