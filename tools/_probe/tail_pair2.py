import sys
sys.path.insert(0, ".")
sys.argv = [sys.argv[0]]
from nsol_amd import _lib
_lib.set_param("pdk_verbose", 1)
_lib.set_param("pd2_enable", 0)
exec(open("tools/_probe/tail_pair.py").read().split("for pd2 in")[0])
for i in range(12):
    print("2 it %.4f" % t(2, reps=10), flush=True)
