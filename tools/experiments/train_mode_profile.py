# main-thread profile of tools/force_trace.py <which> <mols> <steps> <mode>
import cProfile, pstats, runpy, sys, io
sys.argv = ["tools/force_trace.py"] + sys.argv[1:]
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geossl_amd import _lib
_lib.load()
torch.zeros(1, device="cuda:0")
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("tools/force_trace.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(45)
print("\n".join(l for l in s.getvalue().splitlines() if "importlib" not in l and "typing.py" not in l)[:7000])
