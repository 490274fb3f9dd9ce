# host profile of a never-repeating bench line:  python3 tools/experiments/distinct_host_profile.py trainer/set=B/distinct [callees-of]
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from geossl_amd import _lib
_lib.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
name = sys.argv[1] if len(sys.argv) > 1 else "trainer/set=B/distinct"
pr = cProfile.Profile()
pr.enable()
r = bench.run_secondary(name, dev, 0, 1)
pr.disable()
print({k: r[k] for k in r if k in ("value", "ms_per_step", "captures", "error")})
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
if len(sys.argv) > 2:
    for fn in sys.argv[2:]:
        st.sort_stats("cumtime").print_callees(fn)
else:
    st.sort_stats("tottime").print_stats(40)
print(s.getvalue()[:12000])
