# one-process profile target for rocprofv3: a few eager SchNet forward passes (inference, no saved activations) at
# BASELINE config 2 (bs = 1024 molecules x 18 atoms, SchNet F=128 L=6 G=51 cutoff 5 A, fp32)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geossl_amd import pretrain_GeoSSL as pg
from geossl_amd.Geom3D.models import SchNet
from geossl_amd.synthetic import make_batch
dev = 'cuda:0'
torch.manual_seed(0)
model = SchNet(128, 128, 6, 51, 5.0, node_class=9).to(dev)
bt = pg.Batch.from_numpy(make_batch(1024, seed=0), dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with torch.no_grad():
    for i in range(steps):
        out = model(bt.x[:, 0], bt.positions, bt.batch)
torch.cuda.synchronize()
print("checksum", float(out.double().sum()))
