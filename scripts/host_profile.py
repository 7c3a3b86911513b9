import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from rcot_amd import lib
from rcot_amd.net_restormer import F_net, T_net
from rcot_amd.ops import default_backend
from rcot_amd.synth import make_batch
from rcot_amd.trainer import FlatOptimizer, MinimaxStep
B, P = 8, 128
be = default_backend(); be.prec = lib.PREC_BF16X3
Tn, Fn = T_net(decoder=True, seed=1234), F_net(patch_size=P, seed=1235)
st = MinimaxStep(Tn, Fn, FlatOptimizer(Tn, "RMSprop", 5e-5), FlatOptimizer(Fn, "RMSprop", 1e-4), 1.0, 10000.0)
de = [2] * B; st.set_de_ids(de)
de_dev = torch.tensor(de, dtype=torch.int32, device="cuda")
_, x, y = make_batch(1, B, P, de, unpaired=False); x, y = x.cuda(), y.cuda(); al = torch.rand(B).cuda()
for _ in range(3): st.iteration(x, y, de_dev, al, True)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): st.iteration(x, y, de_dev, al, True)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
