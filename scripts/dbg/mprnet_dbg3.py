"""debug: the ten configs[0] iterations — reference fixture vs MPRNet on HIP vs the stock-ops loop on the GPU (torch ops)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import seeded_tensor
from rcot_amd import mprnet as MP, params as P
from rcot_amd.synth import make_batch
from test_mprnet_gpu import _params, _step
from rcot_amd.ops import default_backend
fx = np.load(os.path.join(ROOT, "tests/golden/mprnet.npz"))
cfg = [int(v) for v in fx["traj_cfg"]]
B, ps, steps, _sT, sF, sb, sa = cfg[:7]
de = cfg[7:]
be = default_backend()
_Tn, _Fn, st = _step(be, ps, sF)
st.set_de_ids(de)
de_dev = torch.tensor(de, dtype=torch.int32).cuda()
Tm, Fm = MP.MPRNetT(seed=0, device="cuda"), MP.FNetTorch(ps, seed=0, device="cuda")
Tm.load_state_dict(_params())
Fm.load_state_dict({k: torch.from_numpy(v) for k, v in P.seeded_params(P.fnet_param_shapes(ps), sF, "F").items()})
To, Fo = torch.optim.RMSprop(Tm.parameters(), lr=5e-5), torch.optim.RMSprop(Fm.parameters(), lr=1e-4)
for i in range(steps):
    _, x, y = make_batch(sb + i, B, ps, de)
    alpha = seeded_tensor(sa + i, (B, 1, 1, 1), lo=0.0, hi=1.0).view(B)
    st.iteration(x.cuda(), y.cuda(), de_dev, alpha.cuda(), False)
    s = st.scalars()
    m = MP.torch_minimax_iteration(Tm, Fm, To, Fo, x.cuda(), y.cuda(), de, alpha.cuda(), 1.0, 10000.0, False)
    print(i, "ref", fx["traj"][i].tolist(), "| hip", [round(s[k], 6) for k in ("Loss_F", "Loss_T", "Loss_mse")], "| torch-gpu", [round(m[k], 6) for k in ("Loss_F", "Loss_T", "Loss_mse")], flush=True)
