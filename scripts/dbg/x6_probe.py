import sys, ctypes as C, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rcot_amd import lib
from rcot_amd.ops import HipBackend
be = HipBackend(); be.prec = lib.PREC_BF16X6
B, Ci, Co, N = 8, 96, 510, 4096
W, X = torch.randn(Co, Ci, device="cuda") * 0.1, torch.randn(B, Ci, N, device="cuda")
WT, WP = (torch.zeros(*s, device="cuda") for s in be.pack_shapes(Co, Ci))
(st6,), (sp6,) = be.split6_shapes(Co, Ci)
s6 = (torch.zeros(st6, device="cuda"), torch.zeros(sp6, device="cuda"), None)
be.pack_weight(W, WT, WP, None, None, s6)
Y = torch.zeros(B, Co, N, device="cuda")
be.conv1x1_fwd(W, X, Y, packed=(WT, WP, None, None, s6))
torch.cuda.synchronize()
buf = C.create_string_buffer(192); be.L.rcot_last_kernel(buf, 192); print(buf.value)
ref = torch.einsum("oc,bcn->bon", W.double(), X.double())
print(float((Y.double() - ref).abs().max() / ref.abs().max()))
print("prec", be.prec, "split_of", be._split_of((WT, WP, None, None, s6)) is s6)
v = be._bcn_z
rc = be.L.rcot_gemm_kmajor(WT.data_ptr(), WT.stride(0), 0, 0, WT.shape[0], X.data_ptr(), N, Ci * N, 0, Y.data_ptr(), N, Co * N, 0, None, 0, 0, 0,
                           None, 0, 0, None, None, 0, 0, None, None, None, None, s6[0].data_ptr(), B, 1, Co, N, Ci, 0.0, be.ws.data_ptr(), be.ws_bytes,
                           2, be._st())
torch.cuda.synchronize()
be.L.rcot_last_kernel(buf, 192); print("raw rc", rc, buf.value)
print(float((Y.double() - ref).abs().max() / ref.abs().max()))
