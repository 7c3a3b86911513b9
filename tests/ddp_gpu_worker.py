"""Helper of tests/test_pipeline_gpu.py: one rank of a 2-rank data-parallel run of the trainer CLI on ONE GPU (gloo
transport for the collectives — RCCL refuses two ranks on a device), dumping the replica's parameters at the end."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rcot_amd import trainer  # noqa: E402

out = sys.argv[1]
Tn, Fn = trainer.main(sys.argv[2:])
torch.cuda.synchronize()
torch.save({"T": Tn.store.flat.cpu(), "F": Fn.store.flat.cpu(), "seed": trainer.opt.seed}, out)
