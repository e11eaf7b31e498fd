"""What the atomic tail of ln_bwd_vec costs: the same launch with the column sums through atomics (MODE=atomic) and through
per-workgroup partial buffers (MODE=partial; the second reduction launch is a kernel of its own in the trace).
Run under rocprofv3 --kernel-trace --stats, once per mode."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neurst_amd import ops  # noqa: E402

DEV = torch.device('cuda:0')
mode = os.environ.get('MODE', 'atomic')
rows, C = int(os.environ.get('LN_ROWS', '16000')), 768
x = torch.randn(rows, C, device=DEV).to(torch.bfloat16)
y = torch.randn(rows, C, device=DEV).to(torch.bfloat16)
g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
out, mean, rstd = ops.dropout_add_layernorm_fwd(x, y, g, b, 0.1, 5)
dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
for _ in range(40):
    ops.dropout_add_layernorm_bwd(out, y, g, mean, rstd, dg, db, 0.1, 5, deterministic=(mode == 'partial'))
torch.cuda.synchronize()
print('done', mode, rows)
