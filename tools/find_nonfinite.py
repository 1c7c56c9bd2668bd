"""Lists the gradient / parameter tensors that are not finite after one policy + value pass (debug helper)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from tests.util import make_pair, make_batches, to_dev
B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 48, 64
oracle, eng = make_pair(B, H, W, seed=5)
pol, val = make_batches(B, H, W, seed=5)
dpol, dval = to_dev(pol), to_dev(val)
for name, fn, ap, models in (('policy', lambda: eng.policy_forward_backward(dpol), eng.policy_apply, ('policy', 'trunk')),
                             ('value', lambda: eng.value_forward_backward(dval), eng.value_apply, ('value', 'trunk'))):
    fn()
    torch.cuda.synchronize()
    for m in models:
        for k, g in eng.grad_views(m).items():
            if not torch.isfinite(g).all():
                print(name, 'GRAD', m, k, 'nonfinite', int((~torch.isfinite(g)).sum()), 'of', g.numel())
    ap()
    torch.cuda.synchronize()
    for m in models:
        for k, p in eng.param_views(m).items():
            if not torch.isfinite(p).all():
                print(name, 'PARAM', m, k, 'nonfinite', int((~torch.isfinite(p)).sum()), 'of', p.numel())
print('done')
