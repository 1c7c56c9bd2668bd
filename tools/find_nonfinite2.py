import sys
import torch
sys.path.insert(0, '.')
from tests.util import make_pair, make_batches, to_dev
B, H, W = 32, 48, 64
oracle, eng = make_pair(B, H, W, seed=5)
pol, val = make_batches(B, H, W, seed=5)
dpol = to_dev(pol)
eng.policy_forward_backward(dpol)
torch.cuda.synchronize()
g = eng.grad_views('trunk')
for k in g:
    if k.startswith('img.s0.u0.sc_') or k.startswith('img.stem'):
        t = g[k]
        print('GRAD', k, tuple(t.shape), 'absmax %.3e' % float(t.abs().max()), 'min %.3e' % float(t.min()), 'finite', bool(torch.isfinite(t).all()))
p0 = {k: v.clone() for k, v in eng.param_views('trunk').items() if k.startswith('img.s0.u0.sc_')}
eng.policy_apply()
torch.cuda.synchronize()
m, v = eng.adam_views('trunk')
for k in p0:
    if k.endswith('moving_mean') or k.endswith('moving_var'):
        continue
    p1 = eng.param_views('trunk')[k]
    print('AFTER', k, 'param finite', bool(torch.isfinite(p1).all()), 'm absmax %.3e' % float(m[k].abs().max()) if k in m else '', 'v max %.3e' % float(v[k].max()) if k in v else '')
    if not torch.isfinite(p1).all():
        bad = ~torch.isfinite(p1.view(-1))
        idx = bad.nonzero().view(-1)[:8]
        print('   bad idx', idx.tolist(), 'grad there', g[k].view(-1)[idx].tolist(), 'old param', p0[k].view(-1)[idx].tolist(), 'm', m[k].view(-1)[idx].tolist(), 'v', v[k].view(-1)[idx].tolist())
