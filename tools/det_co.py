#!/usr/bin/env python3
"""Co-execution determinism probe: the fused 1x1-conv backward (bf16 storage) on one stream while another stream runs
(a) nothing, (b) the LDS filter-gradient GEMM with several column blocks, (c) a torch copy kernel.
usage: CDRL_TN_LDS=2 tools/det_co.py [reps]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from carla_driving_rl_agent_amd import _lib
lib = _lib.load(); DEV = 'cuda:0'; BF = torch.bfloat16
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
dev = lambda x, dt=torch.float32: torch.tensor(np.asarray(x, np.float32), device=DEV).to(dt)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(0)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
SA, SB = C.c_void_p(sA.cuda_stream), C.c_void_p(sB.cuda_stream)
G, Mg, K, N = 4, 12288, 232, 232
M = G * Mg
xb = dev(rng.standard_normal((M, K)), BF); yb = dev(rng.standard_normal((M, N)) * 1.3 + 0.2, BF)
w = dev(rng.standard_normal((K, N)) / np.sqrt(K))
gam, bet = dev(rng.uniform(0.5, 1.5, N)), dev(rng.uniform(1.0, 3.0, N))
ctot, coff = 2 * N, N
dob = dev(rng.standard_normal((M, ctot)), BF)
wtp = torch.zeros(int(lib.cdrl_pwconv_pack_elems(K, N)), device=DEV)
lib.cdrl_set_op_activation_type(1)
with torch.cuda.stream(sA):
    _lib.check(lib.cdrl_pwconv_pack(P(w), N, K, 1, N, P(wtp), 1, SA))
    stats = torch.zeros(4 * G * N, device=DEV); tmp = torch.zeros((M, N), dtype=BF, device=DEV)
    ws0 = torch.zeros(G * 256 * 2 * N, dtype=torch.float64, device=DEV); mm, mv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    _lib.check(lib.cdrl_bn_train_fwd(P(yb), G, Mg, N, P(gam), P(bet), P(mm), P(mv), 1, 1, P(tmp), N, 0, 0, P(stats), P(ws0), SA))
# co-runner operands: the head conv's filter gradient (M = 49152, K = 464, N = 768)
M2, K2, N2 = 49152, 464, 768
a2 = torch.randn(M2, K2, device=DEV).to(BF); d2 = torch.randn(M2, N2, device=DEV).to(BF)
ws2 = torch.zeros(int(lib.cdrl_gemm_tn_workspace_elems(M2, N2, K2)), device=DEV); out2 = torch.zeros(K2, N2, device=DEV)
big = torch.zeros(256 << 20, dtype=torch.uint8, device=DEV); big2 = torch.zeros_like(big)
torch.cuda.synchronize()
for mode in ('alone', 'with_tn_lds'):
    outs = []
    for rep in range(reps):
        ws = torch.full((int(lib.cdrl_pwconv_bn_bwd_workspace_bytes(G, Mg, N, K)),), rep % 251, dtype=torch.uint8, device=DEV)
        dg, dbt, coef = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(3 * G * N, device=DEV)
        dx = torch.zeros((M, K), dtype=BF, device=DEV); dw, db = torch.zeros((K, N), device=DEV), torch.zeros(N, device=DEV)
        torch.cuda.synchronize()
        if mode == 'with_tn_lds':
            for _ in range(3): _lib.check(lib.cdrl_gemm_tn(P(a2), K2, 0, P(d2), N2, 0, P(out2), M2, N2, K2, P(ws2), 0, SB))
        elif mode == 'with_copy':
            with torch.cuda.stream(sB):
                for _ in range(3): big2.copy_(big)
        _lib.check(lib.cdrl_pwconv_bn_bwd_packed(P(dob), ctot, coff, ctot, 1, P(yb), P(stats), P(xb), K, 0, None, P(w), G, Mg, N, K, P(dg), P(dbt), P(coef),
                                                 P(dx), K, 0, 0, P(dw), P(db), P(ws), P(wtp), 1, SA))
        torch.cuda.synchronize()
        outs.append((dx.clone(), db.clone(), dw.clone(), dg.clone(), out2.clone(), coef.clone()))
    print(mode, 'dx', [int(torch.equal(outs[0][0], o[0])) for o in outs[1:]], 'db', [int(torch.equal(outs[0][1], o[1])) for o in outs[1:]],
          'dw', [int(torch.equal(outs[0][2], o[2])) for o in outs[1:]], 'dgamma', all(torch.equal(outs[0][3], o[3]) for o in outs[1:]),
          'co-runner out', all(torch.equal(outs[0][4], o[4]) for o in outs[1:]))
    if mode != 'alone':
        bad = [i for i, o in enumerate(outs) if not torch.equal(outs[0][0], o[0])]
        if bad:
            d = (outs[0][0].float() - outs[bad[0]][0].float())
            nz = d.nonzero()
            print('   first differing rep', bad[0], 'elements', nz.shape[0], 'of', d.numel(), 'rows', int(nz[:, 0].min()), '..', int(nz[:, 0].max()),
                  'cols', int(nz[:, 1].min()), '..', int(nz[:, 1].max()), 'max abs', float(d.abs().max()))
            rows = torch.unique(nz[:, 0])
            print('   distinct rows', rows.numel(), 'first rows', rows[:24].tolist())
            print('   row % 32:', (rows % 32).tolist())
            print('   tile index in group:', ((rows % Mg) // 32).tolist()[:40])
            good, badt = outs[0][0], outs[bad[0]][0]
            cf = outs[0][5].view(3, G, N); st4 = stats.view(4, G, N)
            idx = torch.tensor([((coff + n) & 1) * (ctot >> 1) + ((coff + n) >> 1) for n in range(N)], device=DEV)
            wb = w.to(BF).double()                       # [K out][N in]
            def a_row(r, rz=None, ry=None):
                g = r // Mg
                dz = dob[r if rz is None else rz, idx].float(); y = yb[r if ry is None else ry].float()
                z = st4[2, g] * y + st4[3, g]
                dzm = torch.where((z > 0) & (z < 6), dz, torch.zeros_like(dz))
                xh = (y - st4[0, g]) * st4[1, g]
                return cf[0, g] * (dzm - cf[1, g] - xh * cf[2, g])
            for r in rows[:30].tolist():
                dd = (badt[r].double() - good[r].double())
                cols = dd.nonzero().flatten()
                blk = 0 if int(cols.min()) < 128 else 1
                sel = torch.arange(0, 128, device=DEV) if blk == 0 else torch.arange(128, 232, device=DEV)
                best = None
                for kk in range(N // 2):
                    A2 = wb[sel][:, 2 * kk:2 * kk + 2]           # [cols][2]
                    sol = torch.linalg.lstsq(A2, dd[sel].unsqueeze(1)).solution.flatten()
                    res = float((A2 @ sol - dd[sel]).abs().max())
                    if best is None or res < best[0]: best = (res, kk, sol.tolist())
                a0 = a_row(r)
                kk = best[1]
                cand = {'dz=0': None}
                print('   bad row', r, 'r%32', r % 32, 'block', blk, 'best single pair kk', kk, 'residual', round(best[0], 4), 'delta a', [round(v, 4) for v in best[2]],
                      'a', [round(float(a0[2 * kk]), 4), round(float(a0[2 * kk + 1]), 4)],
                      'a(row-32)', [round(float(v), 4) for v in a_row(r, r - 32, r - 32)[2 * kk:2 * kk + 2]] if r % Mg >= 32 else None,
                      'a(dz row-32)', [round(float(v), 4) for v in a_row(r, r - 32, None)[2 * kk:2 * kk + 2]] if r % Mg >= 32 else None,
                      'a(y row-32)', [round(float(v), 4) for v in a_row(r, None, r - 32)[2 * kk:2 * kk + 2]] if r % Mg >= 32 else None)
lib.cdrl_set_op_activation_type(0)
