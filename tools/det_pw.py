import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from carla_driving_rl_agent_amd import _lib
lib = _lib.load(); DEV='cuda:0'; BF=torch.bfloat16
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
dev = lambda x, dt=torch.float32: torch.tensor(np.asarray(x, np.float32), device=DEV).to(dt)
rng = np.random.default_rng(0)
for (G, Mg, K, N) in [(4, 12288, 232, 232), (4, 49152, 116, 116)]:
    M = G * Mg
    xb = dev(rng.standard_normal((M, K)), BF); yb = dev(rng.standard_normal((M, N)) * 1.3 + 0.2, BF)
    w = dev(rng.standard_normal((K, N)) / np.sqrt(K))
    gam, bet = dev(rng.uniform(0.5, 1.5, N)), dev(rng.uniform(1.0, 3.0, N))
    ctot, coff = 2 * N, N
    dob = dev(rng.standard_normal((M, ctot)), BF)
    wtp = torch.zeros(int(lib.cdrl_pwconv_pack_elems(K, N)), device=DEV)
    _lib.check(lib.cdrl_pwconv_pack(P(w), N, K, 1, N, P(wtp), 1, S()))
    lib.cdrl_set_op_activation_type(1)
    stats = torch.zeros(4 * G * N, device=DEV); tmp = torch.zeros((M, N), dtype=BF, device=DEV)
    ws0 = torch.zeros(G * 256 * 2 * N, dtype=torch.float64, device=DEV); mm, mv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    _lib.check(lib.cdrl_bn_train_fwd(P(yb), G, Mg, N, P(gam), P(bet), P(mm), P(mv), 1, 1, P(tmp), N, 0, 0, P(stats), P(ws0), S()))
    outs = []
    for rep in range(12):
        ws = torch.full((int(lib.cdrl_pwconv_bn_bwd_workspace_bytes(G, Mg, N, K)),), rep % 251, dtype=torch.uint8, device=DEV)
        dg, dbt, coef = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(3 * G * N, device=DEV)
        dx = torch.zeros((M, K), dtype=BF, device=DEV); dw, db = torch.zeros((K, N), device=DEV), torch.zeros(N, device=DEV)
        _lib.check(lib.cdrl_pwconv_bn_bwd_packed(P(dob), ctot, coff, ctot, 1, P(yb), P(stats), P(xb), K, 0, None, P(w), G, Mg, N, K, P(dg), P(dbt), P(coef),
                                                 P(dx), K, 0, 0, P(dw), P(db), P(ws), P(wtp), 1, S()))
        torch.cuda.synchronize()
        outs.append((dx.clone(), db.clone(), dw.clone(), dg.clone()))
    lib.cdrl_set_op_activation_type(0)
    print(G, Mg, K, N, 'dx', [bool(torch.equal(outs[0][0], o[0])) for o in outs[1:]], 'db', [bool(torch.equal(outs[0][1], o[1])) for o in outs[1:]],
          'dw', all(torch.equal(outs[0][2], o[2]) for o in outs[1:]))
