import ctypes as C, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from carla_driving_rl_agent_amd import _lib
lib = _lib.load(); DEV='cuda:0'; BF=torch.bfloat16
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for (M,K,N) in [(196608,116,116),(49152,232,232),(49152,464,768)]:
    a = torch.randn(M,K,device=DEV).to(BF); d = torch.randn(M,N,device=DEV).to(BF)
    ws = torch.zeros(int(lib.cdrl_gemm_tn_workspace_elems(M,N,K)), device=DEV)
    outs=[]
    lib.cdrl_set_op_activation_type(1)
    for rep in range(6):
        out = torch.zeros(K,N,device=DEV)
        ws.fill_(float(rep))
        _lib.check(lib.cdrl_gemm_tn(P(a),K,0,P(d),N,0,P(out),M,N,K,P(ws),0,S()))
        torch.cuda.synchronize(); outs.append(out.clone())
    lib.cdrl_set_op_activation_type(0)
    print(M,K,N,'identical runs:', [bool(torch.equal(outs[0],o)) for o in outs[1:]], 'max diff', max(float((outs[0]-o).abs().max()) for o in outs[1:]))
