import sys, time, ctypes as C
sys.path.insert(0,'.')
import torch
from carla_driving_rl_agent_amd import _lib
lib=_lib.load()
S=lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P=lambda t: C.c_void_p(t.data_ptr())
M,K,N=64,16,16
a=torch.randn(M,K,device='cuda'); w=torch.randn(K,N,device='cuda'); b=torch.randn(N,device='cuda'); y=torch.empty(M,N,device='cuda')
s=S()
for _ in range(100): lib.cdrl_gemm_nn(P(a),K,0,P(w),N,1,P(b),P(y),N,0,M,N,K,0,s)
torch.cuda.synchronize()
for n in (2000,):
    t0=time.perf_counter()
    for _ in range(n): lib.cdrl_gemm_nn(P(a),K,0,P(w),N,1,P(b),P(y),N,0,M,N,K,0,s)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print('gemm_nn tiny: enqueue us/launch',(t1-t0)/n*1e6,'total us/launch',(t2-t0)/n*1e6)
# torch op for comparison
x=torch.zeros(64,device='cuda')
for _ in range(100): x.add_(1)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(2000): x.add_(1)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print('torch add_: enqueue us/launch',(t1-t0)/2000*1e6,'total',(t2-t0)/2000*1e6)
