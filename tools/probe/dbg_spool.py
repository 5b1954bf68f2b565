import sys, torch
sys.path.insert(0, '.')
from lpi_amd import _lib
from lpi_amd._lib import BF16, call
DEV='cuda:0'
B,L,H=2,21,2
d=64*H
g=torch.Generator().manual_seed(1)
x=(torch.randn(B*L,d,generator=g)).half()
gamma=torch.ones(d); beta=torch.zeros(d)
W=(torch.randn(3*d,d,generator=g)*d**-0.5).bfloat16(); bq=torch.zeros(3*d)
q=torch.randn(B,d,generator=g).bfloat16()
x64=x.double(); mean=x64.mean(1); rstd=1/(x64.var(1,unbiased=False)+1e-5).sqrt()
s=torch.cuda.current_stream().cuda_stream
scratch=torch.zeros(4*B*H*d,device=DEV); lse=torch.zeros(B,H,device=DEV); ctx=torch.zeros(B,d,device=DEV,dtype=torch.bfloat16)
call("lpi_spool_attn_fwd", BF16,B,L,H,q.to(DEV),d,W.to(DEV),d,bq.to(DEV),x.to(DEV),d,mean.float().to(DEV),rstd.float().to(DEV),gamma.to(DEV),beta.to(DEV),scratch,lse,ctx,d,s)
torch.cuda.synchronize()
n=B*H*d
qt=scratch[:n].view(B,H,d).cpu(); hbar=scratch[n:2*n].view(B,H,d).cpu()
qt_ref=torch.einsum('bhc,hcj->bhj', q.double().view(B,H,64), W.double()[d:2*d].view(H,64,d))/8
print('qt nan',torch.isnan(qt).any().item(),'err',(qt.double()-qt_ref).abs().max().item(), qt_ref.abs().max().item())
print('lse',lse.cpu())
print('hbar nan',torch.isnan(hbar).any().item(), hbar[0,0,:8])
print('ctx',ctx[0,:8])
