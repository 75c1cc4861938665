"""BatchNorm statistics / apply / backward and the column sum at the postnet's size (55 680 x 512), us per call: the 16-byte forms
(default) or, with T2AMD_ELEMENTWISE_SCALAR=1, the scalar kernels they replaced in round 5.

    python tools/microbench_bn.py; T2AMD_ELEMENTWISE_SCALAR=1 python tools/microbench_bn.py
"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native as nv
nv.load()
dev='cuda'
M,N=55680,512
x=torch.randn(M,N,device=dev); y=torch.empty_like(x); gy=torch.randn(M,N,device=dev)
keep=(torch.rand(M,N,device=dev)>=0.5).to(torch.uint8)
gamma=torch.rand(N,device=dev)+0.5; beta=torch.randn(N,device=dev)
mean=torch.empty(N,device=dev); invstd=torch.empty(N,device=dev); ws=torch.empty(2*64*N,dtype=torch.float64,device=dev)
dg=torch.empty(N,device=dev); db=torch.empty(N,device=dev); cs=torch.empty(N,device=dev)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3*e0.elapsed_time(e1)/n
print('scalar' if os.environ.get('T2AMD_ELEMENTWISE_SCALAR') else 'vector')
print(' bn_stats  %.1f us'%t(lambda: nv.bn_stats(x,ws,mean,invstd)))
print(' bn_act_fwd %.1f us'%t(lambda: nv.bn_act_fwd(x,y,mean,invstd,gamma,beta,2,keep,2.0)))
print(' bn_act_bwd %.1f us'%t(lambda: nv.bn_act_bwd(gy,y,x,mean,invstd,gamma,2,keep,2.0,ws,dg,db)))
print(' colsum    %.1f us'%t(lambda: nv.colsum(x,ws,cs)))
# round 6: the backward with its two followers (bias column sum, bf16 halo image) as three passes and folded into stage 2
B_, T_, pad = 64, 870, 2
img = torch.empty(B_ * (T_ + 2 * pad) + 2 * pad, N, dtype=torch.bfloat16, device=dev)
def separate():
    nv.bn_act_bwd(gy, y, x, mean, invstd, gamma, 2, keep, 2.0, ws, dg, db)
    nv.colsum(gy, ws, cs)
    nv.cast_halo_bf16(gy, img, T_, pad)
print(' bn_act_bwd + colsum + cast_halo  %.1f us' % t(separate))
print(' bn_act_bwd_img                   %.1f us' % t(lambda: nv.bn_act_bwd_img(gy, y, x, mean, invstd, gamma, 2, keep, 2.0, ws, dg, db, img, T_, pad, cs)))
print(' bn_act_bwd_img (keeps f32)       %.1f us' % t(lambda: nv.bn_act_bwd_img(gy, y, x, mean, invstd, gamma, 2, keep, 2.0, ws, dg, db, img, T_, pad, cs, keep_f32=True)))
print(' cast_halo alone                  %.1f us' % t(lambda: nv.cast_halo_bf16(gy, img, T_, pad)))
x16 = torch.randn(M, 4096, device=dev).to(torch.bfloat16); x32 = x16.float(); c4 = torch.empty(4096, device=dev)
ws4 = torch.empty(2 * 64 * 4096, dtype=torch.float64, device=dev)
print(' colsum f32 55680 x 4096          %.1f us' % t(lambda: nv.colsum(x32, ws4, c4)))
print(' colsum bf16 55680 x 4096         %.1f us' % t(lambda: nv.colsum16(x16, ws4, c4)))
yimg = torch.empty_like(img)
def fwd_separate():
    nv.bn_act_fwd(x, y, mean, invstd, gamma, beta, 2, keep, 2.0)
    nv.cast_halo_bf16(y, yimg, T_, pad)
print(' bn_act_fwd + cast_halo           %.1f us' % t(fwd_separate))
print(' bn_act_fwd_img                   %.1f us' % t(lambda: nv.bn_act_fwd_img(x, y, mean, invstd, gamma, beta, 2, keep, 2.0, yimg, T_, pad)))
