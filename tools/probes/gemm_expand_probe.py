"""HRNet's stage-1 expand conv (64 -> 256 @64x48, N = 128: 12.9 GFLOP against 830 MB with the residual) through the blocked-GEMM
(tuner id 10) and the streaming (8) fp32 1x1 kernels, with and without the residual tensor; stream-timed, 20 launches.
   python tools/probes/gemm_expand_probe.py     (MINDPOSE_HIP_LIB=build/conv_gemm_f32_<tag>/... for the ablation builds of tools/variant_builds.sh)"""
import ctypes, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mindpose_amd import _lib
lib=_lib.load(); dev=torch.device('cuda:0')
n,cin,cout,h,w=128,64,256,64,48
x=torch.randn(n,cin,h,w,device=dev); wt=torch.randn(cout,cin,1,1,device=dev)*0.1
scale=torch.rand(cout,device=dev)+0.5; shift=torch.randn(cout,device=dev); res=torch.randn(n,cout,h,w,device=dev); out=torch.empty(n,cout,h,w,device=dev)
d=_lib.ConvDesc(n=n,cin=cin,h=h,w=w,cout=cout,kh=1,kw=1,stride=1,pad_top=0,pad_left=0,conv_h=h,conv_w=w,out_h=h,out_w=w,out_mul=1,out_rep=1,out_off_y=0,out_off_x=0,relu=1,flags=0)
pd=torch.empty(lib.mp_conv_packed_weight_bytes(cout,cin,1,1)//4,device=dev)
_lib.check(lib.mp_conv_pack_weight(_lib.ptr(wt),_lib.ptr(pd),cout,cin,1,1,0,0,0,_lib.stream()),'pack')
def run(v, r):
    def fn(): _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d),v,_lib.ptr(x),_lib.ptr(pd),_lib.ptr(scale),_lib.ptr(shift),_lib.ptr(r) if r is not None else None,None,_lib.ptr(out),_lib.stream()),'conv')
    for _ in range(5): fn()
    ts=[]
    for _ in range(7):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1)/20*1e3)
    return statistics.median(ts)
for v in (10, 8):
    print('variant',v,'with res %.1f us'%run(v,res),' without %.1f us'%run(v,None))
