"""CPU, oracle only: how far apart are two evaluations of the SAME bf16-operand DAC model (same rounded weights, same rounding points) that
differ only in fp32 summation order (float64 accumulation) or by a tiny error in sin? -> the floor of any END-TO-END comparison of the bf16
codec kernels with DacOracle(precision="bf16"), and the reason the kernels are pinned stage by stage (tests/test_dac_stage_parity_gpu.py).
    python tools/dac_bf16_sensitivity.py [frames]   -> profiles/r04_dac_bf16_sensitivity.txt"""
import sys, math, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from oracle import dac_oracle as DA
from parler_tts_amd.synthetic import random_dac_state_dict
torch.set_num_threads(8)
spec=DA.DAC_44KHZ
dsd=random_dac_state_dict(seed=4321)
T=int(sys.argv[1]) if len(sys.argv)>1 else 60
codes=torch.randint(0,1024,(1,9,T),generator=torch.Generator().manual_seed(7))
o=DA.DacOracle(spec,dsd,precision="bf16")
ref=o.decode(codes)
o32=DA.DacOracle(spec,dsd).decode(codes)
rel=lambda a,b: float((a-b).pow(2).mean().sqrt()/b.pow(2).mean().sqrt())
print("bf16 oracle vs fp32 oracle", rel(ref,o32))
# variant: same rounding points, float64 accumulation (then to fp32), sin perturbed
class V(DA.DacOracle):
    def __init__(s,*a,sin_eps=0.0,f64=True,**k):
        super().__init__(*a,**k); s.sin_eps=sin_eps; s.f64=f64
    def decode_latents(self,z):
        w,d=self.w,"decoder.model."
        rb=DA._rb
        def conv(x,wt,b,**k):
            if self.f64: return F.conv1d(x.double(),wt.double(),b.double(),**k).float()
            return F.conv1d(x,wt,b,**k)
        def convt(x,wt,b,**k):
            if self.f64: return F.conv_transpose1d(x.double(),wt.double(),b.double(),**k).float()
            return F.conv_transpose1d(x,wt,b,**k)
        def snake(x,al):
            s=torch.sin(al*x)
            if self.sin_eps: s=s+self.sin_eps*torch.randn_like(s)
            return x+(al+1e-9).reciprocal()*s.pow(2)
        x=conv(rb(z),w[d+"0.weight"],w[d+"0.bias"],padding=3)
        for bi,s in enumerate(self.spec.decoder_rates):
            b=f"{d}{bi+1}.block."
            x=rb(snake(x,w[b+"0.alpha"]))
            x=convt(x,w[b+"1.weight"],w[b+"1.bias"],stride=s,padding=math.ceil(s/2))
            for ri,dil in enumerate((1,3,9)):
                r=f"{b}{ri+2}.block."
                y=rb(snake(x,w[r+"0.alpha"]))
                y=conv(y,w[r+"1.weight"],w[r+"1.bias"],dilation=dil,padding=3*dil)
                y=rb(snake(y,w[r+"2.alpha"]))
                y=conv(y,w[r+"3.weight"],w[r+"3.bias"])
                x=x+y
        n=len(self.spec.decoder_rates)
        x=snake(x,w[f"{d}{n+1}.alpha"])
        x=F.conv1d(x,w[f"{d}{n+2}.weight"],w[f"{d}{n+2}.bias"],padding=3)
        return torch.tanh(x)
print("f64-accumulate variant vs bf16 oracle", rel(V(spec,dsd,precision="bf16").decode(codes),ref))
for eps in (1e-7,1e-6,1e-5,1e-4):
    print("sin eps",eps, rel(V(spec,dsd,precision="bf16",sin_eps=eps,f64=False).decode(codes),ref))
