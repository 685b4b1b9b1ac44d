import sys; sys.path[:0]=[".","probnmn-clevr_amd"]
import torch, ctypes
from probnmn import _hip
from probnmn.modules import seq2seq_base as sb
dev=torch.device("cuda:0")
g=torch.Generator().manual_seed(0)
r=lambda *s, scale=1.0: (torch.randn(*s, generator=g)*scale).to(dev)
B,T,S,Hd=512,46,27,256
enc,h0=r(B,S,Hd),r(B,Hd); mask=torch.ones(B,S,device=dev)
w_c,w_hh=r(4*Hd,Hd,scale=0.05),r(4*Hd,Hd,scale=0.05); xe=r(B,T,4*Hd,scale=0.5)
w_p,b_p=r(96,Hd),r(96)
# monkeypatch workspace to keep it
keep={}
orig=sb._decoder_workspace
def ws(batch, backward, device):
    t=orig(batch, backward, device); keep['ws']=t; return t
sb._decoder_workspace=ws
for _ in range(3):
    hs,_=sb._AttnLSTMDecoder.apply(xe,None,enc,mask,h0,w_c,w_hh,w_p,b_p,0,T,5,0,0,1,2)
torch.cuda.synchronize()
dbg=keep['ws'][16384:16384+64].view(torch.int64).cpu().tolist()
names=["attention","signal1","wait1","gates(load+mfma)","lds+cell+stores","signal2","wait2"]
tot=sum(dbg[:7])
for n,v in zip(names,dbg): print("%-18s %8.2f us/step (%4.1f%%)"%(n, v/T/100.0, 100.0*v/tot))   # clock64 = 100 MHz? print raw too
print("raw", dbg[:7], "total cycles/step", tot/T)
