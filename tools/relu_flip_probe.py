# Diagnostic (GPU): compares per-layer dz of the HIP backward with the fp64 oracle and shows that the only
# large element-wise differences are ReLU-mask flips (BatchNorm output ~ 0).  python tools/relu_flip_probe.py B
import sys; sys.path.insert(0,'/root/repo')
import torch, torch.nn.functional as F
from oracle import tactilesr_oracle as O
import tactilesr_amd
from tactilesr_amd.model import tactileSR_model as M
from tactilesr_amd.model._train import TrainEngine
B=int(sys.argv[1])
cfg=dict(seqsCnt=1, patternFeatureExtraLayerCnt=1)
sd=O.random_state_dict(O.tactilesr_state_shapes(**cfg), 977)
g=torch.Generator().manual_seed(978)
LR=torch.rand(B,3,4,4,generator=g)*8; HR=torch.rand(B,1,40,40,generator=g)*25
# oracle with recorded conv outputs
rec=[]
orig=F.conv2d
def myconv(*a,**k):
    y=orig(*a,**k); 
    if y.requires_grad: y.retain_grad(); rec.append(y)
    return y
F.conv2d=myconv
leaves={k:v.double().requires_grad_(True) for k,v in sd.items() if O.is_trainable(k)}
full={k:(v.double() if v.is_floating_point() else v) for k,v in sd.items()}; full.update(leaves)
out=O.tactilesr_forward(full, LR.double(), training=True, new_stats={})
loss=F.mse_loss(out,HR.double()); loss.backward()
F.conv2d=orig
print("n conv", len(rec), [tuple(r.shape[1:2]) for r in rec])
m=M.TactileSR(**cfg); m.load_state_dict(sd); m=m.cuda().train()
m.train_engine().debug={}
o=m(LR.cuda()); l=F.mse_loss(o,HR.cuda()); l.backward()
dbg=m._train_engine.debug
# conv order in oracle: stem conv1, stem conv2, fuse, c31, c51, c32, c52, conf, force stem, res1, res2, head0, head
dz2=M.from_cb16(dbg["msrb0.dz2"],B,256,40,40).cpu().double()
ref=torch.cat([rec[5].grad, rec[6].grad],1)
for b in range(B):
    for h,(lo,hi) in enumerate(((0,128),(128,256))):
        e=(dz2[b,lo:hi]-ref[b,lo:hi]).abs().max()/ref[:,lo:hi].abs().max()
        print("dz2 img",b,"half",h,float(e))
dz1=M.from_cb16(dbg["msrb0.dz1"],B,128,40,40).cpu().double()
ref1=torch.cat([rec[3].grad, rec[4].grad],1)
for b in range(B):
    print("dz1 img",b,float((dz1[b]-ref1[b]).abs().max()/ref1.abs().max()))

dzh=M.from_cb16(dbg["dz_h0"],B,128,40,40).cpu().double()
gh=M.from_cb16(dbg["g_hcat"],B,128,40,40).cpu().double()
for b in range(B):
    print("dz_h0 img",b,float((dzh[b]-rec[11].grad[b]).abs().max()/rec[11].grad.abs().max()),
          "g_hcat pattern", float((gh[b,64:]-rec[7].grad[b]).abs().max()/rec[7].grad.abs().max()),
          "force", float((gh[b,:64]-rec[10].grad[b]).abs().max()/rec[10].grad.abs().max()))
# flip hypothesis: where dz2 differs most, is the BN output (pre-ReLU) ~ 0 ?
z=torch.cat([rec[5], rec[6]],1).detach()   # conv outputs (with bias) fp64
mean=z.mean(dim=(0,2,3),keepdim=True); var=z.var(dim=(0,2,3),unbiased=False,keepdim=True)
gam=torch.cat([sd["patternFeatureExtra_layer.0.conv_3_2.1.weight"], sd["patternFeatureExtra_layer.0.conv_5_2.1.weight"]]).double().view(1,-1,1,1)
bet=torch.cat([sd["patternFeatureExtra_layer.0.conv_3_2.1.bias"], sd["patternFeatureExtra_layer.0.conv_5_2.1.bias"]]).double().view(1,-1,1,1)
y=(z-mean)/torch.sqrt(var+1e-5)*gam+bet
err=(dz2-ref).abs()
idx=torch.topk(err.flatten(),5).indices
for i in idx:
    b_,c_,yy,xx=[int(v) for v in torch.unravel_index(i, err.shape)]
    print("err",float(err[b_,c_,yy,xx]),"at",(b_,c_,yy,xx),"bn_out",float(y[b_,c_,yy,xx]),"typical |bn_out|",float(y.abs().mean()))
print("elements with err>1e-3*max:", int((err>1e-3*ref.abs().max()).sum()), "of", err.numel())
