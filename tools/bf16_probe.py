"""bf16-vs-fp32 and dispatch A/B of the engine's parameter gradients at batch 16 / 4 (a quick GPU probe)."""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from conftest import cosine, rel_err
from dynaboa_amd import assets, _lib
from dynaboa_amd.hmr import get_layout, hmr
lib=_lib.load()
for B in (16, 4):
    L=get_layout(B)
    mp = assets.make_smpl_mean_params(identity_pose=False, seed=3)
    ck = assets.make_synthetic_checkpoint(22, mp, randomize_norm=True, prefix="")["model"]
    m = hmr(mp, seed=1).to("cuda:0").eval(); m.load_state_dict(ck, strict=True)
    img = assets.make_frame(0, B, seed=22)["image"].to("cuda:0")
    g = torch.Generator().manual_seed(7)
    wr, ws_, wc = (torch.randn(s, generator=g).to("cuda:0") for s in ((B,24,3,3),(B,10),(B,3)))
    res={}
    for name,(bf,k4) in dict(fp32=(0,1), fp32_nok4=(0,0), bf16=(1,1)).items():
        L.set_bf16(bool(bf)); lib.dyb_set_option(b"k4", k4); lib.dyb_set_option(b"k4_bwd", k4); m.theta.grad=None
        r,s,c = m(img); ((r*wr).sum()+(s*ws_).sum()+(c*wc).sum()).backward()
        res[name]=m._layout1.unpack(m.theta.grad)
    L.set_bf16(False); lib.dyb_set_option(b"k4", 1); lib.dyb_set_option(b"k4_bwd", 1)
    for other in ("fp32_nok4","bf16"):
        cos={k: cosine(res[other][k].cpu().numpy(), res["fp32"][k].cpu().numpy()) for k in res["fp32"]}
        low=sorted(cos.items(), key=lambda kv: kv[1])[:5]
        v=np.array(list(cos.values()))
        print("B",B,other,"min",round(v.min(),4),"median",round(float(np.median(v)),5),"n<0.99",int((v<0.99).sum()), [(k,round(x,3)) for k,x in low])
