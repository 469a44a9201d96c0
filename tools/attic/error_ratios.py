"""Dev tool (GPU): per-layer gradient error of the f16x3 product mode and of the fp32 device mode against the float64 oracle, as multiples of
the error a host fp32 evaluation makes, at the trained 8x64 fixture weights perturbed by a relative scale (argv[1]); round-3 study of the
selection effect at an optimiser's own optimum.  python tools/error_ratios.py [scale]"""
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from oracle import pinn_oracle as po, golden_points as gp
from pinn_elastodynamics_amd.hip_engine import HipEngine
gd='/root/repo/tests/golden'; dev=torch.device('cuda:0')
import sys as _s
SC=float(_s.argv[1]) if len(_s.argv)>1 else 0.0
for case in ('wave64',):
    w=np.load(f"{gd}/weights_{case}.npz"); g=np.load(f"{gd}/golden_{case}_32k.npz")
    layers=[int(v) for v in w["layers"]]; L=len(layers)-1
    flat=po.pack_params([w[f"W{i}"] for i in range(L)],[w[f"b{i}"] for i in range(L)])
    flat=flat*(1+SC*np.random.default_rng(77).standard_normal(flat.size))
    lb,ub,norm,n=g["lb"],g["ub"],bool(g["normalize"]),int(g["n"])
    X=gp.wave_points(lb,ub,tuple(g["src"]),n); tw=np.ones(7)/n
    for m in (8192, n):
        Xm=X[:m]; twm=np.ones(7)/m
        _,g64,_=po.wave2d_loss_grad(flat,layers,Xm[:,0],Xm[:,1],Xm[:,2],lb,ub,norm,term_weights=twm)
        _,g32,_=po.wave2d_loss_grad(flat.astype(np.float32),layers,Xm[:,0],Xm[:,1],Xm[:,2],lb,ub,norm,term_weights=twm,dtype=np.float32)
        th=torch.from_numpy(flat.astype(np.float32)).to(dev); xs=[torch.from_numpy(np.ascontiguousarray(Xm[:,k],dtype=np.float32)).to(dev) for k in range(3)]
        out={}
        for prec in ('f16x3','fp32'):
            eng=HipEngine(layers,precision=prec,device=dev,max_points=1024)
            _,gr=eng.wave_loss_grad(th,*xs,lb,ub,norm,twm); out[prec]=gr.cpu().numpy().astype(np.float64)
        W64,b64=po.unpack_params(g64,layers)
        def errs(gv):
            W,b=po.unpack_params(np.asarray(gv,np.float64),layers)
            return [np.linalg.norm(W[l]-W64[l])/np.linalg.norm(W64[l]) for l in range(L)],[np.linalg.norm(b[l]-b64[l])/np.linalg.norm(b64[l]) for l in range(L)]
        eW16,eb16=errs(out['f16x3']); eW32,eb32=errs(g32); eWd,ebd=errs(out['fp32'])
        print(case,m,'W: f16x3/hostfp32', ' '.join('%.1f'%(a/b) for a,b in zip(eW16,eW32)), '| b:', ' '.join('%.1f'%(a/b) for a,b in zip(eb16,eb32)))
        print('   host fp32 rel err W', ' '.join('%.1e'%a for a in eW32), '  dev-fp32/host', ' '.join('%.1f'%(a/b) for a,b in zip(eWd,eW32)))
