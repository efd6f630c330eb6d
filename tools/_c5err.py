import sys, os, time
import numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tsp-gnn_amd"))
import tspgnn
from oracle import params as P, torch_oracle as TO
for (n,B,d,T) in ((200,4,128,64),(40,8,64,32),(20,8,128,8)):
    t=tspgnn.synthetic_batch([n]*B, seed=7)
    EV,W,C,r,nv,ne=t
    params=P.init_params(d,seed=3)
    model=tspgnn.build_network(d,float_dtype=torch.bfloat16); sess=tspgnn.Session(model); sess.run(tspgnn.global_variables_initializer()); model.store.load(params)
    feed={model["EV"]:EV,model["W"]:W,model["C"]:C,model["time_steps"]:T,model["route_exists"]:r,model["n_vertices"]:nv,model["n_edges"]:ne}
    pred,last,loss=sess.run([model["predictions"],model["last_states"],model["loss"]],feed_dict=feed)
    batch={"ev_uv":EV.uv,"W":W,"C":C,"route_exists":r,"n_vertices":nv,"n_edges":ne}
    t0=time.time()
    with torch.no_grad():
        ref=TO.forward(TO.to_torch(params,torch.float64),batch,T,bf16=True)
        full=TO.forward(TO.to_torch(params,torch.float64),batch,T)
    dt=time.time()-t0
    def stats(a,b):
        a=np.asarray(a,dtype=np.float64); b=np.asarray(b,dtype=np.float64); s=np.abs(b).max()
        e=np.abs(a-b)/s
        return "max %.2e rms %.2e frac>1e-3 %.2e"%(e.max(), np.sqrt((e**2).mean()), (e>1e-3).mean())
    print("n=%d B=%d d=%d T=%d oracle %.0fs"%(n,B,d,T,dt))
    print("  vs bf16 oracle: E.h", stats(last["E"].h, ref["last_states"]["E"][0].numpy()), "| V.c", stats(last["V"].c, ref["last_states"]["V"][1].numpy()), "| pred", stats(pred, ref["predictions"].numpy()), "loss", abs(float(loss)-ref["loss"].item()))
    print("  vs fp32 semantics: E.h", stats(last["E"].h, full["last_states"]["E"][0].numpy()), "| pred", stats(pred, full["predictions"].numpy()))
