import sys, os, json
import numpy as np
ROOT="/root/repo" if os.path.isdir("/root/repo/dspeed_amd") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests")); sys.path.insert(0, os.path.join(ROOT,"tools"))
import recipes
from bench_configs import synth, timed
from dspeed_amd.device import DeviceArray, Stream, sync
from dspeed_amd.processing_chain import WaveformInput, build_processing_chain
rows=131072
st=Stream()
wf, bl, _tp = synth(rows, 8192, np.int16, st, bl_lo=-3000.0, bl_hi=3000.0); sync()
tb={"waveform": WaveformInput(wf,16.0,48000.0),"baseline":bl}
for outs in (["bl_std","pz_std"], ["A_max","tp_aoe_max"]):
    chain,_,_=build_processing_chain(recipes.ICPC, tb, outputs=outs)
    chain.link(tb,{k: DeviceArray((rows,),np.float32) for k in outs}); chain._ensure()
    dt=timed(chain, steps=5, warmup=2)
    print(json.dumps({"outputs":outs,"ms":dt*1e3,"kernels":[k for _w,k in chain.kernels()]}))
