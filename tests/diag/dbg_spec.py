import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import f2cnn_oracle as orc
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
ctx=_lib.default_context()
C=128
coefs=filters.make_erb_filters(16000, filters.centre_freqs(16000,C,100))
def run(w, lpf=False, **o):
    offs=np.array([0,len(w)],np.int64); env=np.full(C*len(w),np.nan)
    with ctx.options(**o):
        ctx.filterbank_envelope_fused(w,_lib.WAVE_I16,offs,coefs,1,C,lpf,50.0,_lib.FFT_F32,env,None,_lib.MEM_HOST)
        fl=ctx.get_option("spectral_flagged")
    return env.reshape(C,-1), fl
w=orc.synth_utterance(1,16000)
w0=w.copy(); w0[12000:]=0
for name,x in (("noise",w),("quiet tail",w0)):
    g,fl=run(x,spectral=1,spectral_tol=1.0)
    ref=orc.filter_and_envelope(x,coefs,False,0)
    e=np.abs(g-ref).max(axis=1)/np.abs(ref).max(axis=1)
    print(name,"flagged",fl,"err max",e.max(),"chan",e.argmax(),"median",np.median(e), "nan",np.isnan(g).sum())
    print(" per chan:",np.array2string(e[::16],precision=2))
    c=64
    d=np.abs(g[c]-ref[c]); print(" chan64 err profile:", [float(d[i:i+2000].max()) for i in range(0,16000,2000)], "refmax",ref[c].max())
    print(" g[64,:6]",g[c,:6]," ref",ref[c,:6])
    print(" g[64,8000:8006]",g[c,8000:8006]," ref",ref[c,8000:8006])
