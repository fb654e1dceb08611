"""FASTA scan throughput (GPU box): 60-column wrapped records and long single-line records."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exon_duckdb_amd import abi, device
from oracle import pyoracle
torch.cuda.set_device(0)
def time_ms(fn, reps=5, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ev=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); ev.append((a,b))
    torch.cuda.synchronize(); t=sorted(x.elapsed_time(y) for x,y in ev); return t[len(t)//2]
out={}
body = pyoracle.synth_fasta(20000)                    # ~33 MB, 60-col lines
for label, data in (("wrapped60", np.tile(body, 30)),):
    n=len(data); d_in=device.upload(data.tobytes()); scan=device.FastaScan(n, capacity_records=20000*30+16)
    ms=time_ms(lambda: scan.launch(d_in)); r=scan.fetch()
    assert r.error_code==0 and r.n_records==20000*30, (r.error_code, r.n_records)
    out[label]={"bytes":n,"records":int(r.n_records),"ms":ms,"GBps":n/ms/1e6}
print(json.dumps(out,indent=1))
