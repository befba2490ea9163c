import os, sys, json, torch
sys.path.insert(0, os.getcwd())
from sgracex1_amd import graphs, ops
from sgracex1_amd import _lib
from sgracex1_amd.hipevents import Event
def timed(fn, iters=12):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.current_stream().cuda_stream; ts=[]
    for _ in range(iters):
        b,e=Event(),Event(); b.record(s); fn(); e.record(s); ts.append(b.elapsed_ms(e))
    return min(ts)
n=1<<22
H=torch.rand((n,64),device='cuda').half(); D=torch.empty((n,64),device='cuda',dtype=torch.float16)
for thr in ('0.7','2.0'):
    os.environ['SGX_PLAN_REORDER_BELOW']=thr
    _lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
    A=graphs.uniform_graph(n,100_000_000,seed=12345)
    A.plan
    print(json.dumps({'reorder_below':thr,'reordered':A.plan.reordered,'util':A.plan.natural_utilization,'ms':timed(lambda: ops.spmm(A,H,relu=True,out=D))}))
    del A
