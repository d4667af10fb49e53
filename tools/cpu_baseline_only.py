import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
if __name__ == "__main__":
    t0=time.time()
    r = bench.cpu_baseline("cfg3", None, None, 1<<12, 1<<18)
    print(time.time()-t0, "s")
    print(json.dumps({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='sample'}) for k,v in r.items()}, indent=1)[:3000])
