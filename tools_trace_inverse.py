#!/usr/bin/env python3
"""Reads gpurun_out/prof_inv/inv_results.db (tools_trace_inverse.sh): the kernels of the LAST
explicit inverse of the run in launch order, per phase (potrf | trtri | lauum): wall time, time
per kernel name, and the GEMM launches by grid size."""
import os
ROOT = os.path.dirname(os.path.abspath(__file__))
import re
import sqlite3

con = sqlite3.connect(os.path.join(ROOT, 'gpurun_out', 'prof_inv', 'inv_results.db'))
rows=list(con.execute("select name,start,end,grid_x,workgroup_x,grid_y,grid_z from kernels order by start"))
def short(n):
    n=re.sub(r"\(anonymous namespace\)::","",n); return re.sub(r"^void ","",n).split("(")[0].replace("eps::k::","")[:50]
rows=[(short(r[0]),)+r[1:] for r in rows]
sym=[i for i,r in enumerate(rows) if r[0].startswith("Symmetrize")]
first_potrf=[i for i,r in enumerate(rows) if r[0].startswith("PotrfDiagStep")]
lo=[i for i in first_potrf if i>sym[-2]][0] if len(sym)>1 else first_potrf[0]; hi=sym[-1]
seg=rows[lo:hi+1]
phase="potrf"; agg={}; wall={}
for r in seg:
    n,s,e,gx,wx,gy,gz=r
    wgs=(gx//max(wx,1))*max(gy,1)*max(gz,1)
    for k in ((phase,n,None),(phase,n,wgs)):
        a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=(e-s)/1e3
    w=wall.setdefault(phase,[s,e]); w[1]=e
    if n.startswith("TrtriDiagBlocks"): phase="trtri"
    elif phase=="trtri" and n.startswith("EwKernel"): phase="lauum"
for ph in ("potrf","trtri","lauum"):
    print("== %s wall %.2f ms"%(ph,(wall[ph][1]-wall[ph][0])/1e6))
    for (p,n,g),(c,t) in sorted(agg.items(), key=lambda x:-x[1][1]):
        if p==ph and g is None: print("   %-50s calls %4d total %8.1f us avg %7.1f"%(n,c,t,t/c))
    for (p,n,g),(c,t) in sorted(agg.items(), key=lambda x:-x[1][1])[:60]:
        if p==ph and g is not None and n.startswith("Gemm"): print("        wgs %5d %-40s calls %4d total %8.1f us avg %7.1f"%(g,n[:40],c,t,t/c))
