"""Per-layer table out of two rocprofv3 rocpd databases (direct/direct_results.db, lds/lds_results.db in the cwd) of tools/dac_probe.py."""
import sqlite3, re, sys
import numpy as np
res={}
for tag in ("direct","lds"):
    db=sqlite3.connect(f"{tag}/{tag}_results.db")
    cur=db.cursor()
    tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')]
    names={r[0]:r[1] for r in cur.execute(f"select id, kernel_name from {ks[-1]}")}
    rows=list(cur.execute("select kernel_id,start,end,grid_size_x,grid_size_y,workgroup_size_x from rocpd_kernel_dispatch order by start"))
    seqs=[];curseq=None
    for k,s,e,gx,gy,wx in rows:
        n=names[k]
        if 'rvq_gather' in n:
            curseq=[];seqs.append((gx,curseq))
        if curseq is not None: curseq.append((n,e-s,gx,gy,wx))
    full=[q for gx,q in seqs if gx==860*256 and len(q)==31]
    print(tag,len(seqs),len(full), set(len(q) for gx,q in seqs), set(gx for gx,q in seqs))
    use=full[-10:]
    avg=np.mean([[d for _,d,_,_,_ in q] for q in use],axis=0)
    res[tag]=(use[0],avg)
d0,da=res["direct"]; l0,la=res["lds"]
def short(n):
    m=re.search(r'(conv_\w+_kernel)ILi(\d+)ELi?b?(\d+)(?:ELi(\d+)ELi(\d+))?',n)
    return n.split('(')[0][-30:] if not m else m.group(1)+"<"+",".join(x for x in m.groups()[1:] if x)+">"
print(f"{'i':>2} {'direct kernel':30s} {'us':>7} | {'lds kernel':30s} {'us':>7} grid")
for i in range(31):
    print(f"{i:2d} {short(d0[i][0]):30s} {da[i]/1e3:7.1f} | {short(l0[i][0]):30s} {la[i]/1e3:7.1f} {l0[i][2]//l0[i][4]}x{l0[i][3]} wg{l0[i][4]}")
print("total", da.sum()/1e3, la.sum()/1e3)
