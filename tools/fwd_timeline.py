import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ks=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows]
ks.sort()
# last replay: find last k_vox_insert
idx=[i for i,k in enumerate(ks) if "k_vox_insert" in k[2]]
a=idx[-2]; b=idx[-1]
seg=ks[a:b]
t0=seg[0][0]
print("replay wall %.1f us, %d kernels"%((seg[-1][1]-t0)/1e3, len(seg)))
busy=0; last_end=t0
# union busy time
iv=sorted((s,e) for s,e,_,_ in seg)
cur_s,cur_e=iv[0]
for s,e in iv[1:]:
    if s<=cur_e: cur_e=max(cur_e,e)
    else: busy+=cur_e-cur_s; cur_s,cur_e=s,e
busy+=cur_e-cur_s
print("GPU busy (any queue) %.1f us"%(busy/1e3))
for s,e,n,q in seg:
    nm=n.replace("(anonymous namespace)::","").replace("void ","")[:58]
    print("%8.1f %8.1f  q%s  %6.1f us  %s"%((s-t0)/1e3,(e-t0)/1e3,q,(e-s)/1e3,nm))
